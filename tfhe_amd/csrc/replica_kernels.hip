// Blind rotation with the negacyclic products evaluated the way the REFERENCE evaluates them: its
// double-precision complex FFT (reference src/types.hpp:152-197 and :237-292), replayed operation
// by operation, rounding included.  The reference's product is the exact one or one unit lower,
// depending on the rounding of that FFT; a context created with TFHE_AMD_CTX_REFERENCE_FFT therefore
// returns ciphertexts that are bit-identical to those of the reference compiled without FMA
// contraction (g++ -O2 on x86-64, the reference's CI build), where the default exact NTT path
// returns the exact ones.  Two kernels: a parameter-generic one (one workgroup per gate, rings
// 32 .. 1024) and, further down, a wave-per-gate one for the default parameter set.
//
// What is replayed (M = N/2): fold N reals into M complex numbers (first half real parts, second half
// imaginary parts); twist by powers of e^(i pi / N) built by repeated multiplication; an M-point
// transform whose stage i multiplies the upper half of block j by omega^bitrev(2j), adds and
// subtracts, followed by a bit-reversal permutation; pointwise product; the same transform with the
// omega table reversed from index 1; scale by 1/M; untwist; truncating double -> torus conversion of
// every part (reference src/types.hpp:28-33); the (K+1)L products of a column are summed mod 2^32
// after each has been rounded on its own (reference src/types.hpp:433-445).  Complex products are
// (ac - bd, ad + bc) in plain IEEE double operations; the library is built with -ffp-contract=off.
// The power tables come from the host (capi.hip), where libm's cos and sin are the ones the
// reference's std::polar calls.  The butterflies of a stage touch disjoint pairs, so running them in
// place and in parallel gives the same values as the reference's array-per-stage loop; the final
// permutation is folded into the indices of the next read.
#include "fft256_wave.hpp"
#include "kernels.hpp"

namespace tfhe_amd {

namespace {

constexpr int kRepThreads = 256;

__device__ __forceinline__ d2 c_mul(d2 x, d2 y) {
    const double ac = x.x * y.x, bd = x.y * y.y, ad = x.x * y.y, bc = x.y * y.x;
    d2 r;
    r.x = ac - bd;
    r.y = ad + bc;
    return r;
}

__device__ __forceinline__ int bitrev(int v, int bits) { return (int)(__brev((unsigned)v) >> (32 - bits)); }

// reference src/types.hpp:28-33; the conversion goes through 64 bits as on x86-64, where a value
// that rounds up to exactly 2^32 wraps to 0
__device__ __forceinline__ uint32_t torus_from_double_ref(double d) {
    d = fmod(d, 1.0);
    if (d < 0.0) d += 1.0;
    d *= 4294967296.0;
    return (uint32_t)(unsigned long long)d;
}

// `count` transforms of M points each, in place in LDS (buf: [count][M]), all stages.
// omega: [M] table for this direction.  Output in bit-reversed positions.
__device__ void fft_block(d2* buf, int count, int M, int bits, const d2* omega) {
    const int half_total = count * (M >> 1);
    for (int i = 0; i < bits; i++) {
        const int span = 1 << (bits - i), half = span >> 1;
        __syncthreads();
        for (int t = threadIdx.x; t < half_total; t += kRepThreads) {
            const int arr = t / (M >> 1), b = t - arr * (M >> 1);
            const int j = b / half, k = b - j * half;
            d2* a = buf + (size_t)arr * M + j * span + k;
            const d2 even = a[0];
            const d2 odd = c_mul(a[half], omega[bitrev(j << 1, bits)]);
            d2 s, d;
            s.x = even.x + odd.x; s.y = even.y + odd.y;
            d.x = even.x - odd.x; d.y = even.y - odd.y;
            a[0] = s;
            a[half] = d;
        }
    }
    __syncthreads();
}

struct RepShape {
    int N, K1, L, R, bg, bits;  // bits = log2(N / 2)
};

}  // namespace

// Tables: [4][M] complex = twist powers, forward omega powers, untwist powers, reversed omega powers.
// Key spectra: for every key polynomial, fft(twist(fold(poly / 2^32))) / M in natural order, [M] complex.
// wave_order != 0 (N = 512 only): entry 64 k + lane holds the spectrum value of index
// bitrev8(4 lane + k), the order in which the wave-per-gate kernel's first transform leaves its output.
__global__ void __launch_bounds__(kRepThreads)
replica_key_kernel(const uint32_t* __restrict__ bk, const d2* __restrict__ tables, d2* __restrict__ spec, int N,
                   int bits, int wave_order) {
    extern __shared__ d2 s_rep[];
    const int M = N >> 1;
    d2* s_omega = s_rep;      // [M]
    d2* buf = s_rep + M;      // [M]
    const size_t poly = blockIdx.x;
    const uint32_t* b = bk + poly * N;
    for (int i = threadIdx.x; i < M; i += kRepThreads) {
        s_omega[i] = tables[M + i];
        d2 v;
        v.x = (double)b[i] / 4294967296.0;
        v.y = (double)b[i + M] / 4294967296.0;
        buf[i] = c_mul(v, tables[i]);
    }
    fft_block(buf, 1, M, bits, s_omega);
    // The reference multiplies the second transform's output by 1/M, a power of two, which commutes
    // exactly with every rounding in between: the factor is applied here, once per key.
    const double scale = 1.0 / M;
    for (int i = threadIdx.x; i < M; i += kRepThreads) {
        // natural index i sits at buf[bitrev(i)]; in wave order entry i = 64 k + lane wants natural
        // index bitrev8(4 lane + k), which sits at buf[4 lane + k]
        d2 v = wave_order ? buf[4 * (i & 63) + (i >> 6)] : buf[bitrev(i, bits)];
        v.x *= scale;
        v.y *= scale;
        spec[poly * M + i] = v;
    }
}

// One workgroup per gate.  Same interface as blind_rotate_generic_kernel.
__global__ void __launch_bounds__(kRepThreads)
blind_rotate_replica_kernel(const d2* __restrict__ spec, const d2* __restrict__ tables, const uint32_t* __restrict__ c,
                            const uint32_t* __restrict__ tv, uint32_t* __restrict__ out, RepShape s, int n,
                            int extract) {
    extern __shared__ d2 s_rep[];
    const int N = s.N, M = N >> 1, W = s.K1 * N, R = s.R, bits = s.bits;
    d2* s_twist = s_rep;            // [M]
    d2* s_omega = s_twist + M;      // [M]
    d2* s_untwist = s_omega + M;    // [M]
    d2* s_omega_r = s_untwist + M;  // [M]
    d2* lhs = s_omega_r + M;        // [R][M] spectra of the digit polynomials (bit-reversed positions)
    d2* prod = lhs + (size_t)R * M; // [R][M]
    uint32_t* acc = reinterpret_cast<uint32_t*>(prod + (size_t)R * M);  // [K1][N]
    uint32_t* diff = acc + W;                                           // [K1][N]
    const size_t gate = blockIdx.x;
    const uint32_t* cg = c + gate * (size_t)(n + 1);
    const int sh = 32 - (bits + 2);  // 32 - log2(2N)

    for (int i = threadIdx.x; i < 4 * M; i += kRepThreads) s_rep[i] = tables[i];
    const int bbar = (int)(cg[0] >> sh);
    for (int e = threadIdx.x; e < W; e += kRepThreads) {
        const int q = e / N, j = e - q * N;
        const int src = (j + bbar) & (2 * N - 1);
        const uint32_t v = tv ? tv[q * N + (src & (N - 1))] : (q == 0 ? (1u << 29) : 0u);
        acc[e] = (src & N) ? 0u - v : v;
    }
    __syncthreads();

    uint32_t off = 1u << (32 - s.L * s.bg - 1);
    for (int l = 0; l < s.L; l++) off += 1u << (32 - s.bg * (l + 1) + s.bg - 1);
    const uint32_t mask = (1u << s.bg) - 1;
    const int half_digit = 1 << (s.bg - 1);

    for (int i = 0; i < n; i++) {
        const int abar = (int)((cg[1 + i] + (1u << (sh - 1))) >> sh);
        for (int e = threadIdx.x; e < W; e += kRepThreads) {
            const int q = e / N, j = e - q * N;
            const int src = (j - abar) & (2 * N - 1);
            const uint32_t v = acc[q * N + (src & (N - 1))];
            diff[e] = ((src & N) ? 0u - v : v) - acc[e];
        }
        __syncthreads();
        // digit polynomial r = q*L + l, folded and twisted
        for (int e = threadIdx.x; e < R * M; e += kRepThreads) {
            const int r = e / M, m = e - r * M, q = r / s.L, l = r - q * s.L;
            const uint32_t t0 = diff[q * N + m] + off, t1 = diff[q * N + m + M] + off;
            d2 v;
            v.x = (double)((int)((t0 >> (32 - s.bg * (l + 1))) & mask) - half_digit);
            v.y = (double)((int)((t1 >> (32 - s.bg * (l + 1))) & mask) - half_digit);
            lhs[e] = c_mul(v, s_twist[m]);
        }
        fft_block(lhs, R, M, bits, s_omega);
        const d2* key = spec + (size_t)i * R * s.K1 * M;
        for (int col = 0; col < s.K1; col++) {
            for (int e = threadIdx.x; e < R * M; e += kRepThreads) {
                const int r = e / M, m = e - r * M;
                prod[e] = c_mul(lhs[r * M + bitrev(m, bits)], key[((size_t)r * s.K1 + col) * M + m]);
            }
            fft_block(prod, R, M, bits, s_omega_r);
            // every product rounded to the torus on its own, then summed mod 2^32 and added to acc
            for (int m = threadIdx.x; m < M; m += kRepThreads) {
                uint32_t lo = 0, hi = 0;
                for (int r = 0; r < R; r++) {
                    const d2 v = c_mul(prod[r * M + bitrev(m, bits)], s_untwist[m]);  // 1/M is in the key spectra
                    lo += torus_from_double_ref(v.x);
                    hi += torus_from_double_ref(v.y);
                }
                acc[col * N + m] += lo;
                acc[col * N + m + M] += hi;
            }
            __syncthreads();
        }
    }
    if (extract) {
        uint32_t* o = out + gate * (size_t)((s.K1 - 1) * N + 1);
        for (int e = threadIdx.x; e < (s.K1 - 1) * N; e += kRepThreads) {
            const int q = e / N, j = e - q * N;
            const uint32_t v = acc[(1 + q) * N + j];
            if (j == 0) o[1 + q * N] = v;
            else        o[1 + q * N + (N - j)] = 0u - v;
        }
        if (threadIdx.x == 0) o[0] = acc[0];
    } else {
        for (int e = threadIdx.x; e < W; e += kRepThreads) out[gate * (size_t)W + e] = acc[e];
    }
}

// ------------------------------------------------------------------------------------------
// The default parameter set (N = 512, K = 2, L = 2) at speed: one WAVE per gate, the reference's
// 256-point transforms evaluated in registers (fft256_wave.hpp), same structure as
// blind_rotate_512_kernel.  Per step and accumulator polynomial q the two digit polynomials
// (levels 0 and 1) run as one interleaved pair of transforms: fold and twist, forward transforms,
// then per column: pointwise products with the two key spectra (stored in the order the forward
// transform leaves its output, and carrying the 1/M factor), the pair of second transforms (run
// from that order, so nothing is permuted in between, and ending in natural order), untwist,
// truncating conversion of both parts, wrap-around sum.  The reference rounds every product on its own, so there are
// (K+1) L (K+1) = 18 second transforms per step where the exact path needs K + 1 = 3 inverse NTTs.
// Pairing hides the LDS latency of one transform's transposition behind the other's butterflies;
// the first transposition of every transform stays in registers (half-wave and row swaps).
// tables: the [4][256] base tables followed by the block described in fft256_wave.hpp.
// ------------------------------------------------------------------------------------------
template <int K, int L>
__global__ void __launch_bounds__(256, 2)
blind_rotate_512_replica_kernel(const d2* __restrict__ spec, const d2* __restrict__ tables,
                                const uint32_t* __restrict__ c, const uint32_t* __restrict__ tv,
                                uint32_t* __restrict__ out, int n_gates, int n, int extract) {
    static_assert(L == 2, "the two levels of a digit polynomial run as one interleaved pair of transforms");
    constexpr int K1 = K + 1, R = K1 * L, BG = 8, M = 256;
    __shared__ d2 s_rows[2 * kFftRows * 3];
    __shared__ d2 s_untwist[M];  // natural order
    __shared__ d2 s_tr[4][2][kHalfTrWords];
    __shared__ uint32_t s_acc[4][K1][kN];

    for (int i = threadIdx.x; i < 2 * kFftRows * 3; i += 256) s_rows[i] = tables[4 * M + i];
    for (int i = threadIdx.x; i < M; i += 256) s_untwist[i] = tables[2 * M + i];
    __syncthreads();

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int gate = blockIdx.x * 4 + wave;
    if (gate >= n_gates) return;  // waves are independent from here on
    d2* tr0 = s_tr[wave][0];
    d2* tr1 = s_tr[wave][1];
    uint32_t(*acc)[kN] = s_acc[wave];
    const uint32_t* cg = c + (size_t)gate * (n + 1);
    const d2* rows_fwd = s_rows + kFftFwdRows;
    const d2* rows_inv = s_rows + kFftInvRows;
    {
        const int bbar = (int)(cg[0] >> 22);
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int j = 64 * k + lane;
            const int s = (j + bbar) & (2 * kN - 1);
#pragma unroll
            for (int q = 0; q < K1; q++) {
                const uint32_t v = tv ? tv[q * kN + (s & (kN - 1))] : (q == 0 ? (1u << 29) : 0u);
                acc[q][j] = (s & kN) ? 0u - v : v;
            }
        }
    }
    wave_lds_fence();

    uint32_t offset = 1u << (32 - L * BG - 1), flip = 0u;
#pragma unroll
    for (int l = 0; l < L; l++) {
        offset += 1u << (32 - BG * (l + 1) + BG - 1);
        flip |= 1u << (32 - BG * (l + 1) + BG - 1);
    }
    // wave-uniform twiddles of the second stage: kept in scalar registers
    auto uniform = [](double v) {
        return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)),
                                __builtin_amdgcn_readfirstlane(__double2loint(v)));
    };
    d2 w23_fwd = rows_fwd[2], w23_inv = rows_inv[2];
    w23_fwd.x = uniform(w23_fwd.x); w23_fwd.y = uniform(w23_fwd.y);
    w23_inv.x = uniform(w23_inv.x); w23_inv.y = uniform(w23_inv.y);
    const d2* row2_fwd = rows_fwd + 3 * (fft_row_offset(2) + fft_row<2>(lane));
    const d2* twist = tables + lane;  // powers for the points 64 k + lane

    uint32_t cvals = 0;
    for (int i = 0; i < n; i++) {
        if ((i & 63) == 0) cvals = (i + lane < n) ? cg[1 + i + lane] : 0u;
        const uint32_t ci = (uint32_t)__builtin_amdgcn_readlane((int)cvals, i & 63);
        const int abar = (int)((ci + (1u << 21)) >> 22);

        uint32_t lo[K1][4], hi[K1][4];
#pragma unroll
        for (int col = 0; col < K1; col++)
#pragma unroll
            for (int k = 0; k < 4; k++) lo[col][k] = hi[col][k] = 0u;

        const d2* key_row = spec + (size_t)i * (R * K1 * M);
#pragma unroll 1
        for (int q = 0; q < K1; q++) {
            // spectra of the key polynomials (q, level 0) and (q, level 1), column 0
            const d2* key0 = key_row + (size_t)(q * L) * K1 * M;
            const d2* key1 = key0 + (size_t)K1 * M;
            d2 y0[4], y1[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                y0[k] = key0[64 * k + lane];
                y1[k] = key1[64 * k + lane];
            }
            uint32_t t[8];
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const int sidx = (64 * k + lane - abar) & (2 * kN - 1);
                const uint32_t neg = 0u - (uint32_t)((sidx >> 9) & 1);
                const uint32_t v = acc[q][sidx & (kN - 1)];
                t[k] = ((((v ^ neg) - neg) - acc[q][64 * k + lane]) + offset) ^ flip;
            }
            // fold (real part: coefficient m, imaginary part: coefficient m + 256) and twist, both levels
            Cx4 x0, x1;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const d2 tw = twist[64 * k];
                const double dr0 = (double)(((int32_t)t[k]) >> (32 - BG));
                const double di0 = (double)(((int32_t)t[k + 4]) >> (32 - BG));
                const double dr1 = (double)(((int32_t)(t[k] << BG)) >> (32 - BG));
                const double di1 = (double)(((int32_t)(t[k + 4] << BG)) >> (32 - BG));
                const double ac0 = dr0 * tw.x, bd0 = di0 * tw.y, ad0 = dr0 * tw.y, bc0 = di0 * tw.x;
                const double ac1 = dr1 * tw.x, bd1 = di1 * tw.y, ad1 = dr1 * tw.y, bc1 = di1 * tw.x;
                x0.re[k] = ac0 - bd0;
                x0.im[k] = ad0 + bc0;
                x1.re[k] = ac1 - bd1;
                x1.im[k] = ad1 + bc1;
            }
            fft256_wave2(x0, x1, rows_fwd, w23_fwd, row2_fwd, tr0, tr1, lane);
#pragma unroll
            for (int col = 0; col < K1; col++) {
                Cx4 p0, p1;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const double ac0 = x0.re[k] * y0[k].x, bd0 = x0.im[k] * y0[k].y;
                    const double ad0 = x0.re[k] * y0[k].y, bc0 = x0.im[k] * y0[k].x;
                    const double ac1 = x1.re[k] * y1[k].x, bd1 = x1.im[k] * y1[k].y;
                    const double ad1 = x1.re[k] * y1[k].y, bc1 = x1.im[k] * y1[k].x;
                    p0.re[k] = ac0 - bd0;
                    p0.im[k] = ad0 + bc0;
                    p1.re[k] = ac1 - bd1;
                    p1.im[k] = ad1 + bc1;
                }
                if (col + 1 < K1) {  // the next column's spectra arrive behind this column's transforms
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        y0[k] = key0[(col + 1) * M + 64 * k + lane];
                        y1[k] = key1[(col + 1) * M + 64 * k + lane];
                    }
                }
                fft256_wave2_rev(p0, p1, rows_inv, w23_inv, tr0, tr1, lane);
                double vr[8], vi[8];  // untwisted real and imaginary parts: level 0 in [k], level 1 in [4 + k]
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const d2 u = s_untwist[64 * k + lane];
                    const double ac0 = p0.re[k] * u.x, bd0 = p0.im[k] * u.y;
                    const double ad0 = p0.re[k] * u.y, bc0 = p0.im[k] * u.x;
                    const double ac1 = p1.re[k] * u.x, bd1 = p1.im[k] * u.y;
                    const double ad1 = p1.re[k] * u.y, bc1 = p1.im[k] * u.x;
                    vr[k] = ac0 - bd0;
                    vr[4 + k] = ac1 - bd1;
                    vi[k] = ad0 + bc0;
                    vi[4 + k] = ad1 + bc1;
                }
                uint32_t tr_[8], ti_[8];
                torus_from_double_16(tr_, ti_, vr, vi);
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    lo[col][k] += tr_[k] + tr_[4 + k];
                    hi[col][k] += ti_[k] + ti_[4 + k];
                }
            }
        }
        wave_lds_fence();
#pragma unroll
        for (int col = 0; col < K1; col++)
#pragma unroll
            for (int k = 0; k < 4; k++) {
                acc[col][64 * k + lane] += lo[col][k];
                acc[col][64 * k + lane + M] += hi[col][k];
            }
        wave_lds_fence();
    }

    if (!extract) {
        uint32_t* o = out + (size_t)gate * (K1 * kN);
#pragma unroll
        for (int q = 0; q < K1; q++)
#pragma unroll
            for (int k = 0; k < 8; k++) o[q * kN + 64 * k + lane] = acc[q][64 * k + lane];
        return;
    }
    uint32_t* o = out + (size_t)gate * (K * kN + 1);
    if (lane == 0) o[0] = acc[0][0];
#pragma unroll
    for (int q = 0; q < K; q++)
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int j = 64 * k + lane;
            const uint32_t v = acc[1 + q][j];
            if (j == 0) o[1 + q * kN] = v;
            else        o[1 + q * kN + (kN - j)] = 0u - v;
        }
}

static size_t replica_lds_bytes(int N, int K1, int R) {
    const size_t M = (size_t)N / 2;
    return sizeof(d2) * (4 * M + 2 * (size_t)R * M) + sizeof(uint32_t) * (size_t)2 * K1 * N;
}

// the reference multiplies exactly below ring degree 32 (src/types.hpp:225-236): nothing to replay there
bool replica_has_kernel(int N, int K, int L) {
    if (N < 32 || N > 2048 || (N & (N - 1))) return false;
    return replica_lds_bytes(N, K + 1, (K + 1) * L) <= 160 * 1024;
}

hipError_t launch_replica_key(const uint32_t* bk, const d2* tables, d2* spec, size_t n_polys, int N, int wave_order,
                              hipStream_t stream) {
    if (n_polys == 0) return hipSuccess;
    int bits = 0;
    while ((1 << bits) < N / 2) bits++;
    const size_t lds = sizeof(d2) * (size_t)N;  // 2 M
    if (wave_order && N != 512) return hipErrorInvalidValue;
    replica_key_kernel<<<(unsigned)n_polys, kRepThreads, lds, stream>>>(bk, tables, spec, N, bits, wave_order);
    return hipGetLastError();
}

bool replica_has_fast_kernel(int N, int K, int L, int bg) { return N == 512 && K == 2 && L == 2 && bg == 8; }

hipError_t launch_blind_rotate_replica(const d2* spec, const d2* tables, const uint32_t* c, const uint32_t* tv,
                                       uint32_t* out, size_t n_gates, int n, int N, int K, int L, int bg, int extract,
                                       int generic, hipStream_t stream) {
    if (n_gates == 0) return hipSuccess;
    if (!replica_has_kernel(N, K, L)) return hipErrorInvalidValue;
    if (!generic && replica_has_fast_kernel(N, K, L, bg)) {
        blind_rotate_512_replica_kernel<2, 2><<<(unsigned)((n_gates + 3) / 4), 256, 0, stream>>>(
            spec, tables, c, tv, out, (int)n_gates, n, extract);
        return hipGetLastError();
    }
    RepShape s{N, K + 1, L, (K + 1) * L, bg, 0};
    while ((1 << s.bits) < N / 2) s.bits++;
    const size_t lds = replica_lds_bytes(N, s.K1, s.R);
    hipError_t e = hipFuncSetAttribute((const void*)blind_rotate_replica_kernel,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    blind_rotate_replica_kernel<<<(unsigned)n_gates, kRepThreads, lds, stream>>>(spec, tables, c, tv, out, s, n, extract);
    return hipGetLastError();
}

}  // namespace tfhe_amd
