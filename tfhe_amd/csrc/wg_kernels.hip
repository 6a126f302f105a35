// Blind rotation for the parameter sets beside the default one (N = 2048 with K = 1, N = 1024 and
// N = 512 with K = 1 or 2):
// one workgroup of N/8 threads owns one gate and transforms every polynomial cooperatively
// (ntt_wg.hpp).
//
// The field's prime (p/2 = 2^49.6) is smaller than the largest negacyclic accumulation of these
// sets ((K+1) L N 2^7 2^31 = 2^51.6 for N = 2048, K = 1, L = 3), so the bootstrapping key is split
// before it is transformed: bk = hi * 2^shift + lo with signed halves, each half is transformed
// and accumulated separately (each sum is then below 2^36), and the two exact integer results
// are recombined modulo 2^32 after the inverse transforms.  Same reference semantics as
// kernels.hip (reference src/trgsw.hpp:131-182).
#include "kernels.hpp"
#include "ntt_wg.hpp"

namespace tfhe_amd {

// signed split v = hi * 2^shift + lo, lo in [-2^(shift-1), 2^(shift-1))
__device__ __forceinline__ int32_t split_lo(int32_t v, int shift) { return (int32_t)((uint32_t)v << (32 - shift)) >> (32 - shift); }

// One workgroup per (polynomial, part).  Stored value: NTT(part of centred bk) / N, centred, in
// the order [pair kk in 0..3][thread t][2] of the transform's output layout (position 8t + k).
template <int LOGN>
__global__ void __launch_bounds__(WgNtt<LOGN>::T)
bk_transform_wg_kernel(const NttUniform u, const d2* __restrict__ table, const uint32_t* __restrict__ bk,
                       d2* __restrict__ bk_ntt, int parts, int shift) {
    using W = WgNtt<LOGN>;
    __shared__ double s_tr[W::BUF_DOUBLES];
    const int t = threadIdx.x;
    const size_t poly = blockIdx.x / parts;
    const int part = blockIdx.x % parts;
    const uint32_t* src = bk + poly * W::N;
    double x[8];
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const int32_t v = (int32_t)src[k * W::T + t];
        int32_t piece = v;
        if (parts == 2) {
            const int32_t lo = split_lo(v, shift);
            piece = part == 0 ? lo : (int32_t)(((int64_t)v - lo) >> shift);
        }
        x[k] = (double)piece;
    }
    int parity = 0;
    W::template forward<false>(x, u, table, s_tr, parity, t);
    // transformed key: [.. r][col][part], i.e. raw polynomial index * parts + part = blockIdx.x
    d2* dst = bk_ntt + ((size_t)blockIdx.x) * (W::N / 2);
#pragma unroll
    for (int kk = 0; kk < 4; kk++) {
        d2 v;
        v.x = reduce(mulmod_tw(x[2 * kk], u.ninv, u.ninvp, u.p), u.p, u.pinv);
        v.y = reduce(mulmod_tw(x[2 * kk + 1], u.ninv, u.ninvp, u.p), u.p, u.pinv);
        dst[kk * W::T + t] = v;
    }
}

// Transformed key layout: [n][R][K1][parts] polynomials (part innermost), each [4][T][2] doubles.
// Two workgroups per CU (<= 256 registers per lane, 71 KB of LDS each): while one waits at a
// transpose barrier the other one computes.  Measured alternatives: one workgroup per CU with
// the L digit transforms of a polynomial interleaved in each wave (more parallelism per wave,
// fewer barriers) is 20 % slower, three workgroups per CU spill and are 35 % slower.
template <int LOGN, int K, int L, int PARTS>
__global__ void __launch_bounds__(WgNtt<LOGN>::T, 2)
blind_rotate_wg_kernel(const NttUniform u, const d2* __restrict__ table, const d2* __restrict__ bk_ntt,
                       const uint32_t* __restrict__ c, const uint32_t* __restrict__ tv, uint32_t* __restrict__ out,
                       int n, int shift, int extract) {
    using W = WgNtt<LOGN>;
    constexpr int N = W::N, T = W::T, K1 = K + 1, R = K1 * L, BG = 8;
    __shared__ double s_tr[W::BUF_DOUBLES];
    __shared__ uint32_t s_acc[K1][N];
    const int t = threadIdx.x;
    const size_t gate = blockIdx.x;
    const uint32_t* cg = c + gate * (size_t)(n + 1);
    const double p = u.p, pinv = u.pinv;
    constexpr int SH = 32 - (LOGN + 1);

    {
        const int bbar = (int)(cg[0] >> SH);
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int j = k * T + t;
            const int s = (j + bbar) & (2 * N - 1);
#pragma unroll
            for (int q = 0; q < K1; q++) {
                const uint32_t v = tv ? tv[q * N + (s & (N - 1))] : (q == 0 ? (1u << 29) : 0u);
                s_acc[q][j] = (s & N) ? 0u - v : v;
            }
        }
    }
    __syncthreads();

    uint32_t offset = 1u << (32 - L * BG - 1), flip = 0u;
#pragma unroll
    for (int l = 0; l < L; l++) {
        offset += 1u << (32 - BG * (l + 1) + BG - 1);
        flip |= 1u << (32 - BG * (l + 1) + BG - 1);
    }

    int parity = 0;
    for (int i = 0; i < n; i++) {
        const int abar = (int)((cg[1 + i] + (1u << (SH - 1))) >> SH);

        double acc_ntt[K1][PARTS][8];
#pragma unroll
        for (int col = 0; col < K1; col++)
#pragma unroll
            for (int part = 0; part < PARTS; part++)
#pragma unroll
                for (int k = 0; k < 8; k++) acc_ntt[col][part][k] = 0.0;

        const d2* bk_row = bk_ntt + (size_t)i * (R * K1 * PARTS) * (N / 2);
#pragma unroll 1
        for (int q = 0; q < K1; q++) {
            uint32_t tw_[8];
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const int sidx = (k * T + t - abar) & (2 * N - 1);
                const uint32_t neg = 0u - (uint32_t)((sidx >> LOGN) & 1);
                const uint32_t v = s_acc[q][sidx & (N - 1)];
                tw_[k] = ((((v ^ neg) - neg) - s_acc[q][k * T + t]) + offset) ^ flip;
            }
#pragma unroll 1
            for (int level = 0; level < L; level++) {  // rolled: keeps one transform's worth of code and registers live
                const int r = q * L + level;
                const d2* bk_poly = bk_row + (size_t)r * (K1 * PARTS) * (N / 2);
                // Key row: column 0 is requested before the transform, every further column while
                // the previous one is being multiplied, so that at most two columns are in flight.
                // Key row: K1 * PARTS tiles of 4 pairs per thread; tile 0 is requested before the
                // transform, every further tile while the previous one is being multiplied.
                d2 y[4], ynext[4];
#pragma unroll
                for (int kk = 0; kk < 4; kk++) y[kk] = bk_poly[kk * T + t];
                double x[8];
#pragma unroll
                for (int k = 0; k < 8; k++) x[k] = (double)(((int32_t)(tw_[k] << (BG * level))) >> (32 - BG));
                W::template forward<true>(x, u, table, s_tr, parity, t);  // |x| <= 3.2 p
#pragma unroll
                for (int tile = 0; tile < K1 * PARTS; tile++) {
                    const int col = tile / PARTS, part = tile % PARTS;
                    if (tile + 1 < K1 * PARTS) {
#pragma unroll
                        for (int kk = 0; kk < 4; kk++) ynext[kk] = bk_poly[(size_t)(tile + 1) * (N / 2) + kk * T + t];
                    }
#pragma unroll
                    for (int kk = 0; kk < 4; kk++) {
                        acc_ntt[col][part][2 * kk] += mulmod_var(x[2 * kk], y[kk].x, p, pinv);
                        acc_ntt[col][part][2 * kk + 1] += mulmod_var(x[2 * kk + 1], y[kk].y, p, pinv);
                    }
                    if (tile + 1 < K1 * PARTS) {
#pragma unroll
                        for (int kk = 0; kk < 4; kk++) y[kk] = ynext[kk];
                    }
                }
                // products are <= 1.41 p each: reduce after every third (running sum < 4.74 p)
                if (r % 3 == 2 && r != R - 1) {
#pragma unroll
                    for (int col = 0; col < K1; col++)
#pragma unroll
                        for (int part = 0; part < PARTS; part++)
#pragma unroll
                            for (int k = 0; k < 8; k++)
                                acc_ntt[col][part][k] = reduce(acc_ntt[col][part][k], p, pinv);
                }
            }
        }
#pragma unroll
        for (int col = 0; col < K1; col++) {
            uint32_t delta[8];
#pragma unroll
            for (int k = 0; k < 8; k++) delta[k] = 0u;
#pragma unroll
            for (int part = 0; part < PARTS; part++) {
                W::inverse(acc_ntt[col][part], u, table, s_tr, parity, t);
#pragma unroll
                for (int k = 0; k < 8; k++) delta[k] += to_torus(acc_ntt[col][part][k], p, pinv) << (part * shift);
            }
#pragma unroll
            for (int k = 0; k < 8; k++) s_acc[col][k * T + t] += delta[k];
        }
        __syncthreads();
    }

    if (!extract) {
        uint32_t* o = out + gate * (size_t)(K1 * N);
        for (int e = t; e < K1 * N; e += T) o[e] = s_acc[e / N][e % N];
        return;
    }
    uint32_t* o = out + gate * (size_t)(K * N + 1);
    if (t == 0) o[0] = s_acc[0][0];
    for (int e = t; e < K * N; e += T) {
        const int qq = e / N, j = e - qq * N;
        const uint32_t v = s_acc[1 + qq][j];
        if (j == 0) o[1 + qq * N] = v;
        else        o[1 + qq * N + (N - j)] = 0u - v;
    }
}

// ---- host side ---------------------------------------------------------------------------------

// parameter sets with an instantiation (8-bit gadget, L = 2 or 3): N = 2048 with K = 1, N = 1024 with
// K = 1 or 2, and N = 512 with L = 3 (the N = 512 sets with L = 2 have wave-per-gate kernels)
bool wg_has_kernel(int N, int K, int L, int bg) {
    if (bg != 8 || (L != 2 && L != 3)) return false;
    if (N == 2048) return K == 1;
    if (N == 1024) return K == 1 || K == 2;
    if (N == 512) return L == 3;  // L = 2: wave-per-gate kernels (kernels.hip)
    return false;
}

int wg_table_pairs(int N) {
    return N == 2048 ? WgNtt<11>::TABLE_PAIRS : N == 1024 ? WgNtt<10>::TABLE_PAIRS : N == 512 ? WgNtt<9>::TABLE_PAIRS : 0;
}

// number of key parts and the split position for a parameter set: one part if the largest
// accumulation (K+1) L N 2^(bg-1) 2^31 stays below p/2 with margin, else two halves of 16 bits
void wg_split(int N, int K, int L, int bg, int* parts, int* shift) {
    const double bound = (double)(K + 1) * L * N * (double)(1u << (bg - 1)) * 2147483648.0;
    if (bound < 0.49 * 1711528530149377.0) { *parts = 1; *shift = 0; }
    else { *parts = 2; *shift = 16; }
}

hipError_t launch_bk_transform_wg(const NttUniform& u, const d2* table, const uint32_t* bk, d2* bk_ntt, size_t n_polys,
                                  int N, int parts, int shift, hipStream_t stream) {
    if (n_polys == 0) return hipSuccess;
#define WG_KEY(LOGN)                                                                                                 \
    bk_transform_wg_kernel<LOGN><<<(unsigned)(n_polys * parts), WgNtt<LOGN>::T, 0, stream>>>(u, table, bk, bk_ntt, parts, \
                                                                                             shift)
    if (N == 2048) WG_KEY(11);
    else if (N == 1024) WG_KEY(10);
    else if (N == 512) WG_KEY(9);
    else return hipErrorInvalidValue;
#undef WG_KEY
    return hipGetLastError();
}

hipError_t launch_blind_rotate_wg(const NttUniform& u, const d2* table, const d2* bk_ntt, const uint32_t* c,
                                  const uint32_t* tv, uint32_t* out, size_t n_gates, int n, int N, int K, int L,
                                  int parts, int shift, int extract, hipStream_t stream) {
    if (n_gates == 0) return hipSuccess;
    if (!wg_has_kernel(N, K, L, 8)) return hipErrorInvalidValue;
    int want_parts = 0, want_shift = 0;
    wg_split(N, K, L, 8, &want_parts, &want_shift);
    if (parts != want_parts) return hipErrorInvalidValue;
#define WG_GO(LOGN, KK, LL, PP)                                                                                       \
    blind_rotate_wg_kernel<LOGN, KK, LL, PP><<<(unsigned)n_gates, WgNtt<LOGN>::T, 0, stream>>>(u, table, bk_ntt, c, tv, \
                                                                                               out, n, shift, extract)
    if (N == 2048) { if (L == 3) WG_GO(11, 1, 3, 2); else WG_GO(11, 1, 2, 2); }
    else if (N == 1024 && K == 1) { if (L == 3) WG_GO(10, 1, 3, 2); else WG_GO(10, 1, 2, 2); }
    else if (N == 1024) { if (L == 3) WG_GO(10, 2, 3, 2); else WG_GO(10, 2, 2, 2); }
    else if (K == 1) WG_GO(9, 1, 3, 2);
    else WG_GO(9, 2, 3, 2);
#undef WG_GO
    return hipGetLastError();
}

}  // namespace tfhe_amd
