// gfx950 kernels of the gate-bootstrapping path and their launchers.
//
//   nand_combine_kernel     c = (1/8, 0..) - (a + b)                 reference src/hom_nand.hpp:76-78
//   bk_transform_512_kernel bootstrapping key -> NTT domain (once per key)
//   blind_rotate_512_kernel the whole blind rotation + sample extraction of one gate per wave
//                                                                    reference src/trgsw.hpp:131-182,
//                                                                    src/trlwe.hpp:109-122
//   key_switch_kernel       identity key switch, a tile of gates per workgroup
//                                                                    reference src/tlwe_key_switching.hpp:47-65
#include "kernels.hpp"
#include "ntt512.hpp"

namespace tfhe_amd {

// ------------------------------------------------------------------------------------------
// NAND linear step
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) nand_combine_kernel(uint32_t* __restrict__ c, const uint32_t* __restrict__ a,
                                                           const uint32_t* __restrict__ b, size_t total, int n1) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const uint32_t mu = (i % (size_t)n1) == 0 ? (1u << 29) : 0u;
    c[i] = mu - (a[i] + b[i]);
}

// ------------------------------------------------------------------------------------------
// Bootstrapping key -> NTT domain.  One wave per polynomial.  The stored value is
// NTT(centred bk) / N, centred, as a double.  Storage order inside a polynomial: pair index
// kk in 0..3, lane, two values (registers 2kk, 2kk+1 of layout C), so that one wave reads
// one contiguous KiB per load instruction.
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) bk_transform_512_kernel(const NttUniform u, const d2* __restrict__ tw,
                                                               const uint32_t* __restrict__ bk,
                                                               d2* __restrict__ bk_ntt, int n_polys) {
    __shared__ d2 s_tw[kTwPairs];
    __shared__ double s_tr[4][kTrPad];
    for (int i = threadIdx.x; i < kTwPairs; i += 256) s_tw[i] = tw[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int poly = blockIdx.x * 4 + wave;
    if (poly >= n_polys) return;
    const uint32_t* src = bk + (size_t)poly * kN;
    double x[8];
#pragma unroll
    for (int k = 0; k < 8; k++) x[k] = (double)(int32_t)src[64 * k + lane];
    ntt512_forward<false>(x, u, s_tw, s_tr[wave], lane);
    d2* dst = bk_ntt + (size_t)poly * (kN / 2);
#pragma unroll
    for (int kk = 0; kk < 4; kk++) {
        d2 v;
        v.x = reduce(mulmod_tw(x[2 * kk], u.ninv, u.ninvp, u.p), u.p, u.pinv);
        v.y = reduce(mulmod_tw(x[2 * kk + 1], u.ninv, u.ninvp, u.p), u.p, u.pinv);
        dst[kk * 64 + lane] = v;
    }
}

// ------------------------------------------------------------------------------------------
// Blind rotation, default ring (N = 512), K and L as template parameters, 8-bit gadget.
// One wave owns one gate for all n steps; the accumulator lives in LDS (u32, natural
// order) because every step reads it rotated by a data-dependent amount.
//
// Step i (reference src/trgsw.hpp:159-168, :100-117):
//   for every accumulator polynomial q and level l:
//       diff  = acc_q * X^abar - acc_q                  (src/types.hpp:299-312)
//       digit = level l of the signed gadget decomposition of diff   (src/trgsw.hpp:26-50)
//       D     = NTT(digit)
//       out_c += D (.) BK_ntt[i][q*L + l][c]  for c in 0..K
//   acc_c += INTT(out_c) mod 2^32
// The decomposition is evaluated in closed form: adding 2^(bg-1) at every level's bit
// position turns the reference's "digit >= half -> subtract base and carry" chain into
// ordinary binary carries, so digit_l = ((a + offset) >> shift_l & mask) - half.
// ------------------------------------------------------------------------------------------
template <int K, int L>
__global__ void __launch_bounds__(256, 2)
blind_rotate_512_kernel(const NttUniform u, const d2* __restrict__ tw, const d2* __restrict__ bk_ntt,
                        const uint32_t* __restrict__ c, uint32_t* __restrict__ lv1, int n_gates, int n) {
    constexpr int K1 = K + 1, R = K1 * L, BG = 8;
    __shared__ d2 s_tw[kTwPairs];
    __shared__ double s_tr[4][kTrPad];
    __shared__ uint32_t s_acc[4][K1][kN];

    for (int i = threadIdx.x; i < kTwPairs; i += 256) s_tw[i] = tw[i];
    __syncthreads();

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int gate = blockIdx.x * 4 + wave;
    if (gate >= n_gates) return;  // waves are independent from here on: no further block barrier

    double* tr = s_tr[wave];
    uint32_t(*acc)[kN] = s_acc[wave];
    const uint32_t* cg = c + (size_t)gate * (n + 1);
    const double p = u.p, pinv = u.pinv;

    // acc = testvector * X^(-floor(b * 2N)); test vector = (1/8, .., 1/8 | 0 | 0)
    {
        const int bbar = (int)(cg[0] >> 22);
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int j = 64 * k + lane;
            const int s = (j + bbar) & (2 * kN - 1);
            acc[0][j] = (s & kN) ? 0u - (1u << 29) : (1u << 29);
#pragma unroll
            for (int q = 1; q < K1; q++) acc[q][j] = 0u;
        }
    }
    wave_lds_fence();

    // offset = rounding half-unit below the last level + 2^(BG-1) at every level's position
    uint32_t offset = 1u << (32 - L * BG - 1);
#pragma unroll
    for (int l = 0; l < L; l++) offset += 1u << (32 - BG * (l + 1) + BG - 1);

    uint32_t cvals = 0;
    for (int i = 0; i < n; i++) {
        if ((i & 63) == 0) cvals = (i + lane < n) ? cg[1 + i + lane] : 0u;
        const uint32_t ci = (uint32_t)__builtin_amdgcn_readlane((int)cvals, i & 63);
        const int abar = (int)((ci + (1u << 21)) >> 22);  // round(a_i * 2N), in [0, 2N]

        double out[K1][8];
#pragma unroll
        for (int q = 0; q < K1; q++)
#pragma unroll
            for (int k = 0; k < 8; k++) out[q][k] = 0.0;

        const d2* bk_row = bk_ntt + (size_t)i * (R * K1 * (kN / 2));
#pragma unroll 1
        for (int r = 0; r < R; r++) {
            const int q = r / L, level = r % L;
            const int shift = 32 - BG * (level + 1);
            double x[8];
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const int j = 64 * k + lane;
                const int s = (j - abar) & (2 * kN - 1);
                uint32_t v = acc[q][s & (kN - 1)];
                v = (s & kN) ? 0u - v : v;
                const uint32_t t = v - acc[q][j] + offset;
                x[k] = (double)((int)((t >> shift) & ((1u << BG) - 1)) - (1 << (BG - 1)));
            }
            ntt512_forward<true>(x, u, s_tw, tr, lane);
            const d2* bk_poly = bk_row + (size_t)r * K1 * (kN / 2);
#pragma unroll
            for (int col = 0; col < K1; col++) {
#pragma unroll
                for (int kk = 0; kk < 4; kk++) {
                    const d2 y = bk_poly[col * (kN / 2) + kk * 64 + lane];
                    out[col][2 * kk] += mulmod_var(x[2 * kk], y.x, p, pinv);
                    out[col][2 * kk + 1] += mulmod_var(x[2 * kk + 1], y.y, p, pinv);
                }
            }
        }
        // |out| <= R * 0.64 p <= 3.9 p for R = 6
#pragma unroll
        for (int col = 0; col < K1; col++) {
            ntt512_inverse(out[col], u, s_tw, tr, lane);
#pragma unroll
            for (int k = 0; k < 8; k++) acc[col][64 * k + lane] += to_torus(out[col][k], p, pinv);
        }
        wave_lds_fence();
    }

    // sample extraction at index 0 (reference src/trlwe.hpp:109-122):
    // lv1[0] = b[0]; lv1[1 + q N] = a_q[0]; lv1[1 + q N + i] = -a_q[N - i]
    uint32_t* o = lv1 + (size_t)gate * (K * kN + 1);
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

// ------------------------------------------------------------------------------------------
// Identity key switch.  A workgroup owns a tile of GT gates and walks the whole table once:
// for every (i, m) the 2^bb - 1 candidate rows are staged in LDS next to an all-zero row
// for digit 0, and every gate of the tile subtracts the row its digit selects.  Thread t
// accumulates words t, t + 256, .. of every gate in registers.
// ------------------------------------------------------------------------------------------
template <int GT, int WPT>
__global__ void __launch_bounds__(256) key_switch_kernel(const uint32_t* __restrict__ lv1, const uint32_t* __restrict__ ks,
                                                         uint32_t* __restrict__ out, int n_gates, int n1, int nlv1,
                                                         int T, int bb) {
    constexpr int ROW = 256 * WPT;  // padded row length in words
    constexpr int CHUNK = 256;      // lv1 coefficients whose digits are resident at a time
    extern __shared__ uint32_t s_mem[];
    const int nrows = 1 << bb;
    uint32_t* s_dig = s_mem;                    // [CHUNK][GT]
    uint32_t* s_rows = s_mem + CHUNK * GT;      // [2][nrows][ROW]
    const int tid = threadIdx.x;
    const int g0 = blockIdx.x * GT;
    const uint32_t round_offset = 1u << (32 - T * bb - 1);
    const int dshift = 32 - T * bb;
    const uint32_t omask = (uint32_t)nrows - 1;

    uint32_t acc[GT][WPT];
#pragma unroll
    for (int g = 0; g < GT; g++)
#pragma unroll
        for (int w = 0; w < WPT; w++) acc[g][w] = 0u;

    // the zero rows of both buffers
    for (int b = 0; b < 2; b++)
        for (int w = tid; w < ROW; w += 256) s_rows[(b * nrows) * ROW + w] = 0u;

    uint32_t pre[7][WPT];  // rows 1 .. nrows-1 of the next (i, m); nrows <= 8
    auto prefetch = [&](int i, int m) {
        const uint32_t* base = ks + ((size_t)i * T + m) * nrows * n1;
#pragma unroll
        for (int o = 1; o < 8; o++)
            if (o < nrows)
#pragma unroll
                for (int w = 0; w < WPT; w++) {
                    const int word = tid + 256 * w;
                    pre[o - 1][w] = word < n1 ? base[(size_t)o * n1 + word] : 0u;
                }
    };

    const int total = nlv1 * T;
    prefetch(0, 0);
    for (int idx = 0; idx < total; idx++) {
        const int i = idx / T, m = idx - i * T;
        if (m == 0 && (i % CHUNK) == 0) {
            __syncthreads();  // previous chunk's digits are no longer read
            const int ic = i + tid;
#pragma unroll 4
            for (int g = 0; g < GT; g++) {
                uint32_t v = 0u;
                if (g0 + g < n_gates && ic < nlv1) v = lv1[(size_t)(g0 + g) * (nlv1 + 1) + 1 + ic];
                s_dig[tid * GT + g] = (v + round_offset) >> dshift;
            }
        }
        uint32_t* buf = s_rows + (size_t)(idx & 1) * nrows * ROW;
#pragma unroll
        for (int o = 1; o < 8; o++)
            if (o < nrows)
#pragma unroll
                for (int w = 0; w < WPT; w++) buf[o * ROW + tid + 256 * w] = pre[o - 1][w];
        __syncthreads();
        if (idx + 1 < total) prefetch((idx + 1) / T, (idx + 1) % T);

        const int oshift = (T - 1 - m) * bb;
        const uint32_t* dig = s_dig + (i % CHUNK) * GT;
#pragma unroll
        for (int g = 0; g < GT; g++) {
            const uint32_t d = (uint32_t)__builtin_amdgcn_readfirstlane((int)dig[g]);
            const uint32_t o = (d >> oshift) & omask;
            const uint32_t* row = buf + o * ROW + tid;
#pragma unroll
            for (int w = 0; w < WPT; w++) acc[g][w] -= row[256 * w];
        }
    }

    // out = (b, 0, ..) - sum
#pragma unroll
    for (int g = 0; g < GT; g++) {
        if (g0 + g >= n_gates) break;
#pragma unroll
        for (int w = 0; w < WPT; w++) {
            const int word = tid + 256 * w;
            if (word >= n1) continue;
            uint32_t v = acc[g][w];
            if (word == 0) v += lv1[(size_t)(g0 + g) * (nlv1 + 1)];
            out[(size_t)(g0 + g) * n1 + word] = v;
        }
    }
}

// ------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------
hipError_t launch_nand_combine(uint32_t* c, const uint32_t* a, const uint32_t* b, size_t n_gates, int n1,
                               hipStream_t stream) {
    const size_t total = n_gates * (size_t)n1;
    if (total == 0) return hipSuccess;
    const unsigned blocks = (unsigned)((total + 255) / 256);
    nand_combine_kernel<<<blocks, 256, 0, stream>>>(c, a, b, total, n1);
    return hipGetLastError();
}

hipError_t launch_bk_transform_512(const NttUniform& u, const d2* tw, const uint32_t* bk, d2* bk_ntt, int n_polys,
                                   hipStream_t stream) {
    if (n_polys == 0) return hipSuccess;
    bk_transform_512_kernel<<<(n_polys + 3) / 4, 256, 0, stream>>>(u, tw, bk, bk_ntt, n_polys);
    return hipGetLastError();
}

hipError_t launch_blind_rotate_512(const NttUniform& u, const d2* tw, const d2* bk_ntt, const uint32_t* c,
                                   uint32_t* lv1, int n_gates, int n, int K, int L, hipStream_t stream) {
    if (n_gates == 0) return hipSuccess;
    const unsigned blocks = (unsigned)((n_gates + 3) / 4);
    if (K == 2 && L == 2)
        blind_rotate_512_kernel<2, 2><<<blocks, 256, 0, stream>>>(u, tw, bk_ntt, c, lv1, n_gates, n);
    else
        return hipErrorInvalidValue;
    return hipGetLastError();
}

size_t key_switch_lds_bytes(int gt, int wpt, int bb) {
    return sizeof(uint32_t) * ((size_t)256 * gt + (size_t)2 * (1 << bb) * 256 * wpt);
}

hipError_t launch_key_switch(const uint32_t* lv1, const uint32_t* ks, uint32_t* out, int n_gates, int n1, int nlv1,
                             int T, int bb, hipStream_t stream) {
    if (n_gates == 0) return hipSuccess;
    if (bb < 1 || bb > 3 || T * bb > 31 || n1 > 1024) return hipErrorInvalidValue;
    const int wpt = (n1 + 255) / 256;
    constexpr int GT = 32;
    const unsigned blocks = (unsigned)((n_gates + GT - 1) / GT);
    const size_t lds = key_switch_lds_bytes(GT, wpt, bb);
#define KS_CASE(W)                                                                                              \
    case W: {                                                                                                   \
        hipError_t e = hipFuncSetAttribute((const void*)key_switch_kernel<GT, W>,                               \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);               \
        if (e != hipSuccess) return e;                                                                          \
        key_switch_kernel<GT, W><<<blocks, 256, lds, stream>>>(lv1, ks, out, n_gates, n1, nlv1, T, bb);          \
        break;                                                                                                  \
    }
    switch (wpt) {
        KS_CASE(1)
        KS_CASE(2)
        KS_CASE(3)
        KS_CASE(4)
        default: return hipErrorInvalidValue;
    }
#undef KS_CASE
    return hipGetLastError();
}

}  // namespace tfhe_amd
