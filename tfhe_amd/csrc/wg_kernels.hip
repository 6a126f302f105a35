// Blind rotation for the parameter sets beside the default one (N = 2048 with K = 1, N = 1024 and
// N = 512 with K = 1 or 2):
// one workgroup of N/8 threads owns one gate and transforms every polynomial cooperatively
// (ntt_wg.hpp).
//
// The field's prime (p/2 = 2^49.6) is smaller than the largest negacyclic accumulation of these
// sets ((K+1) L N 2^7 2^31 = 2^51.6 for N = 2048, K = 1, L = 3), so the bootstrapping key is split
// before it is transformed: bk = hi * 2^shift + lo with a signed remainder lo of `shift` bits and a
// few-bit quotient hi.  Every column's lo part is accumulated on its own (sums below p/2 by the choice
// of shift).  The hi parts are tiny, so those of two columns share one accumulator, packed as
// hi_even + 2^pack_shift * hi_odd: their two exact sums are separated again after the inverse
// transform, and each column's result is lo_sum + (hi_sum << shift) modulo 2^32.  A step therefore
// costs (K+1) L forward transforms, (K+1) + ceil((K+1)/2) inverse transforms and as many pointwise
// products per digit polynomial (3 + 3 for K = 1, against the 2 + 2 of a modulus wide enough for the
// whole sum and the 4 + 4 of two full halves).  Same reference semantics as kernels.hip (reference
// src/trgsw.hpp:131-182).
#include "kernels.hpp"
#include "ntt_wg.hpp"

namespace tfhe_amd {

// signed split v = hi * 2^shift + lo, lo in [-2^(shift-1), 2^(shift-1))
__device__ __forceinline__ int32_t split_lo(int32_t v, int shift) { return (int32_t)((uint32_t)v << (32 - shift)) >> (32 - shift); }
__device__ __forceinline__ int32_t split_hi(int32_t v, int shift) { return (int32_t)(((int64_t)v - split_lo(v, shift)) >> shift); }

// Transformed key layout: [n][R][TILES] polynomials with TILES = K1 + ceil(K1 / 2): tile c < K1 is the
// lo part of column c, tile K1 + j the packed hi parts of columns 2j and 2j + 1.  One workgroup per
// (row, tile).  Stored value: NTT(tile) / N, centred, in the order [pair kk in 0..3][thread t][2] of
// the transform's output layout (position 8t + k).
template <int LOGN>
__global__ void __launch_bounds__(WgNtt<LOGN>::T)
bk_transform_wg_kernel(const NttUniform u, const d2* __restrict__ table, const uint32_t* __restrict__ bk,
                       d2* __restrict__ bk_ntt, int K1, int shift, int pack_shift, int pack) {
    using W = WgNtt<LOGN>;
    __shared__ double s_tr[W::BUF_DOUBLES];
    const int t = threadIdx.x;
    const int tiles = K1 + (K1 + pack - 1) / pack;
    const size_t row = blockIdx.x / tiles;
    const int tile = blockIdx.x % tiles;
    const uint32_t* src = bk + row * K1 * W::N;
    double x[8];
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const int j = k * W::T + t;
        if (tile < K1) {
            x[k] = (double)split_lo((int32_t)src[(size_t)tile * W::N + j], shift);
        } else {
            const int c0 = pack * (tile - K1), c1 = c0 + 1;
            double v = (double)split_hi((int32_t)src[(size_t)c0 * W::N + j], shift);
            if (pack == 2 && c1 < K1)
                v += (double)split_hi((int32_t)src[(size_t)c1 * W::N + j], shift) * (double)(1 << pack_shift);
            x[k] = v;  // |v| <= 2^(32 - shift) (1 + 2^pack_shift) < 2^31
        }
    }
    int parity = 0;
    W::template forward<false>(x, u, table, s_tr, parity, t);
    d2* dst = bk_ntt + ((size_t)blockIdx.x) * (W::N / 2);
#pragma unroll
    for (int kk = 0; kk < 4; kk++) {
        d2 v;
        v.x = reduce(mulmod_tw(x[2 * kk], u.ninv, u.ninvp, u.p), u.p, u.pinv);
        v.y = reduce(mulmod_tw(x[2 * kk + 1], u.ninv, u.ninvp, u.p), u.p, u.pinv);
        dst[kk * W::T + t] = v;
    }
}

// Two workgroups per CU (<= 256 registers per lane, 71 KB of LDS each): while one waits at a
// transpose barrier the other one computes.  Measured alternatives: one workgroup per CU with
// the L digit transforms of a polynomial interleaved in each wave (more parallelism per wave,
// fewer barriers) is 20 % slower, three workgroups per CU spill and are 35 % slower.
#ifndef TFHE_AMD_WG_LDS_ROWS
#define TFHE_AMD_WG_LDS_ROWS -1  // experiment switch: 0, 1 or 2 forces the level for every instantiation
#endif
template <int LOGN, int K, int L, int BG, int PACK>
__global__ void __launch_bounds__(WgNtt<LOGN>::T, 2)
blind_rotate_wg_kernel(const NttUniform u, const d2* __restrict__ table, const d2* __restrict__ bk_ntt,
                       const uint32_t* __restrict__ c, const uint32_t* __restrict__ tv, uint32_t* __restrict__ out,
                       int n, int shift, int pack_shift, int extract) {
    using W = WgNtt<LOGN>;
    constexpr int N = W::N, T = W::T, K1 = K + 1, R = K1 * L;
    constexpr int PACKS = (K1 + PACK - 1) / PACK, TILES = K1 + PACKS, SMALL = BG == 8 ? 1 : 2;
    static_assert((BG == 8 || BG == 16) && L * BG <= 31 && (PACK == 1 || PACK == 2), "gadget shape");
    // twiddle rows in LDS as far as two workgroups per CU leave room (80 KB each)
    constexpr int ROWS_IN_LDS = TFHE_AMD_WG_LDS_ROWS < 0 ? (LOGN == 11 ? (K1 == 2 ? 2 : 1) : 0) : TFHE_AMD_WG_LDS_ROWS;
    __shared__ double s_tr[W::BUF_DOUBLES];
    __shared__ uint32_t s_acc[K1][N];
    __shared__ d2 s_rows[W::lds_row_pairs(ROWS_IN_LDS) > 0 ? W::lds_row_pairs(ROWS_IN_LDS) : 1];
    const int t = threadIdx.x;
    const size_t gate = blockIdx.x;
    const uint32_t* cg = c + gate * (size_t)(n + 1);
    const double p = u.p, pinv = u.pinv;
    constexpr int SH = 32 - (LOGN + 1);
    W::template stage_rows<ROWS_IN_LDS>(s_rows, table, t);

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
    const double unpack = 1.0 / (double)(1u << pack_shift), pack = (double)(1u << pack_shift);

    // -DTFHE_AMD_WG_PHASE_CLOCKS=1: s_memtime stamps around the phases of a step; every wave of gate 0
    // prints its totals at the end (a diagnostic build: read the shares, not the run time)
#ifdef TFHE_AMD_WG_PHASE_CLOCKS
    unsigned long long ph[6] = {0, 0, 0, 0, 0, 0}, tprev = 0;
#define WPH_START() tprev = __builtin_amdgcn_s_memtime()
#define WPH_MARK(I) { const unsigned long long tn_ = __builtin_amdgcn_s_memtime(); ph[I] += tn_ - tprev; tprev = tn_; }
#define WPH_KEEP(V) asm volatile("" :: "v"(V))
#else
#define WPH_START()
#define WPH_MARK(I)
#define WPH_KEEP(V)
#endif
    int parity = 0;
    for (int i = 0; i < n; i++) {
        WPH_START();
        const int abar = (int)((cg[1 + i] + (1u << (SH - 1))) >> SH);

        double acc_ntt[TILES][8];
#pragma unroll
        for (int tile = 0; tile < TILES; tile++)
#pragma unroll
            for (int k = 0; k < 8; k++) acc_ntt[tile][k] = 0.0;

        // experiment 8 (wrong results): eight key rows in rotation, which stay in the XCD's L2
        const d2* bk_row = bk_ntt + (size_t)(TFHE_AMD_WG_EXPERIMENT == 8 ? (i & 7) : i) * (R * TILES) * (N / 2);
#pragma unroll 1
        for (int q = 0; q < K1; q++) {
            uint32_t tw_[8];
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const int sidx = (k * T + t - abar) & (2 * N - 1);
                const uint32_t neg = 0u - (uint32_t)((sidx >> LOGN) & 1);
                const uint32_t v = s_acc[q][TFHE_AMD_WG_EXPERIMENT == 6 ? k * T + t : sidx & (N - 1)];
                tw_[k] = ((((v ^ neg) - neg) - s_acc[q][k * T + t]) + offset) ^ flip;
            }
            // The L digit polynomials of this accumulator polynomial are transformed group by group
            // together (one load of every twiddle row serves all of them); then each is multiplied
            // with its key row: L * TILES tiles of 4 pairs per thread, tile 0 requested before the
            // transforms, every further tile while the previous one is being multiplied.
            const d2* bk_poly = bk_row + (size_t)(q * L) * TILES * (N / 2);
            d2 y[4], ynext[4];
#pragma unroll
            for (int kk = 0; kk < 4; kk++) y[kk] = bk_poly[kk * T + t];
            double x[L][8];
#pragma unroll
            for (int level = 0; level < L; level++)
#pragma unroll
                for (int k = 0; k < 8; k++) x[level][k] = (double)(((int32_t)(tw_[k] << (BG * level))) >> (32 - BG));
            WPH_KEEP(x[0][0]); WPH_KEEP(x[L - 1][7]);
            WPH_MARK(0)
            W::template forward_many<SMALL, L, ROWS_IN_LDS>(x, u, table, s_tr, parity, t, s_rows);  // |x| <= 3.2 p
            WPH_KEEP(x[0][0]); WPH_KEEP(x[L - 1][7]);
            WPH_MARK(1)
#pragma unroll
            for (int level = 0; level < L; level++) {
#pragma unroll
                for (int tile = 0; tile < TILES; tile++) {
                    const int next = level * TILES + tile + 1;
                    if (next < L * TILES) {
#pragma unroll
                        for (int kk = 0; kk < 4; kk++)
                            ynext[kk] = bk_poly[(size_t)(TFHE_AMD_WG_EXPERIMENT == 5 ? 0 : next) * (N / 2) + kk * T + t];
                    }
                    // the eight products of a tile phase by phase (ntt512.hpp): dependent FP64
                    // instructions end up eight apart instead of back to back
                    double yy[8], prod[8];
#pragma unroll
                    for (int kk = 0; kk < 4; kk++) {
                        yy[2 * kk] = y[kk].x;
                        yy[2 * kk + 1] = y[kk].y;
                    }
                    mulmod_var_n<8>(prod, x[level], yy, p, pinv);
#pragma unroll
                    for (int k = 0; k < 8; k++) acc_ntt[tile][k] += prod[k];
                    if (next < L * TILES) {
#pragma unroll
                        for (int kk = 0; kk < 4; kk++) y[kk] = ynext[kk];
                    }
                }
                // products are <= 1.41 p each: reduce after every third digit polynomial (running sum < 4.74 p)
                if ((q * L + level) % 3 == 2 && q * L + level != R - 1) {
#pragma unroll
                    for (int tile = 0; tile < TILES; tile++) reduce_n<8>(acc_ntt[tile], p, pinv);
                }
            }
            WPH_KEEP(acc_ntt[0][0]); WPH_KEEP(acc_ntt[TILES - 1][7]);
            WPH_MARK(2)
        }
        // all K1 + PACKS inverse transforms group by group: one load of every twiddle row serves them all
        W::template inverse_many<TILES, ROWS_IN_LDS>(acc_ntt, u, table, s_tr, parity, t, s_rows);
        WPH_KEEP(acc_ntt[0][0]); WPH_KEEP(acc_ntt[TILES - 1][7]);
        WPH_MARK(3)
#pragma unroll
        for (int col = 0; col < K1; col++) {
            // the packed accumulator holds the small exact hi sums of two columns: h_even + 2^pack_shift h_odd
            uint32_t hi_sum[8];
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const double v = reduce(acc_ntt[K1 + col / PACK][k], p, pinv);
                if (PACK == 1) {
                    hi_sum[k] = (uint32_t)__double2loint(v + 6755399441055744.0);  // low word of the exact integer
                } else {
                    const double h1 = __builtin_rint(v * unpack);
                    hi_sum[k] = (uint32_t)(int32_t)((col & 1) ? h1 : __builtin_fma(-h1, pack, v));
                }
            }
#pragma unroll
            for (int k = 0; k < 8; k++)
                s_acc[col][k * T + t] += to_torus(acc_ntt[col][k], p, pinv) + (hi_sum[k] << shift);
        }
        WPH_MARK(4)
        __syncthreads();
        WPH_MARK(5)
    }
#ifdef TFHE_AMD_WG_PHASE_CLOCKS
    if (gate == 0 && (t & 63) == 0)
        printf("wave %d ticks: prep %llu forward %llu products %llu inverse %llu update %llu barrier %llu\n", t >> 6, ph[0],
               ph[1], ph[2], ph[3], ph[4], ph[5]);
#endif

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

// TRGSW::external_product / TRGSW::cmux on explicit operands (reference src/trgsw.hpp:100-117) for the
// parameter sets of this file: one workgroup per batch element, the step body of the blind rotation
// without the rotation.  key_ntt: the batch's TRGSW samples transformed and split by
// bk_transform_wg_kernel (element e = "row" e of a bootstrapping key).  mode 0: out = C (x) x0;
// mode 1: out = C (x) (x1 - x0) + x0.
template <int LOGN, int K, int L, int BG, int PACK>
__global__ void __launch_bounds__(WgNtt<LOGN>::T, 2)
external_product_wg_kernel(const NttUniform u, const d2* __restrict__ table, const d2* __restrict__ key_ntt,
                           const uint32_t* __restrict__ x0, const uint32_t* __restrict__ x1, uint32_t* __restrict__ out,
                           int shift, int pack_shift, int mode) {
    using W = WgNtt<LOGN>;
    constexpr int N = W::N, T = W::T, K1 = K + 1, R = K1 * L;
    constexpr int PACKS = (K1 + PACK - 1) / PACK, TILES = K1 + PACKS, SMALL = BG == 8 ? 1 : 2;
    static_assert((BG == 8 || BG == 16) && L * BG <= 31 && (PACK == 1 || PACK == 2), "gadget shape");
    __shared__ double s_tr[W::BUF_DOUBLES];
    const int t = threadIdx.x;
    const size_t e = blockIdx.x;
    const double p = u.p, pinv = u.pinv;
    uint32_t offset = 1u << (32 - L * BG - 1), flip = 0u;
#pragma unroll
    for (int l = 0; l < L; l++) {
        offset += 1u << (32 - BG * (l + 1) + BG - 1);
        flip |= 1u << (32 - BG * (l + 1) + BG - 1);
    }
    const double unpack = 1.0 / (double)(1u << pack_shift), pack = (double)(1u << pack_shift);
    const uint32_t* a0 = x0 + e * (size_t)(K1 * N);
    const uint32_t* a1 = x1 + e * (size_t)(K1 * N);
    const d2* key_row = key_ntt + e * (size_t)(R * TILES) * (N / 2);

    int parity = 0;
    double acc_ntt[TILES][8];
#pragma unroll
    for (int tile = 0; tile < TILES; tile++)
#pragma unroll
        for (int k = 0; k < 8; k++) acc_ntt[tile][k] = 0.0;
#pragma unroll 1
    for (int q = 0; q < K1; q++) {
        uint32_t tw_[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int j = q * N + k * T + t;
            const uint32_t v = mode ? a1[j] - a0[j] : a0[j];
            tw_[k] = (v + offset) ^ flip;
        }
        double x[L][8];
#pragma unroll
        for (int level = 0; level < L; level++)
#pragma unroll
            for (int k = 0; k < 8; k++) x[level][k] = (double)(((int32_t)(tw_[k] << (BG * level))) >> (32 - BG));
        W::template forward_many<SMALL, L>(x, u, table, s_tr, parity, t);  // |x| <= 3.2 p
        const d2* key_poly = key_row + (size_t)(q * L) * TILES * (N / 2);
#pragma unroll
        for (int level = 0; level < L; level++) {
#pragma unroll
            for (int tile = 0; tile < TILES; tile++) {
                double yy[8], prod[8];
#pragma unroll
                for (int kk = 0; kk < 4; kk++) {
                    const d2 y = key_poly[(size_t)(level * TILES + tile) * (N / 2) + kk * T + t];
                    yy[2 * kk] = y.x;
                    yy[2 * kk + 1] = y.y;
                }
                mulmod_var_n<8>(prod, x[level], yy, p, pinv);
#pragma unroll
                for (int k = 0; k < 8; k++) acc_ntt[tile][k] += prod[k];
            }
            if ((q * L + level) % 3 == 2 && q * L + level != R - 1) {
#pragma unroll
                for (int tile = 0; tile < TILES; tile++) reduce_n<8>(acc_ntt[tile], p, pinv);
            }
        }
    }
    W::template inverse_many<TILES>(acc_ntt, u, table, s_tr, parity, t);
    uint32_t* o = out + e * (size_t)(K1 * N);
#pragma unroll
    for (int col = 0; col < K1; col++) {
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const double v = reduce(acc_ntt[K1 + col / PACK][k], p, pinv);
            uint32_t hi;
            if (PACK == 1) {
                hi = (uint32_t)__double2loint(v + 6755399441055744.0);  // low word of the exact integer
            } else {
                const double h1 = __builtin_rint(v * unpack);
                hi = (uint32_t)(int32_t)((col & 1) ? h1 : __builtin_fma(-h1, pack, v));
            }
            const int j = col * N + k * T + t;
            o[j] = to_torus(acc_ntt[col][k], p, pinv) + (hi << shift) + (mode ? a0[j] : 0u);
        }
    }
}

// ---- host side ---------------------------------------------------------------------------------

// parameter sets with an instantiation: N = 2048 and N = 1024 with K = 1 or 2 and N = 512 with K = 1 or 2,
// for the 8-bit gadget with L = 2 or 3 (except N = 512, L = 2, which has wave-per-gate kernels in
// kernels.hip) and for the 16-bit gadget with L = 1
bool wg_has_kernel(int N, int K, int L, int bg) {
    if (K != 1 && K != 2) return false;
    if (N != 512 && N != 1024 && N != 2048) return false;
    if (bg == 16) return L == 1;
    if (bg != 8 || (L != 2 && L != 3)) return false;
    return !(N == 512 && L == 2);
}

int wg_table_pairs(int N) {
    return N == 2048 ? WgNtt<11>::TABLE_PAIRS : N == 1024 ? WgNtt<10>::TABLE_PAIRS : N == 512 ? WgNtt<9>::TABLE_PAIRS : 0;
}

// Split of the key for a parameter set (see the top of this file): `shift` = bits of the signed
// remainder lo, the largest for which a column's lo sum (K+1) L N 2^(bg-1) 2^(shift-1) stays below p/2
// (worst case, so no further margin is needed); one hi sum is then at most
// H = (K+1) L N 2^(bg-1) 2^(32-shift-1).  `pack` = columns whose hi sums share an accumulator: 2
// where H (1 + 2^pack_shift) stays below p/2 with `pack_shift` the smallest position that separates
// the two by rounding (2^(pack_shift-1) > H), else 1 (the 16-bit gadget: every column its own hi
// accumulator, H < p/2).  Returns false if no such split exists.  `tiles` = polynomials per key row
// after the split.
bool wg_split(int N, int K, int L, int bg, int* tiles, int* shift, int* pack_shift, int* pack) {
    const double pp = 1711528530149377.0;
    const double terms = (double)(K + 1) * L * N * (double)(1u << (bg - 1));
    for (int s = 30; s >= 8; s--) {
        if (terms * (double)(1u << (s - 1)) >= 0.49 * pp) continue;
        const double H = terms * (double)(1u << (32 - s - 1));
        if (H >= 0.49 * pp) return false;
        int ps = 1;
        while (ps <= 30 && (double)(1u << (ps - 1)) <= H) ps++;
        const bool two = ps <= 30 && H * (1.0 + (double)(1u << ps)) < 0.49 * pp;
        *pack = two ? 2 : 1;
        *pack_shift = two ? ps : 0;
        *tiles = (K + 1) + (K + *pack) / *pack;
        *shift = s;
        return true;
    }
    return false;
}

hipError_t launch_bk_transform_wg(const NttUniform& u, const d2* table, const uint32_t* bk, d2* bk_ntt, size_t n_rows,
                                  int N, int K, int shift, int pack_shift, int pack, hipStream_t stream) {
    if (n_rows == 0) return hipSuccess;
    if (pack != 1 && pack != 2) return hipErrorInvalidValue;
    const int tiles = (K + 1) + (K + pack) / pack;
#define WG_KEY(LOGN)                                                                                               \
    bk_transform_wg_kernel<LOGN><<<(unsigned)(n_rows * tiles), WgNtt<LOGN>::T, 0, stream>>>(u, table, bk, bk_ntt, K + 1, \
                                                                                           shift, pack_shift, pack)
    if (N == 2048) WG_KEY(11);
    else if (N == 1024) WG_KEY(10);
    else if (N == 512) WG_KEY(9);
    else return hipErrorInvalidValue;
#undef WG_KEY
    return hipGetLastError();
}

// instantiation table shared by the blind rotation and the stand-alone external product:
// GO(LOGN, K, L, BG, PACK) for the parameter set (N, K, L, bg) whose split packs `pack` columns
#define TFHE_AMD_WG_DISPATCH(GO)                                                                   \
    if (bg == 8 && pack == 2) {                                                                    \
        if (N == 2048 && K == 1) { if (L == 3) GO(11, 1, 3, 8, 2); else GO(11, 1, 2, 8, 2); }      \
        else if (N == 2048) { if (L == 3) GO(11, 2, 3, 8, 2); else GO(11, 2, 2, 8, 2); }           \
        else if (N == 1024 && K == 1) { if (L == 3) GO(10, 1, 3, 8, 2); else GO(10, 1, 2, 8, 2); } \
        else if (N == 1024) { if (L == 3) GO(10, 2, 3, 8, 2); else GO(10, 2, 2, 8, 2); }           \
        else if (K == 1) GO(9, 1, 3, 8, 2);                                                        \
        else GO(9, 2, 3, 8, 2);                                                                    \
    } else if (bg == 16 && pack == 1) {                                                            \
        if (N == 2048) { if (K == 1) GO(11, 1, 1, 16, 1); else GO(11, 2, 1, 16, 1); }              \
        else if (N == 1024) { if (K == 1) GO(10, 1, 1, 16, 1); else GO(10, 2, 1, 16, 1); }         \
        else { if (K == 1) GO(9, 1, 1, 16, 1); else GO(9, 2, 1, 16, 1); }                          \
    } else {                                                                                       \
        return hipErrorInvalidValue;                                                               \
    }

// the key must have been split the way the parameter set requires (the kernels assume it)
static bool wg_split_matches(int N, int K, int L, int bg, int shift, int pack_shift, int pack) {
    int t = 0, s = 0, ps = 0, pk = 0;
    return wg_has_kernel(N, K, L, bg) && wg_split(N, K, L, bg, &t, &s, &ps, &pk) && s == shift && ps == pack_shift &&
           pk == pack;
}

hipError_t launch_external_product_wg(const NttUniform& u, const d2* table, const d2* key_ntt, const uint32_t* x0,
                                      const uint32_t* x1, uint32_t* out, size_t batch, int N, int K, int L, int bg,
                                      int shift, int pack_shift, int pack, int mode, hipStream_t stream) {
    if (batch == 0) return hipSuccess;
    if (!wg_split_matches(N, K, L, bg, shift, pack_shift, pack)) return hipErrorInvalidValue;
#define WG_EP(LOGN, KK, LL, BB, PP)                                                                                  \
    external_product_wg_kernel<LOGN, KK, LL, BB, PP><<<(unsigned)batch, WgNtt<LOGN>::T, 0, stream>>>(                \
        u, table, key_ntt, x0, x1, out, shift, pack_shift, mode)
    TFHE_AMD_WG_DISPATCH(WG_EP)
#undef WG_EP
    return hipGetLastError();
}

hipError_t launch_blind_rotate_wg(const NttUniform& u, const d2* table, const d2* bk_ntt, const uint32_t* c,
                                  const uint32_t* tv, uint32_t* out, size_t n_gates, int n, int N, int K, int L, int bg,
                                  int shift, int pack_shift, int pack, int extract, hipStream_t stream) {
    if (n_gates == 0) return hipSuccess;
    if (!wg_split_matches(N, K, L, bg, shift, pack_shift, pack)) return hipErrorInvalidValue;
#define WG_GO(LOGN, KK, LL, BB, PP)                                                                                  \
    blind_rotate_wg_kernel<LOGN, KK, LL, BB, PP><<<(unsigned)n_gates, WgNtt<LOGN>::T, 0, stream>>>(                  \
        u, table, bk_ntt, c, tv, out, n, shift, pack_shift, extract)
    TFHE_AMD_WG_DISPATCH(WG_GO)
#undef WG_GO
    return hipGetLastError();
}

}  // namespace tfhe_amd
