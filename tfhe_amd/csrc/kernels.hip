// gfx950 kernels of the gate-bootstrapping path and their launchers.
//
//   nand_combine_kernel     c = (1/8, 0..) - (a + b)                 reference src/hom_nand.hpp:76-78
//   bk_transform_512_kernel bootstrapping key -> NTT domain (once per key)
//   blind_rotate_512_kernel the whole blind rotation + sample extraction of one gate per wave
//                                                                    reference src/trgsw.hpp:131-182,
//                                                                    src/trlwe.hpp:109-122
//   key_switch_kernel       identity key switch, a tile of gates per workgroup
//   key_switch_small_kernel the same for a handful of gates, one workgroup per (gate, slice)
//                                                                    reference src/tlwe_key_switching.hpp:47-65
#include <algorithm>

#include "kernels.hpp"
#include "ntt512.hpp"
#include "ntt512_half.hpp"

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
// Linear step of the other two-input gates: c = (offset, 0, ..) + sa * a + sb * b, the
// generalisation of the NAND line (reference src/hom_nand.hpp:76-78; the reference has no other
// gate).  true = +1/8, false = -1/8 (reference src/types.hpp:25,34).  Codes: tfhe_amd.h.
// `ops` (one code per gate, device memory) overrides `op` when not null; codes are validated on
// the host for `op` and clamped here for `ops` (an unknown code evaluates as NAND).
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void gate_coefficients(int op, uint32_t& offset, uint32_t& sa, uint32_t& sb) {
    const uint32_t mu = 1u << 29, q = 1u << 30, m1 = 0xffffffffu, m2 = 0xfffffffeu;
    switch (op) {
    default: offset = mu;      sa = m1; sb = m1; break;  // NAND
    case 1:  offset = 0u - mu; sa = 1;  sb = 1;  break;  // AND
    case 2:  offset = mu;      sa = 1;  sb = 1;  break;  // OR
    case 3:  offset = 0u - mu; sa = m1; sb = m1; break;  // NOR
    case 4:  offset = q;       sa = 2;  sb = 2;  break;  // XOR
    case 5:  offset = 0u - q;  sa = m2; sb = m2; break;  // XNOR
    case 6:  offset = 0u - mu; sa = m1; sb = 1;  break;  // ANDNY
    case 7:  offset = 0u - mu; sa = 1;  sb = m1; break;  // ANDYN
    case 8:  offset = mu;      sa = m1; sb = 1;  break;  // ORNY
    case 9:  offset = mu;      sa = 1;  sb = m1; break;  // ORYN
    }
}

__global__ void __launch_bounds__(256) gate_combine_kernel(uint32_t* __restrict__ c, const uint32_t* __restrict__ a,
                                                           const uint32_t* __restrict__ b,
                                                           const uint8_t* __restrict__ ops, int op, size_t total,
                                                           int n1) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const size_t g = i / (size_t)n1;
    uint32_t offset, sa, sb;
    gate_coefficients(ops ? (int)ops[g] : op, offset, sa, sb);
    c[i] = sa * a[i] + sb * b[i] + (i - g * (size_t)n1 == 0 ? offset : 0u);
}

// out = -a: homomorphic NOT, no bootstrap
__global__ void __launch_bounds__(256) negate_kernel(uint32_t* __restrict__ out, const uint32_t* __restrict__ a,
                                                     size_t total) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < total) out[i] = 0u - a[i];
}

// MUX, lv1 stage: dst[j] = dst[j] + src[j] + (1/8, 0, ..) over `total` words in rows of n1 words
__global__ void __launch_bounds__(256) mux_sum_kernel(uint32_t* __restrict__ dst, const uint32_t* __restrict__ src,
                                                      size_t total, int n1) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    dst[i] = dst[i] + src[i] + ((i % (size_t)n1) == 0 ? (1u << 29) : 0u);
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
    ntt512_forward<false, true>(x, u, s_tw, s_tr[wave], lane);
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
#ifndef TFHE_AMD_PIN_THROUGHPUT
#define TFHE_AMD_PIN_THROUGHPUT 1
#endif
// Phase-ordered butterflies (ntt512.hpp) in the throughput form.  Measured on 8,192 gates:
// 47.6 -> 46.3 ms (two waves per SIMD still leave dependent-issue gaps).
constexpr bool kPinThroughput = TFHE_AMD_PIN_THROUGHPUT != 0;
#ifndef TFHE_AMD_PAIR_THROUGHPUT
#define TFHE_AMD_PAIR_THROUGHPUT 1
#endif
// Counter experiment (wrong results): 1 = the rotated accumulator reads of the throughput form
// become aligned ones, to see which share of the LDS bank conflicts they cause.
#ifndef TFHE_AMD_EXPERIMENT_ALIGNED_ACC
#define TFHE_AMD_EXPERIMENT_ALIGNED_ACC 0
#endif

template <int K, int L>
__global__ void __launch_bounds__(256, 2)
blind_rotate_512_kernel(const NttUniform u, const d2* __restrict__ tw, const d2* __restrict__ bk_ntt,
                        const uint32_t* __restrict__ c, const uint32_t* __restrict__ tv, uint32_t* __restrict__ out,
                        int n_gates, int n, int extract) {
    constexpr int K1 = K + 1, R = K1 * L, BG = 8;
    __shared__ d2 s_tw[kTwPairs];
    __shared__ double s_tr[4][TFHE_AMD_PAIR_THROUGHPUT ? 2 : 1][kTrPad];
    __shared__ uint32_t s_acc[4][K1][kN];

    for (int i = threadIdx.x; i < kTwPairs; i += 256) s_tw[i] = tw[i];
    __syncthreads();

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int gate = blockIdx.x * 4 + wave;
    if (gate >= n_gates) return;  // waves are independent from here on: no further block barrier

    double* tr = s_tr[wave][0];
    double* tr2 = s_tr[wave][TFHE_AMD_PAIR_THROUGHPUT ? 1 : 0];
    uint32_t(*acc)[kN] = s_acc[wave];
    const uint32_t* cg = c + (size_t)gate * (n + 1);
    const double p = u.p, pinv = u.pinv;

    // acc = testvector * X^(-floor(b * 2N)); tv == nullptr: the gate-bootstrapping test vector
    // (1/8, .., 1/8 | 0 | 0) of reference src/trgsw.hpp:177-179
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

    // offset = rounding half-unit below the last level + 2^(BG-1) at every level's position;
    // flip = the top bit of every level's field: digit_l = signed field l of ((a + offset) ^ flip)
    uint32_t offset = 1u << (32 - L * BG - 1), flip = 0u;
#pragma unroll
    for (int l = 0; l < L; l++) {
        offset += 1u << (32 - BG * (l + 1) + BG - 1);
        flip |= 1u << (32 - BG * (l + 1) + BG - 1);
    }

#ifdef TFHE_AMD_PHASE_CLOCKS
    unsigned long long ph[6] = {0, 0, 0, 0, 0, 0}, tprev = 0;
#define PH_START() tprev = __builtin_amdgcn_s_memtime()
#define PH_MARK(I) { const unsigned long long tn_ = __builtin_amdgcn_s_memtime(); ph[I] += tn_ - tprev; tprev = tn_; }
#else
#define PH_START()
#define PH_MARK(I)
#endif
    uint32_t cvals = 0;
    for (int i = 0; i < n; i++) {
        PH_START();
        if ((i & 63) == 0) cvals = (i + lane < n) ? cg[1 + i + lane] : 0u;
        const uint32_t ci = (uint32_t)__builtin_amdgcn_readlane((int)cvals, i & 63);
        const int abar = (int)((ci + (1u << 21)) >> 22);  // round(a_i * 2N), in [0, 2N]

        // acc * X^abar at position j = 64k + lane: coefficient (j - abar) mod 2N of the
        // antiperiodic extension, i.e. +-acc[(j - abar) mod N]
        int src[8];
        uint32_t neg[8];  // all ones where the source coefficient enters negated
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int sidx = (64 * k + lane - abar) & (2 * kN - 1);
            src[k] = sidx & (kN - 1);
            neg[k] = 0u - (uint32_t)((sidx >> 9) & 1);
        }

        double out[K1][8];
#pragma unroll
        for (int q = 0; q < K1; q++)
#pragma unroll
            for (int k = 0; k < 8; k++) out[q][k] = 0.0;

        const d2* bk_row = bk_ntt + (size_t)i * (R * K1 * (kN / 2));
#pragma unroll 1
        for (int q = 0; q < K1; q++) {
            // t = (acc_q * X^abar - acc_q + offset) ^ flip, all levels' digits at once
            uint32_t t[8];
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const uint32_t v = acc[q][TFHE_AMD_EXPERIMENT_ALIGNED_ACC ? 64 * k + lane : src[k]];
                t[k] = ((((v ^ neg[k]) - neg[k]) - acc[q][64 * k + lane]) + offset) ^ flip;
            }
#ifdef TFHE_AMD_PHASE_CLOCKS
            asm volatile("" :: "v"(t[0]), "v"(t[7]));
#endif
            PH_MARK(0)
#if TFHE_AMD_PAIR_THROUGHPUT
            {
                static_assert(L == 2 && (K1 == 2 || K1 == 3), "the two levels of a digit polynomial run as one interleaved pair");
                double x0[8], x1[8];
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    x0[k] = (double)(((int32_t)t[k]) >> (32 - BG));
                    x1[k] = (double)(((int32_t)(t[k] << BG)) >> (32 - BG));
                }
                const d2* bk_poly = bk_row + (size_t)(q * L) * K1 * (kN / 2);
                d2 y[K1][4];
#pragma unroll
                for (int col = 0; col < K1; col++)
#pragma unroll
                    for (int kk = 0; kk < 4; kk++) y[col][kk] = bk_poly[col * (kN / 2) + kk * 64 + lane];
                auto mac = [&](double (&x)[8]) {
#pragma unroll
                    for (int col = 0; col < K1; col++) {
                        double yy[8], prod[8];
#pragma unroll
                        for (int kk = 0; kk < 4; kk++) {
                            yy[2 * kk] = y[col][kk].x;
                            yy[2 * kk + 1] = y[col][kk].y;
                        }
                        mulmod_var_n<8>(prod, x, yy, p, pinv);
#pragma unroll
                        for (int k = 0; k < 8; k++) out[col][k] += prod[k];
                    }
                };
                ntt512_forward2_small(x0, x1, u, s_tw, tr, tr2, lane, [&]() {
                    mac(x0);
                    // every product is at most 1.41 p: three sum to <= 4.24 p; reducing after the third
                    // of the six (or four, K = 1) keeps the running sum below 0.5 p + 4.24 p = 4.74 p
                    // < 2^53 = 5.26 p
                    if (q == 1) {
#pragma unroll
                        for (int col = 0; col < K1; col++)
#pragma unroll
                            for (int k = 0; k < 8; k++) out[col][k] = reduce(out[col][k], p, pinv);
                    }
                    // level 1's key row replaces level 0's in the same registers; its latency hides
                    // under the last stage group of x1
#pragma unroll
                    for (int col = 0; col < K1; col++)
#pragma unroll
                        for (int kk = 0; kk < 4; kk++)
                            y[col][kk] = bk_poly[(K1 + col) * (kN / 2) + kk * 64 + lane];
                });
                PH_MARK(1)
                mac(x1);
                PH_MARK(2)
            }
#else
#pragma unroll
            for (int level = 0; level < L; level++) {
                double x[8];
#pragma unroll
                for (int k = 0; k < 8; k++)
                    x[k] = (double)(((int32_t)(t[k] << (BG * level))) >> (32 - BG));  // signed field `level`
                // the key row of this digit polynomial is requested before the transform and
                // consumed after it, so the L2 / Infinity-Cache latency hides under the butterflies
                const d2* bk_poly = bk_row + (size_t)(q * L + level) * K1 * (kN / 2);
                d2 y[K1][4];
#pragma unroll
                for (int col = 0; col < K1; col++)
#pragma unroll
                    for (int kk = 0; kk < 4; kk++) y[col][kk] = bk_poly[col * (kN / 2) + kk * 64 + lane];
                // |x| <= 3.2 p: every product below is at most (0.5 + 1.5 * 3.2 / 5.26) p = 1.41 p
                ntt512_forward<true, false, kPinThroughput>(x, u, s_tw, tr, lane);
#pragma unroll
                for (int col = 0; col < K1; col++) {
                    if (kPinThroughput) {
                        double yy[8], prod[8];
#pragma unroll
                        for (int kk = 0; kk < 4; kk++) {
                            yy[2 * kk] = y[col][kk].x;
                            yy[2 * kk + 1] = y[col][kk].y;
                        }
                        mulmod_var_n<8>(prod, x, yy, p, pinv);
#pragma unroll
                        for (int k = 0; k < 8; k++) out[col][k] += prod[k];
                        continue;
                    }
#pragma unroll
                    for (int kk = 0; kk < 4; kk++) {
                        out[col][2 * kk] += mulmod_var(x[2 * kk], y[col][kk].x, p, pinv);
                        out[col][2 * kk + 1] += mulmod_var(x[2 * kk + 1], y[col][kk].y, p, pinv);
                    }
                }
                // three products sum to <= 4.24 p; reducing after every third keeps the running
                // sum below 0.5 p + 4.24 p = 4.74 p < 2^53 = 5.26 p
                if ((q * L + level) % 3 == 2 && q * L + level != R - 1) {
#pragma unroll
                    for (int col = 0; col < K1; col++)
#pragma unroll
                        for (int k = 0; k < 8; k++) out[col][k] = reduce(out[col][k], p, pinv);
                }
            }
#endif
        }
        // |out| <= 4.74 p
#if TFHE_AMD_PAIR_THROUGHPUT
        ntt512_inverse2(out[0], out[1], u, s_tw, tr, tr2, lane);
        PH_MARK(3)
        auto update = [&](int col) {
#pragma unroll
            for (int k = 0; k < 8; k++) acc[col][64 * k + lane] += to_torus(out[col][k], p, pinv);
        };
        if constexpr (K1 == 3) {
            // the accumulator updates of the first two columns run while the third transform's
            // transpositions are in flight
            ntt512_inverse_overlapped(out[2], u, s_tw, tr, lane, [&]() { update(0); }, [&]() { update(1); });
            update(2);
        } else {
            update(0);
            update(1);
        }
#else
#pragma unroll
        for (int col = 0; col < K1; col++) {
            ntt512_inverse<kPinThroughput>(out[col], u, s_tw, tr, lane);
#pragma unroll
            for (int k = 0; k < 8; k++) acc[col][64 * k + lane] += to_torus(out[col][k], p, pinv);
        }
#endif
        wave_lds_fence();
        PH_MARK(4)
    }
#ifdef TFHE_AMD_PHASE_CLOCKS
    if (gate == 5 && lane == 0)
        printf("phase ticks gate 5: prep %llu fwd2+mac0 %llu mac1 %llu inv2 %llu inv3+update %llu\n", ph[0], ph[1], ph[2], ph[3], ph[4]);
#endif

    if (!extract) {  // raw accumulator [K1][N]
        uint32_t* o = out + (size_t)gate * (K1 * kN);
#pragma unroll
        for (int q = 0; q < K1; q++)
#pragma unroll
            for (int k = 0; k < 8; k++) o[q * kN + 64 * k + lane] = acc[q][64 * k + lane];
        return;
    }
    // sample extraction at index 0 (reference src/trlwe.hpp:109-122):
    // lv1[0] = b[0]; lv1[1 + q N] = a_q[0]; lv1[1 + q N + i] = -a_q[N - i]
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

// ------------------------------------------------------------------------------------------
// TRGSW::external_product / TRGSW::cmux on explicit operands (reference src/trgsw.hpp:100-117),
// default ring: one wave per batch element, the step body of the blind rotation without the
// rotation.  key_ntt: the batch's TRGSW samples transformed by bk_transform_512_kernel (element e =
// "row" e of a bootstrapping key).  mode 0: out = C (x) x0; mode 1: out = C (x) (x1 - x0) + x0.
// ------------------------------------------------------------------------------------------
template <int K, int L>
__global__ void __launch_bounds__(256, 2)
external_product_512_kernel(const NttUniform u, const d2* __restrict__ tw, const d2* __restrict__ key_ntt,
                            const uint32_t* __restrict__ x0, const uint32_t* __restrict__ x1, uint32_t* __restrict__ out,
                            int batch, int mode) {
    constexpr int K1 = K + 1, R = K1 * L, BG = 8;
    __shared__ d2 s_tw[kTwPairs];
    __shared__ double s_tr[4][kTrPad];
    for (int i = threadIdx.x; i < kTwPairs; i += 256) s_tw[i] = tw[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int e = blockIdx.x * 4 + wave;
    if (e >= batch) return;
    double* tr = s_tr[wave];
    const double p = u.p, pinv = u.pinv;
    uint32_t offset = 1u << (32 - L * BG - 1), flip = 0u;
#pragma unroll
    for (int l = 0; l < L; l++) {
        offset += 1u << (32 - BG * (l + 1) + BG - 1);
        flip |= 1u << (32 - BG * (l + 1) + BG - 1);
    }
    const uint32_t* a0 = x0 + (size_t)e * (K1 * kN);
    const uint32_t* a1 = x1 + (size_t)e * (K1 * kN);
    const d2* key_row = key_ntt + (size_t)e * (R * K1 * (kN / 2));

    double acc[K1][8];
#pragma unroll
    for (int col = 0; col < K1; col++)
#pragma unroll
        for (int k = 0; k < 8; k++) acc[col][k] = 0.0;
#pragma unroll 1
    for (int q = 0; q < K1; q++) {
        uint32_t t[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int j = q * kN + 64 * k + lane;
            const uint32_t v = mode ? a1[j] - a0[j] : a0[j];
            t[k] = (v + offset) ^ flip;
        }
#pragma unroll
        for (int level = 0; level < L; level++) {
            double x[8];
#pragma unroll
            for (int k = 0; k < 8; k++) x[k] = (double)(((int32_t)(t[k] << (BG * level))) >> (32 - BG));
            ntt512_forward<true, false, true>(x, u, s_tw, tr, lane);  // |x| <= 3.2 p
            const d2* key_poly = key_row + (size_t)(q * L + level) * K1 * (kN / 2);
#pragma unroll
            for (int col = 0; col < K1; col++) {
                double yy[8], prod[8];
#pragma unroll
                for (int kk = 0; kk < 4; kk++) {
                    const d2 y = key_poly[col * (kN / 2) + kk * 64 + lane];
                    yy[2 * kk] = y.x;
                    yy[2 * kk + 1] = y.y;
                }
                mulmod_var_n<8>(prod, x, yy, p, pinv);  // <= 1.41 p each
#pragma unroll
                for (int k = 0; k < 8; k++) acc[col][k] += prod[k];
            }
            // reduce after every third product: the running sum stays below 0.5 p + 3 * 1.41 p < 2^53
            if ((q * L + level) % 3 == 2 && q * L + level != R - 1) {
#pragma unroll
                for (int col = 0; col < K1; col++) reduce_n<8>(acc[col], p, pinv);
            }
        }
    }
    uint32_t* o = out + (size_t)e * (K1 * kN);
#pragma unroll
    for (int col = 0; col < K1; col++) {
        ntt512_inverse<true>(acc[col], u, s_tw, tr, lane);
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int j = col * kN + 64 * k + lane;
            o[j] = to_torus(acc[col][k], p, pinv) + (mode ? a0[j] : 0u);
        }
    }
}

// ------------------------------------------------------------------------------------------
// Blind rotation for small batches (latency form).  One WORKGROUP owns one gate; every forward
// transform is split over two waves by halves of the vector and every inverse transform by the parity
// of the position (ntt512_half.hpp), so a step is spread over 2 (K+1) L = 12 waves, three per SIMD:
//   phase 1: wave (r, h) decomposes digit polynomial r, transforms half h of it and multiplies
//            with half h of key row r.  Its lanes hold both inputs of two first-stage butterflies
//            of the inverse transform per column; that stage is linear and commutes with the
//            summation over r, so the wave stores the sum and the difference of each such pair of
//            partial products;
//   phase 2: wave (col, e) adds up the partial sums (e = 0) or differences (e = 1) of column col,
//            finishes the first stage (e = 1: one twiddle product), runs the other eight stages
//            alone and updates the accumulator coefficients of parity e.  No stage pairs the two
//            waves.
// Against one wave per gate (blind_rotate_512_kernel) the dependent chain of a step is one half
// forward transform, one multiply and one half inverse transform instead of six, eighteen and
// three whole ones, and all four SIMDs carry the same load in phase 1.  Two block barriers per step
// (a form whose inverse was split by halves of the vector needed a third phase for the stage
// that pairs the halves: 2.31 ms per gate against 2.26 ms; six waves, one per transform: 2.74 ms).
// Same arithmetic, same results.  `tw_half`: the table of ntt512_half.hpp.
// ------------------------------------------------------------------------------------------
// key rows requested this many steps ahead of their use (1 or 2)
#ifndef TFHE_AMD_LATENCY_KEY_AHEAD
#define TFHE_AMD_LATENCY_KEY_AHEAD 1
#endif
template <int K, int L>
__global__ void __launch_bounds__(128 * (K + 1) * L, 1)
blind_rotate_512_latency_kernel(const NttUniform u, const d2* __restrict__ tw_half, const d2* __restrict__ bk_ntt,
                               const uint32_t* __restrict__ c, const uint32_t* __restrict__ tv,
                               uint32_t* __restrict__ out, int n_gates, int n, int extract) {
    constexpr int K1 = K + 1, R = K1 * L, BG = 8, WAVES = 2 * R, THREADS = 64 * WAVES;
    constexpr int FWD = 3 * kHalfRows;  // forward rows of the table; the inverse part follows
    __shared__ d2 s_tw[kHalfPairs];
    __shared__ double s_tr[WAVES][kHalfTrWords];
    __shared__ uint32_t s_acc[K1][kN];
    __shared__ d2 s_part[R][K1][2][kPairSlots / 2];  // [digit polynomial][column][sum | difference][pair index i, padded]

    for (int i = threadIdx.x; i < kHalfPairs; i += THREADS) s_tw[i] = tw_half[i];
    const d2* s_first = s_tw + FWD;       // [4][64] first-stage twiddles of the inverse
    const d2* s_rows = s_tw + FWD + 256;  // rows of its other stages
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int r = w >> 1, h = w & 1;
    const int gate = blockIdx.x;
    const uint32_t* cg = c + (size_t)gate * (n + 1);
    const double p = u.p, pinv = u.pinv;
    double* tr = s_tr[w];

    {
        const int bbar = (int)(cg[0] >> 22);
        for (int e = threadIdx.x; e < K1 * kN; e += THREADS) {
            const int q = e / kN, j = e - q * kN;
            const int s = (j + bbar) & (2 * kN - 1);
            const uint32_t v = tv ? tv[q * kN + (s & (kN - 1))] : (q == 0 ? (1u << 29) : 0u);
            s_acc[q][j] = (s & kN) ? 0u - v : v;
        }
    }
    __syncthreads();

    uint32_t offset = 1u << (32 - L * BG - 1), flip = 0u;
#pragma unroll
    for (int l = 0; l < L; l++) {
        offset += 1u << (32 - BG * (l + 1) + BG - 1);
        flip |= 1u << (32 - BG * (l + 1) + BG - 1);
    }
    const int q = r / L, level = r % L;
    const unsigned digit_at = 32 - BG * (level + 1);  // lowest bit of this wave's digit

    // This wave's share of a key polynomial: spectrum positions 256 h + 4 lane + k, i.e. pairs
    // kk = 2 (lane & 1) + {0, 1} of lane 32 h + (lane >> 1) in the stored [4][64][2] order.
    const int key_off = (2 * (lane & 1)) * 64 + 32 * h + (lane >> 1);
    // Phase 1 holds positions 256 h + 4 lane + k, i.e. the first-stage pairs i = 128 h + 2 lane + {0, 1};
    // phase 2 wants i = 4 lane + k.
    const int put_at = pair_slot(128 * h + 2 * lane) / 2, get_at = pair_slot(4 * lane) / 2;
    d2 y[K1][2], ynext[K1][2];
#if TFHE_AMD_LATENCY_KEY_AHEAD == 2
    d2 ynext2[K1][2];
#endif
    {
        const d2* bk_poly = bk_ntt + (size_t)r * K1 * (kN / 2) + key_off;
#pragma unroll
        for (int col = 0; col < K1; col++)
#pragma unroll
            for (int t = 0; t < 2; t++) ynext[col][t] = bk_poly[col * (kN / 2) + t * 64];
#if TFHE_AMD_LATENCY_KEY_AHEAD == 2
        if (n > 1) {
            const d2* bk_poly1 = bk_poly + (size_t)R * K1 * (kN / 2);
#pragma unroll
            for (int col = 0; col < K1; col++)
#pragma unroll
                for (int t = 0; t < 2; t++) ynext2[col][t] = bk_poly1[col * (kN / 2) + t * 64];
        }
#endif
    }
#ifdef TFHE_AMD_LATENCY_CLOCKS
    unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev = 0;
#define LPH_START() tprev = __builtin_amdgcn_s_memtime()
#define LPH_MARK(I) { const unsigned long long tn_ = __builtin_amdgcn_s_memtime(); ph[I] += tn_ - tprev; tprev = tn_; }
#else
#define LPH_START()
#define LPH_MARK(I)
#endif
    uint32_t cvals = 0;
    for (int i = 0; i < n; i++) {
        LPH_START();
        if ((i & 63) == 0) cvals = (i + lane < n) ? cg[1 + i + lane] : 0u;
        const uint32_t ci = (uint32_t)__builtin_amdgcn_readlane((int)cvals, i & 63);
        const int abar = (int)((ci + (1u << 21)) >> 22);

        // ---- phase 1 ----
#pragma unroll
        for (int col = 0; col < K1; col++)
#pragma unroll
            for (int t = 0; t < 2; t++) y[col][t] = ynext[col][t];
#if TFHE_AMD_LATENCY_KEY_AHEAD == 2
#pragma unroll
        for (int col = 0; col < K1; col++)
#pragma unroll
            for (int t = 0; t < 2; t++) ynext[col][t] = ynext2[col][t];
        if (i + 2 < n) {
            const d2* bk_poly = bk_ntt + ((size_t)(i + 2) * R + r) * K1 * (kN / 2) + key_off;
#pragma unroll
            for (int col = 0; col < K1; col++)
#pragma unroll
                for (int t = 0; t < 2; t++) ynext2[col][t] = bk_poly[col * (kN / 2) + t * 64];
        }
#else
        if (i + 1 < n) {
            const d2* bk_poly = bk_ntt + ((size_t)(i + 1) * R + r) * K1 * (kN / 2) + key_off;
#pragma unroll
            for (int col = 0; col < K1; col++)
#pragma unroll
                for (int t = 0; t < 2; t++) ynext[col][t] = bk_poly[col * (kN / 2) + t * 64];
        }
#endif
        double d[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int j = 64 * k + lane;
            const int sidx = (j - abar) & (2 * kN - 1);
            const uint32_t neg = 0u - (uint32_t)((sidx >> 9) & 1);
            const uint32_t v = s_acc[q][sidx & (kN - 1)];
            const uint32_t t = ((((v ^ neg) - neg) - s_acc[q][j]) + offset) ^ flip;
            d[k] = (double)(int32_t)__builtin_amdgcn_sbfe(t, digit_at, BG);  // signed digit `level`, one instruction
        }
#ifdef TFHE_AMD_LATENCY_CLOCKS
        asm volatile("" :: "v"(d[0]), "v"(d[7]));
#endif
        LPH_MARK(0)
        double x[4];
        half_forward_small(x, d, h, u, s_tw, tr, lane);  // |x| <= 2.2 p
#ifdef TFHE_AMD_LATENCY_CLOCKS
        asm volatile("" :: "v"(x[0]), "v"(x[3]));
#endif
        LPH_MARK(1)
#pragma unroll
        for (int col = 0; col < K1; col++) {
            double pr[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const double yk = (k & 1) ? y[col][k >> 1].y : y[col][k >> 1].x;
                pr[k] = mulmod_var(x[k], yk, p, pinv);  // <= 1.13 p
            }
            s_part[r][col][0][put_at] = d2{pr[0] + pr[1], pr[2] + pr[3]};  // <= 2.26 p
            s_part[r][col][1][put_at] = d2{pr[0] - pr[1], pr[2] - pr[3]};
        }
        LPH_MARK(2)
        __syncthreads();
        LPH_MARK(3)

        // ---- phase 2: parity e = h of column r (waves with r <= K) ----
        if (r < K1) {
            const int e = h;
            uint32_t acc_old[4];
#pragma unroll
            for (int k = 0; k < 4; k++) acc_old[k] = s_acc[r][128 * k + 2 * lane + e];
            double x2[4];
#pragma unroll
            for (int k = 0; k < 4; k++) x2[k] = 0.0;
            static_assert(R % 2 == 0, "the sums below are reduced after every second term");
#pragma unroll
            for (int rr = 0; rr < R; rr++) {
                const d2 v0 = s_part[rr][r][e][get_at], v1 = s_part[rr][r][e][get_at + 1];
                x2[0] += v0.x;
                x2[1] += v0.y;
                x2[2] += v1.x;
                x2[3] += v1.y;
                // the running sum stays below 0.5 p + 2 * 2.26 p = 5.02 p < 2^53
                if ((rr & 1) && rr != R - 1) reduce_n<4>(x2, p, pinv);
            }
            if (e == 0) {
                reduce_n<4>(x2, p, pinv);
            } else {
                double df[4], ww[4], wwp[4];
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    df[k] = x2[k];
                    const d2 t1 = s_first[64 * k + lane];
                    ww[k] = t1.x;
                    wwp[k] = t1.y;
                }
                mulmod_tw_n<4>(x2, df, ww, wwp, p);  // <= 1.22 p
            }
#ifdef TFHE_AMD_LATENCY_CLOCKS
            asm volatile("" :: "v"(x2[0]), "v"(x2[3]));
#endif
            LPH_MARK(4)
            parity_inverse(x2, u, s_rows, tr, lane);  // <= 2.21 p, position 128 k + 2 lane + e
#ifdef TFHE_AMD_LATENCY_CLOCKS
            asm volatile("" :: "v"(x2[0]), "v"(x2[3]));
#endif
            LPH_MARK(5)
#pragma unroll
            for (int k = 0; k < 4; k++) s_acc[r][128 * k + 2 * lane + e] = acc_old[k] + to_torus(x2[k], p, pinv);
        }
        LPH_MARK(6)
        __syncthreads();
        LPH_MARK(7)
    }
#ifdef TFHE_AMD_LATENCY_CLOCKS
    if (gate == 0 && lane == 0)
        printf("wave %2d ticks: decompose %llu forward %llu products %llu barrier %llu | sum+stage1 %llu inverse %llu update %llu barrier %llu\n",
               w, ph[0], ph[1], ph[2], ph[3], ph[4], ph[5], ph[6], ph[7]);
#endif

    if (!extract) {
        uint32_t* o = out + (size_t)gate * (K1 * kN);
        for (int e = threadIdx.x; e < K1 * kN; e += THREADS) o[e] = s_acc[e / kN][e % kN];
        return;
    }
    uint32_t* o = out + (size_t)gate * (K * kN + 1);
    if (threadIdx.x == 0) o[0] = s_acc[0][0];
    for (int e = threadIdx.x; e < K * kN; e += THREADS) {
        const int qq = e / kN, j = e - qq * kN;
        const uint32_t v = s_acc[1 + qq][j];
        if (j == 0) o[1 + qq * kN] = v;
        else        o[1 + qq * kN + (kN - j)] = 0u - v;
    }
}

// ------------------------------------------------------------------------------------------
// Identity key switch.
//
// Table layout on the device ("packed"): [N*K][T][2^bb - 1][ROWP] words, the never-read digit-0
// rows dropped and every row padded to ROWP (a multiple of 320 words) so that rows are
// 16-byte aligned and the rows of one lv1 coefficient are one contiguous block.
//
// Work split: a workgroup (4 waves) owns a tile of 64 gates, a slab of 320 output words and
// one of `parts` equal slices of the lv1 coefficients.  Lane = gate; wave v accumulates words
// [80 v, 80 v + 80) of the slab for its 64 gates in registers.  The slab's share of the rows
// of MG consecutive levels m of one coefficient i is staged in LDS (double buffered, behind
// one all-zero row); per level every lane turns its own digit into a row address (digit 0 ->
// the zero row, which is what the reference's `if (o)` amounts to) and reads its 80 words with
// twenty 16-byte LDS reads.  Rows are 324 words apart in LDS so that the (at most 2^bb)
// different rows a 16-lane group touches fall on different banks.  There is no scalar
// selection and one LDS round trip per level.  With parts > 1 the slices are combined with
// integer atomic adds into the zero-initialised output; wrap-around addition is associative,
// so the result does not depend on arrival order.
// ------------------------------------------------------------------------------------------
constexpr int kKsThreads = 256;   // 4 waves, one per SIMD: five waves of 64 words leave the SIMD that gets two the bottleneck
constexpr int kKsWaveWords = 80;  // output words per wave
constexpr int kKsSlab = 320;      // output words per workgroup
constexpr int kKsRowLds = 324;    // LDS row stride in words: 320 + 4 (bank skew between rows)
constexpr int kKsChunk = 64;      // lv1 coefficients whose digits are resident at a time
constexpr int kKsTile = 64;       // gates per workgroup = lanes per wave

__global__ void __launch_bounds__(256)
ks_pack_kernel(uint32_t* __restrict__ packed, const uint32_t* __restrict__ ks, size_t rows_out, int nr1, int n1,
               int rowp) {
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= rows_out * (size_t)rowp) return;
    const size_t row = e / rowp;
    const int w = (int)(e - row * rowp);
    const size_t im = row / nr1;
    const int o = (int)(row - im * nr1) + 1;
    packed[e] = w < n1 ? ks[(im * (nr1 + 1) + o) * n1 + w] : 0u;
}

template <int NR1, int MG>
__global__ void __launch_bounds__(kKsThreads, 2)
key_switch_kernel(const uint32_t* __restrict__ lv1, const uint32_t* __restrict__ ks_packed,
                  uint32_t* __restrict__ out, int n_gates, int n1, int nlv1, int T, int bb, int rowp, int parts) {
    constexpr int SR = MG * NR1;                          // rows per stage
    constexpr int STAGE_V4 = SR * kKsSlab / 4;            // 16-byte vectors per stage
    constexpr int PRE = (STAGE_V4 + kKsThreads - 1) / kKsThreads;
    constexpr int BUF = SR * kKsRowLds;                   // words per stage buffer
    extern __shared__ uint32_t s_mem[];
    uint32_t* s_zero = s_mem;                             // [kKsRowLds] all-zero row
    uint32_t* s_rows = s_mem + kKsRowLds;                 // [2][SR][kKsRowLds]
    uint32_t* s_dig = s_rows + 2 * BUF;                   // [kKsChunk][kKsTile] digit words
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int slabs = rowp / kKsSlab;
    int b = blockIdx.x;
    const int part = b % parts; b /= parts;
    const int slab = b % slabs; b /= slabs;
    const int g0 = b * kKsTile;
    const int i_begin = (int)((long)nlv1 * part / parts), i_end = (int)((long)nlv1 * (part + 1) / parts);
    const uint32_t round_offset = 1u << (32 - T * bb - 1);
    const int dshift = 32 - T * bb;
    const uint32_t omask = (1u << bb) - 1;

    static_assert(kKsWaveWords * (kKsThreads / 64) == kKsSlab && kKsWaveWords % 40 == 0, "slab split over the waves");
    uint32_t acc[kKsWaveWords];
#pragma unroll
    for (int k = 0; k < kKsWaveWords; k++) acc[k] = 0u;
    for (int w = tid; w < kKsRowLds; w += kKsThreads) s_zero[w] = 0u;

    const int stages_per_i = T / MG;
    const int total = (i_end - i_begin) * stages_per_i;
    static_assert(PRE <= 5, "stage too large for the prefetch registers");
    uint4 pre0 = {}, pre1 = {}, pre2 = {}, pre3 = {}, pre4 = {};
    // threads past the end of the stage re-read its last vector (never stored)
#define KS_LOAD1(Q, DST)                                                                       \
    if (Q < PRE) {                                                                             \
        const int e_ = min(tid + Q * kKsThreads, STAGE_V4 - 1);                                \
        const int r_ = e_ / (kKsSlab / 4), c_ = e_ - r_ * (kKsSlab / 4);                       \
        DST = *reinterpret_cast<const uint4*>(src_ + (size_t)r_ * rowp + 4 * c_);              \
    }
#define KS_PREFETCH(ST)                                                                        \
    {                                                                                          \
        const int pi_ = i_begin + (ST) / stages_per_i, pm_ = ((ST) % stages_per_i) * MG;       \
        const uint32_t* src_ = ks_packed + ((size_t)pi_ * T + pm_) * NR1 * rowp + slab * kKsSlab; \
        KS_LOAD1(0, pre0) KS_LOAD1(1, pre1) KS_LOAD1(2, pre2) KS_LOAD1(3, pre3) KS_LOAD1(4, pre4) \
    }
#define KS_STORE1(Q, SRC)                                                                      \
    if (Q < PRE) {                                                                             \
        const int e_ = tid + Q * kKsThreads;                                                   \
        if (e_ < STAGE_V4) {                                                                   \
            const int r_ = e_ / (kKsSlab / 4), c_ = e_ - r_ * (kKsSlab / 4);                   \
            *reinterpret_cast<uint4*>(buf + r_ * kKsRowLds + 4 * c_) = SRC;                    \
        }                                                                                      \
    }

    if (total > 0) KS_PREFETCH(0)
    for (int st = 0; st < total; st++) {
        const int irel = st / stages_per_i, m0 = (st - irel * stages_per_i) * MG;
        if (m0 == 0 && (irel % kKsChunk) == 0) {
            __syncthreads();  // the previous chunk's digits are no longer read
            for (int e = tid; e < kKsChunk * kKsTile; e += kKsThreads) {
                const int ii = e % kKsChunk, g = e / kKsChunk;
                const int i = i_begin + irel + ii;
                uint32_t v = 0u;
                if (g0 + g < n_gates && i < i_end) v = lv1[(size_t)(g0 + g) * (nlv1 + 1) + 1 + i];
                s_dig[ii * kKsTile + g] = (v + round_offset) >> dshift;
            }
        }
        uint32_t* buf = s_rows + (st & 1) * BUF;
        KS_STORE1(0, pre0) KS_STORE1(1, pre1) KS_STORE1(2, pre2) KS_STORE1(3, pre3) KS_STORE1(4, pre4)
        __syncthreads();  // stage st is in LDS; all reads of stage st-1 precede the next write of its buffer
        if (st + 1 < total) KS_PREFETCH(st + 1)

        const uint32_t d = s_dig[(irel % kKsChunk) * kKsTile + lane];
        // byte offsets from s_zero: this wave's first word of the zero row, or of row (mm*NR1 + o - 1) of buf
        const int zero_off = 4 * kKsWaveWords * wave;
        const int buf_off = 4 * ((int)(buf - s_zero) + kKsWaveWords * wave);
        const char* base = reinterpret_cast<const char*>(s_zero);
#pragma unroll
        for (int mm = 0; mm < MG; mm++) {
            const uint32_t o = (d >> ((T - 1 - (m0 + mm)) * bb)) & omask;
            const int off = o ? buf_off + 4 * kKsRowLds * (mm * NR1 + (int)o - 1) : zero_off;
            const uint4* row = reinterpret_cast<const uint4*>(base + off);
            // two batches of ten 16-byte reads; the scheduling barriers keep the compiler from
            // hoisting later batches' reads (and their 40 registers each) above this one
#pragma unroll
            for (int h = 0; h < 2; h++) {
                constexpr int HW = kKsWaveWords / 2, HV = HW / 4;
                uint4 v[HV];
#pragma unroll
                for (int q = 0; q < HV; q++) v[q] = row[HV * h + q];
#pragma unroll
                for (int q = 0; q < HV; q++) {
                    acc[HW * h + 4 * q] -= v[q].x;
                    acc[HW * h + 4 * q + 1] -= v[q].y;
                    acc[HW * h + 4 * q + 2] -= v[q].z;
                    acc[HW * h + 4 * q + 3] -= v[q].w;
                }
                // opaque to the optimiser: stops it from summing several levels' rows first
                // (which would keep all of them live) before touching the accumulators
#pragma unroll
                for (int k = 0; k < HW; k++) asm volatile("" : "+v"(acc[HW * h + k]));
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }

#undef KS_PREFETCH
#undef KS_LOAD1
#undef KS_STORE1

    // Epilogue: out (+)= (b, 0, ..) - sum, transposed through a per-wave LDS tile so that
    // consecutive lanes touch consecutive words.
    __syncthreads();
    uint32_t* s_t = s_mem + wave * (64 * 17);
    const int wbase = slab * kKsSlab + kKsWaveWords * wave;
#pragma unroll
    for (int pass = 0; pass < kKsWaveWords / 16; pass++) {
#pragma unroll
        for (int k = 0; k < 16; k++) s_t[lane * 17 + k] = acc[16 * pass + k];
        wave_lds_fence();
#pragma unroll 4
        for (int it = 0; it < 16; it++) {
            const int gl = it * 4 + (lane >> 4), k = lane & 15;
            uint32_t v = s_t[gl * 17 + k];
            const int word = wbase + 16 * pass + k, g = g0 + gl;
            if (g < n_gates && word < n1) {
                if (word == 0 && part == 0) v += lv1[(size_t)g * (nlv1 + 1)];
                uint32_t* dst = out + (size_t)g * n1 + word;
                if (parts == 1) *dst = v;
                else if (v != 0u) atomicAdd(dst, v);
            }
        }
        wave_lds_fence();
    }
}

// ------------------------------------------------------------------------------------------
// Identity key switch of a handful of gates, where the tiled kernel above would leave most of the
// GPU idle (a tile is 64 gates): one workgroup per (gate, slice of the lv1 coefficients), thread =
// two output words, so every row of the packed table is read with coalesced loads straight from
// L2 / Infinity Cache.  Digit 0 reads a row anyway and masks it (the loads of a coefficient's T
// levels are then independent of the digits and all in flight together).  Slices are combined with
// integer atomic adds into the zero-initialised output, like the slices of the tiled kernel.
// ------------------------------------------------------------------------------------------
constexpr int kKsSmallThreads = 320;
constexpr int kKsSmallMaxSlice = 64;  // lv1 coefficients per workgroup at most
#ifndef TFHE_AMD_KS_SMALL_MAX_GATES
#define TFHE_AMD_KS_SMALL_MAX_GATES 16
#endif
// larger batches take the tiled kernel (measured: 0.03 against 0.04 ms at 16 gates, 0.06 against 0.05 ms at 32)
constexpr int kKsSmallMaxGates = TFHE_AMD_KS_SMALL_MAX_GATES;

template <int T, int BB>
__global__ void __launch_bounds__(kKsSmallThreads)
key_switch_small_kernel(const uint32_t* __restrict__ lv1, const uint32_t* __restrict__ ks_packed,
                        uint32_t* __restrict__ out, int n1, int nlv1, int rowp, int slices) {
    constexpr int NR1 = (1 << BB) - 1;
    __shared__ uint32_t s_dig[kKsSmallMaxSlice];
    const int g = blockIdx.x / slices, part = blockIdx.x % slices;
    const int i_begin = (int)((long)nlv1 * part / slices), i_end = (int)((long)nlv1 * (part + 1) / slices);
    const int count = i_end - i_begin, t = threadIdx.x;
    const uint32_t* lv1_g = lv1 + (size_t)g * (nlv1 + 1);
    const uint32_t round_offset = 1u << (32 - T * BB - 1);
    if (t < count) s_dig[t] = (lv1_g[1 + i_begin + t] + round_offset) >> (32 - T * BB);
    __syncthreads();
    // words t and t + 320 of every row (rowp is a multiple of 320; rows are zero-padded beyond n1)
    const int words = rowp / kKsSmallThreads;
    for (int wi = 0; wi < words; wi += 2) {
        const bool two = wi + 1 < words;
        uint32_t acc0 = 0u, acc1 = 0u;
        const uint32_t* col = ks_packed + (size_t)i_begin * T * NR1 * rowp + wi * kKsSmallThreads + t;
#pragma unroll 2
        for (int ii = 0; ii < count; ii++) {
            const uint32_t d = s_dig[ii];
            const uint32_t* rows_i = col + (size_t)ii * T * NR1 * rowp;
#pragma unroll
            for (int m = 0; m < T; m++) {
                const uint32_t o = (d >> ((T - 1 - m) * BB)) & (uint32_t)NR1;
                const uint32_t* row = rows_i + (size_t)(m * NR1 + (o ? (int)o - 1 : 0)) * rowp;
                const uint32_t keep = o ? ~0u : 0u;
                acc0 += row[0] & keep;
                if (two) acc1 += row[kKsSmallThreads] & keep;
            }
        }
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const int word = (wi + h) * kKsSmallThreads + t;
            if ((h == 0 || two) && word < n1) {
                uint32_t v = 0u - (h ? acc1 : acc0);
                if (word == 0 && part == 0) v += lv1_g[0];
                if (v != 0u) atomicAdd(out + (size_t)g * n1 + word, v);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// Identity key switch for any decomposition (run-time T and base_bit): the same work split as the
// kernel above -- one workgroup per (gate, slice of the lv1 coefficients), a thread per output word
// and 320-word stripe, rows read with coalesced loads from the packed table -- for the bases the
// tiled kernel has no instantiation for (base_bit >= 4: staging the 2^base_bit - 1 rows of a level
// in LDS for 64 gates that use at most 64 of them pays no longer).  The digit of a level is the same
// for the whole workgroup, so a zero digit skips its row without divergence.
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kKsSmallThreads)
key_switch_gather_kernel(const uint32_t* __restrict__ lv1, const uint32_t* __restrict__ ks_packed,
                         uint32_t* __restrict__ out, int n1, int nlv1, int T, int bb, int rowp, int slices) {
    __shared__ uint32_t s_dig[kKsSmallMaxSlice];
    const size_t nr1 = ((size_t)1 << bb) - 1;
    const int g = blockIdx.x / slices, part = blockIdx.x % slices;
    const int i_begin = (int)((long)nlv1 * part / slices), i_end = (int)((long)nlv1 * (part + 1) / slices);
    const int t = threadIdx.x;
    const uint32_t* lv1_g = lv1 + (size_t)g * (nlv1 + 1);
    const uint32_t round_offset = 1u << (32 - T * bb - 1), omask = (uint32_t)nr1;
    const int dshift = 32 - T * bb;
    for (int w = t; w < rowp; w += kKsSmallThreads) {
        uint32_t acc = 0u;
        for (int c0 = i_begin; c0 < i_end; c0 += kKsSmallMaxSlice) {
            const int count = min(kKsSmallMaxSlice, i_end - c0);
            __syncthreads();
            if (t < count) s_dig[t] = (lv1_g[1 + c0 + t] + round_offset) >> dshift;
            __syncthreads();
            for (int ii = 0; ii < count; ii++) {
                const uint32_t d = s_dig[ii];
                const uint32_t* rows_i = ks_packed + (size_t)(c0 + ii) * T * nr1 * rowp + w;
                for (int m = 0; m < T; m++) {
                    const uint32_t o = (d >> ((T - 1 - m) * bb)) & omask;
                    if (o) acc += rows_i[((size_t)m * nr1 + (o - 1)) * rowp];
                }
            }
        }
        if (w < n1) {
            uint32_t v = 0u - acc;
            if (w == 0 && part == 0) v += lv1_g[0];
            if (slices == 1) out[(size_t)g * n1 + w] = v;
            else if (v != 0u) atomicAdd(out + (size_t)g * n1 + w, v);
        }
    }
}

// Zero-fill of the output before the kernels that combine slices with atomic adds.  A kernel, not
// hipMemsetAsync: a memset node recorded by stream capture replayed a stale fill pattern on the
// second and later launches of the graph (ROCm 7.0 runtime), and these launchers must be capturable.
__global__ void __launch_bounds__(256) zero_words_kernel(uint32_t* __restrict__ dst, size_t words) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < words) dst[i] = 0u;
}
static hipError_t launch_zero_words(uint32_t* dst, size_t words, hipStream_t stream) {
    if (words == 0) return hipSuccess;
    zero_words_kernel<<<(unsigned)((words + 255) / 256), 256, 0, stream>>>(dst, words);
    return hipGetLastError();
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

hipError_t launch_gate_combine(uint32_t* c, const uint32_t* a, const uint32_t* b, const uint8_t* ops, int op,
                               size_t n_gates, int n1, hipStream_t stream) {
    const size_t total = n_gates * (size_t)n1;
    if (total == 0) return hipSuccess;
    gate_combine_kernel<<<(unsigned)((total + 255) / 256), 256, 0, stream>>>(c, a, b, ops, op, total, n1);
    return hipGetLastError();
}

hipError_t launch_negate(uint32_t* out, const uint32_t* a, size_t words, hipStream_t stream) {
    if (words == 0) return hipSuccess;
    negate_kernel<<<(unsigned)((words + 255) / 256), 256, 0, stream>>>(out, a, words);
    return hipGetLastError();
}

hipError_t launch_mux_sum(uint32_t* dst, const uint32_t* src, size_t n_rows, int lv1_words, hipStream_t stream) {
    const size_t total = n_rows * (size_t)lv1_words;
    if (total == 0) return hipSuccess;
    mux_sum_kernel<<<(unsigned)((total + 255) / 256), 256, 0, stream>>>(dst, src, total, lv1_words);
    return hipGetLastError();
}

hipError_t launch_bk_transform_512(const NttUniform& u, const d2* tw, const uint32_t* bk, d2* bk_ntt, int n_polys,
                                   hipStream_t stream) {
    if (n_polys == 0) return hipSuccess;
    bk_transform_512_kernel<<<(n_polys + 3) / 4, 256, 0, stream>>>(u, tw, bk, bk_ntt, n_polys);
    return hipGetLastError();
}

// Batches up to this size run one gate per workgroup (latency form); larger ones one gate per wave.
constexpr int kLatencyMaxGates = 512;

hipError_t launch_blind_rotate_512(const NttUniform& u, const d2* tw, const d2* bk_ntt, const uint32_t* c,
                                   const uint32_t* tv, uint32_t* out, int n_gates, int n, int K, int L, int extract,
                                   int form, hipStream_t stream) {
    if (n_gates == 0) return hipSuccess;
    if (!((K == 1 || K == 2) && L == 2)) return hipErrorInvalidValue;
    const bool latency = form == 1 || (form == 0 && n_gates <= kLatencyMaxGates);
    const unsigned blocks = (unsigned)((n_gates + 3) / 4);
    if (latency && K == 2)
        blind_rotate_512_latency_kernel<2, 2><<<(unsigned)n_gates, 128 * 6, 0, stream>>>(u, tw + kTwPairs, bk_ntt, c, tv,
                                                                                       out, n_gates, n, extract);
    else if (latency)
        blind_rotate_512_latency_kernel<1, 2><<<(unsigned)n_gates, 128 * 4, 0, stream>>>(u, tw + kTwPairs, bk_ntt, c, tv,
                                                                                       out, n_gates, n, extract);
    else if (K == 2)
        blind_rotate_512_kernel<2, 2><<<blocks, 256, 0, stream>>>(u, tw, bk_ntt, c, tv, out, n_gates, n, extract);
    else
        blind_rotate_512_kernel<1, 2><<<blocks, 256, 0, stream>>>(u, tw, bk_ntt, c, tv, out, n_gates, n, extract);
    return hipGetLastError();
}

bool external_product_512_has_kernel(int N, int K, int L, int bg) {
    return N == 512 && (K == 1 || K == 2) && L == 2 && bg == 8;
}

hipError_t launch_external_product_512(const NttUniform& u, const d2* tw, const d2* key_ntt, const uint32_t* x0,
                                       const uint32_t* x1, uint32_t* out, size_t batch, int K, int L, int mode,
                                       hipStream_t stream) {
    if (batch == 0) return hipSuccess;
    if (!external_product_512_has_kernel(512, K, L, 8)) return hipErrorInvalidValue;
    const unsigned blocks = (unsigned)((batch + 3) / 4);
    if (K == 2) external_product_512_kernel<2, 2><<<blocks, 256, 0, stream>>>(u, tw, key_ntt, x0, x1, out, (int)batch, mode);
    else        external_product_512_kernel<1, 2><<<blocks, 256, 0, stream>>>(u, tw, key_ntt, x0, x1, out, (int)batch, mode);
    return hipGetLastError();
}

template <int NR1, int MG>
static hipError_t launch_key_switch_inst(const uint32_t* lv1, const uint32_t* ks_packed, uint32_t* out, int n_gates,
                                         int n1, int nlv1, int T, int bb, hipStream_t stream) {
    const size_t lds = sizeof(uint32_t) * ((size_t)(1 + 2 * MG * NR1) * kKsRowLds + (size_t)kKsChunk * kKsTile);
    hipError_t e = hipFuncSetAttribute((const void*)key_switch_kernel<NR1, MG>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    const int rowp = key_switch_row_words(n1);
    const int tiles = (n_gates + kKsTile - 1) / kKsTile, slabs = rowp / kKsSlab;
    // slices of the lv1 range: aim at two workgroups per CU, never finer than 16 coefficients
    // (every slice adds one atomic per output word and gate)
    int parts = 1;
    while (tiles * slabs * parts * 2 <= 512 && parts * 2 * 16 <= nlv1) parts *= 2;
    if (parts > 1) {
        e = launch_zero_words(out, (size_t)n_gates * n1, stream);
        if (e != hipSuccess) return e;
    }
    key_switch_kernel<NR1, MG><<<(unsigned)(tiles * slabs * parts), kKsThreads, lds, stream>>>(
        lv1, ks_packed, out, n_gates, n1, nlv1, T, bb, rowp, parts);
    return hipGetLastError();
}

// every decomposition the reference accepts (src/tlwe_key_switching.hpp:52-60: T * BASE_BIT <= 31)
bool key_switch_has_kernel(int n1, int T, int bb) {
    if (bb < 1 || T < 1 || T * bb > 31) return false;
    return n1 >= 1;
}

int key_switch_row_words(int n1) { return (n1 + kKsSlab - 1) / kKsSlab * kKsSlab; }

size_t key_switch_packed_words(int n1, int nlv1, int T, int bb) {
    return (size_t)nlv1 * T * ((1 << bb) - 1) * key_switch_row_words(n1);
}

hipError_t launch_ks_pack(uint32_t* packed, const uint32_t* ks, int n1, int nlv1, int T, int bb, hipStream_t stream) {
    const int nr1 = (1 << bb) - 1, rowp = key_switch_row_words(n1);
    const size_t rows = (size_t)nlv1 * T * nr1, total = rows * rowp;
    ks_pack_kernel<<<(unsigned)((total + 255) / 256), 256, 0, stream>>>(packed, ks, rows, nr1, n1, rowp);
    return hipGetLastError();
}

hipError_t launch_key_switch(const uint32_t* lv1, const uint32_t* ks_packed, uint32_t* out, int n_gates, int n1,
                             int nlv1, int T, int bb, hipStream_t stream) {
    if (n_gates == 0) return hipSuccess;
    if (!key_switch_has_kernel(n1, T, bb)) return hipErrorInvalidValue;
    // a handful of gates at the default decomposition: one workgroup per (gate, slice), enough
    // slices to put about a thousand workgroups on the GPU
    if (n_gates <= kKsSmallMaxGates && T == 5 && bb == 2 && nlv1 >= 8) {
        int slices = 1024 / n_gates;
        slices = std::min(slices, nlv1 / 8);
        slices = std::max(slices, (nlv1 + kKsSmallMaxSlice - 1) / kKsSmallMaxSlice);
        hipError_t e = launch_zero_words(out, (size_t)n_gates * n1, stream);
        if (e != hipSuccess) return e;
        key_switch_small_kernel<5, 2><<<(unsigned)(n_gates * slices), kKsSmallThreads, 0, stream>>>(
            lv1, ks_packed, out, n1, nlv1, key_switch_row_words(n1), slices);
        return hipGetLastError();
    }
    // The tiled kernel is bound by the LDS of a CU: two workgroups on one CU take twice as long as one,
    // so a batch a little above a whole number of workgroups per CU (256 workgroups = 8,192 gates at the
    // default row length) would pay a full extra round for its remainder.  The remainder is launched on
    // its own instead, where it gets the slicing of a small batch and spreads over all CUs.
    if (bb <= 3) {
        const int slabs = key_switch_row_words(n1) / kKsSlab;
        const int round_gates = kKsTile * std::max(1, 256 / slabs);
        const int rem = n_gates % round_gates;
        if (n_gates > round_gates && rem != 0) {
            const int head = n_gates - rem;
            hipError_t e = launch_key_switch(lv1, ks_packed, out, head, n1, nlv1, T, bb, stream);
            if (e != hipSuccess) return e;
            return launch_key_switch(lv1 + (size_t)head * (nlv1 + 1), ks_packed, out + (size_t)head * n1, rem, n1, nlv1, T,
                                     bb, stream);
        }
    }
    if (bb > 3) {
        // slices: about a thousand workgroups on the GPU, never finer than 8 coefficients
        int slices = (int)std::max<long>(1, 1024 / n_gates);
        slices = std::max(1, std::min(slices, nlv1 / 8));
        if (slices > 1) {
            hipError_t e = launch_zero_words(out, (size_t)n_gates * n1, stream);
            if (e != hipSuccess) return e;
        }
        key_switch_gather_kernel<<<(unsigned)(n_gates * slices), kKsSmallThreads, 0, stream>>>(
            lv1, ks_packed, out, n1, nlv1, T, bb, key_switch_row_words(n1), slices);
        return hipGetLastError();
    }
    // MG levels per LDS stage: the largest instantiated divisor of T
#define KS_GO(NR1, MG) return launch_key_switch_inst<NR1, MG>(lv1, ks_packed, out, n_gates, n1, nlv1, T, bb, stream)
    if (bb == 1) { if (T % 5 == 0) KS_GO(1, 5); if (T % 4 == 0) KS_GO(1, 4); KS_GO(1, 1); }
    if (bb == 2) { if (T % 5 == 0) KS_GO(3, 5); if (T % 4 == 0) KS_GO(3, 4); KS_GO(3, 1); }
    if (T % 2 == 0) KS_GO(7, 2);
    KS_GO(7, 1);
#undef KS_GO
}

}  // namespace tfhe_amd
