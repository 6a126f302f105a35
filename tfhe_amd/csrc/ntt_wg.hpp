// Workgroup-cooperative exact negacyclic NTT of length N = 2^LOGN (N >= 512) over the same
// prime field and with the same FP64 modular arithmetic as ntt512.hpp.  N/8 threads transform
// one polynomial: 8 coefficients per thread, groups of up to three radix-2 stages in registers,
// a transpose through LDS between groups: with a block barrier and two alternating buffers where
// data crosses waves (the first transpose of a forward transform, the last of an inverse one),
// through a wave-private buffer with no barrier where it does not.
//
// Stage s (distance N / 2^(s+1)) pairs positions that differ in bit beta = LOGN-1-s.  Group g
// covers stages 3g .. 3g+2; its register layout keeps bits b+2..b of the position in the
// register index k, with b = LOGN - 3(g+1), and spreads the other bits over the threads:
//     position(t, k) = ((t >> b) << (b + 3)) | (k << b) | (t & (2^b - 1)).
// When LOGN is not a multiple of 3 the last group uses b = 0 and applies only the remaining
// one or two stages.  Twiddles: group 0 is thread-uniform (kernel argument); the other groups
// read 7 (w, RN(w/p)) pairs per thread from a table row U = t >> b:
//     [0] = tw[m + U], [1 + h] = tw[2m + 2U + h], [3 + h] = tw[4m + 4U + h],  m = 2^(3g)
// (for the partial last group m = 2^(LOGN-3) and the unused leading entries are ignored).
#pragma once

#include "ntt512.hpp"

namespace tfhe_amd {

template <int LOGN>
struct WgNtt {
    static constexpr int N = 1 << LOGN;
    static constexpr int T = N / 8;                  // threads per polynomial
    static constexpr int GROUPS = (LOGN + 2) / 3;
    static constexpr int LAST = LOGN - 3 * (GROUPS - 1);  // stages in the last group: 1, 2 or 3
    // block transpose buffer: one pad slot per 8 positions, none for LOGN = 11 (see block_pos)
    static constexpr int PAD = LOGN == 11 ? N : N + N / 8;
    // A transpose between two layouts whose bases are both <= 6 moves data only inside a wave: the
    // thread bits from 6 up are position bits from 9 up in both, so wave v holds positions
    // [512 v, 512 v + 512) before and after.  Such a transpose goes through the wave's own buffer
    // and needs no block barrier.
    static constexpr int WAVE_PAD = 512 + 512 / 8;
    static constexpr int BUF_DOUBLES = 2 * PAD + (T / 64) * WAVE_PAD;  // two block buffers, one buffer per wave

    __host__ __device__ static constexpr int base(int g) { return (g == GROUPS - 1) ? 0 : LOGN - 3 * (g + 1); }
    __host__ __device__ static constexpr int stages(int g) { return (g == GROUPS - 1) ? LAST : 3; }
    __host__ __device__ static constexpr int rows(int g) { return T >> base(g); }
    // table row offset (in rows) of group g >= 1 inside one direction's table
    __host__ __device__ static constexpr int row_offset(int g) {
        int o = 0;
        for (int i = 1; i < g; i++) o += rows(i);
        return o;
    }
    static constexpr int TABLE_ROWS = row_offset(GROUPS);  // per direction
    static constexpr int TABLE_PAIRS = 2 * TABLE_ROWS * 7;  // forward then inverse

    __device__ static __forceinline__ int position(int g, int t, int k) {
        const int b = base(g);
        return ((t >> b) << (b + 3)) | (k << b) | (t & ((1 << b) - 1));
    }

    // Twiddle rows kept in LDS by kernels that have room (the rest is read from L1 / L2): level 1 = group 1
    // of both directions (T >> base(1) rows each, every row shared by 2^base(1) threads), level 2 = also
    // group 2 of the forward direction.  Layout: [forward group 1][inverse group 1][forward group 2].
    __host__ __device__ static constexpr int lds_row_pairs(int level) {
        return level == 0 ? 0 : 7 * (2 * rows(1) + (level >= 2 ? rows(2) : 0));
    }
    template <int LEVEL>
    __device__ static __forceinline__ void stage_rows(d2* lds, const d2* __restrict__ table, int t) {
        if (LEVEL == 0) return;
        for (int e = t; e < 7 * rows(1); e += T) {
            lds[e] = table[7 * row_offset(1) + e];
            lds[7 * rows(1) + e] = table[7 * TABLE_ROWS + 7 * row_offset(1) + e];
        }
        if (LEVEL >= 2)
            for (int e = t; e < 7 * rows(2); e += T) lds[14 * rows(1) + e] = table[7 * row_offset(2) + e];
    }
    // rows of group g (>= 1) of direction dir (0 forward, 1 inverse)
    template <int LEVEL>
    __device__ static __forceinline__ const d2* group_rows(const d2* __restrict__ table, const d2* lds, int dir, int g) {
        if (LEVEL >= 1 && g == 1) return lds + dir * 7 * rows(1);
        if (LEVEL >= 2 && g == 2 && dir == 0) return lds + 14 * rows(1);
        return table + dir * 7 * TABLE_ROWS + 7 * row_offset(g);
    }

    // Identity the optimiser cannot see through.  The twiddle rows are loop-invariant in a blind
    // rotation; without this their loads are hoisted out of the step loop and every group's 28
    // registers stay live for the whole kernel (168 registers, which costs the second workgroup
    // per CU).
    __device__ static __forceinline__ int opaque(int v) {
        asm volatile("" : "+v"(v));
        return v;
    }
#ifndef TFHE_AMD_WG_EXPERIMENT
#define TFHE_AMD_WG_EXPERIMENT 0
#endif
#ifndef TFHE_AMD_WG_TR_PAD
#define TFHE_AMD_WG_TR_PAD 1  // per-exchange pad of the wave-private transpositions (0: one slot per 8 positions)
#endif
#ifndef TFHE_AMD_WG_PIPE_EXCHANGE
#define TFHE_AMD_WG_PIPE_EXCHANGE -1  // -1: where it was measured to pay (forward_many), 0 / 1: never / always
#endif
    // timing experiments (wrong results): 1 = every thread reads twiddle row 0 of its group,
    // 2 = no twiddle loads (thread-uniform values), 3 = no block barrier in the transposes,
    // 4 = no transposition at all, 5 = every key tile read from the first tile's address (wg_kernels.hip),
    // 6 = aligned instead of rotated accumulator reads (wg_kernels.hip), 7 = no wave-private transposition,
    // 8 = the key rows of steps 0..7 in rotation (all key reads hit L2; wg_kernels.hip)
    __device__ static __forceinline__ void load_row(Tw7& tw, const d2* rows, int row, const NttUniform& u) {
        if (TFHE_AMD_WG_EXPERIMENT == 2) {
#pragma unroll
            for (int i = 0; i < 7; i++) { tw.w[i] = u.fw[1 + i]; tw.wp[i] = u.fwp[1 + i]; }
            return;
        }
        load_tw7(tw, rows + 7 * (TFHE_AMD_WG_EXPERIMENT == 1 ? opaque(0) : row));
    }

    // Slot of position j in a block buffer.  The block-level transposition pairs group 0 (a wave's
    // lanes on 64 consecutive positions) with group 1 (32-lane halves on 2^b consecutive positions,
    // b = base(1)).  For LOGN = 11 the halves of a wave are 256 positions = 2 KiB apart, every access of
    // either layout is a run of consecutive doubles, and any padding only makes the ends of a run
    // collide; for LOGN = 10 the 16-lane quarters of group 1 are 1 KiB apart and need the pad slot
    // per 8 positions to land on different banks.
    __device__ static __forceinline__ int block_pos(int j) { return LOGN == 11 ? j : pad_pos(j); }

    // registers in layout `from` -> registers in layout `to`.  bufs: BUF_DOUBLES doubles; `parity`
    // selects the block buffer and is advanced once per block-level transpose.
    __device__ static __forceinline__ void transpose(double (&x)[8], double* bufs, int& parity, int from, int to, int t) {
        if (TFHE_AMD_WG_EXPERIMENT == 4) return;  // no exchange at all
        if (TFHE_AMD_WG_EXPERIMENT == 7 && base(from) <= 6 && base(to) <= 6) return;  // no wave-private exchange
        if (base(from) <= 6 && base(to) <= 6) {
            double* wb = bufs + 2 * PAD + (t >> 6) * WAVE_PAD;
            // One pad run per 2^a positions, a = the larger of the two bases (at least 3), 2^(a-3) slots
            // long: a 32-lane group of the layout with base a then reads consecutive doubles and one of
            // the other layout runs that start a multiple of 2^(a-3) banks-of-doubles apart, so neither
            // side has a bank conflict (with the pad slot per 8 positions of pad_pos for every exchange,
            // a base above 3 put the ends of its 2^a-runs on the same banks).
            const int a = (base(from) > base(to) ? base(from) : base(to)) < 3 ? 3 : (base(from) > base(to) ? base(from) : base(to));
            auto slot = [a](int j) { return TFHE_AMD_WG_TR_PAD ? j + ((j >> a) << (a - 3)) : pad_pos(j); };
#pragma unroll
            for (int k = 0; k < 8; k++) wb[slot(position(from, t, k) & 511)] = x[k];
            wave_lds_fence();
#pragma unroll
            for (int k = 0; k < 8; k++) x[k] = wb[slot(position(to, t, k) & 511)];
            wave_lds_fence();
            return;
        }
        double* buf = bufs + (parity & 1) * PAD;
        parity++;
#pragma unroll
        for (int k = 0; k < 8; k++) buf[block_pos(position(from, t, k))] = x[k];
        if (TFHE_AMD_WG_EXPERIMENT != 3) __syncthreads();
#pragma unroll
        for (int k = 0; k < 8; k++) x[k] = buf[block_pos(position(to, t, k))];
    }

    template <int STAGES, bool REDUCE_A>
    __device__ static __forceinline__ void ct_group(double (&x)[8], const Tw7& tw, double p, double pinv) {
        if (STAGES >= 3) {
#pragma unroll
            for (int k = 0; k < 4; k++) {
                if (REDUCE_A) x[k] = reduce(x[k], p, pinv);
                bf_ct(x[k], x[k + 4], tw.w[0], tw.wp[0], p);
            }
        }
        if (STAGES >= 2) {
#pragma unroll
            for (int h = 0; h < 2; h++)
#pragma unroll
                for (int k = 0; k < 2; k++) {
                    if (REDUCE_A && STAGES == 2) x[4 * h + k] = reduce(x[4 * h + k], p, pinv);
                    bf_ct(x[4 * h + k], x[4 * h + k + 2], tw.w[1 + h], tw.wp[1 + h], p);
                }
        }
#pragma unroll
        for (int g = 0; g < 4; g++) {
            if (REDUCE_A && STAGES == 1) x[2 * g] = reduce(x[2 * g], p, pinv);
            bf_ct(x[2 * g], x[2 * g + 1], tw.w[3 + g], tw.wp[3 + g], p);
        }
    }

    // lazy reductions of the inverse transform: the compile-time plan of ntt512.hpp for this ring's
    // stage groups (register k of a thread has a bound that depends on k alone)
    // N = 1024 keeps the plain scheme (all eight registers reduced at every group start, written as
    // one loop over the groups): its kernels sit at the 256-register limit, and the planned form,
    // although it executes 12 reductions fewer per transform, makes the compiler spill more there
    // (measured: 1024/K=1/L=2 4 % slower, K=2 1.5 % slower; N = 512 and N = 2048 0.7 - 0.8 % faster).
    static constexpr bool kPlanned = LOGN != 10;
    static constexpr GsPlan kInversePlan = make_gs_plan(GROUPS, LAST, !kPlanned);
    static_assert(kInversePlan.out <= 4.0, "the callers of the inverse transform assume outputs below 4 p");
    template <int G, int S>
    static constexpr unsigned pre_mask() { return kInversePlan.pre[G][S]; }
    template <int G>
    static constexpr unsigned carry_mask() { return kInversePlan.carry[G]; }

    // the STAGES stages of inverse group G (twiddles in tw: [3..6] first stage, [1..2] second, [0] third)
    template <int G, int STAGES>
    __device__ static __forceinline__ void gs_group(double (&x)[8], const double* w, const double* wp, double p,
                                                    double pinv) {
        tfhe_amd::reduce_masked<pre_mask<G, 0>(), false>(x, p, pinv);
#pragma unroll
        for (int g = 0; g < 4; g++) bf_gs(x[2 * g], x[2 * g + 1], w[3 + g], wp[3 + g], p);
        if (STAGES >= 2) {
            tfhe_amd::reduce_masked<pre_mask<G, 1>(), false>(x, p, pinv);
#pragma unroll
            for (int h = 0; h < 2; h++)
#pragma unroll
                for (int k = 0; k < 2; k++) bf_gs(x[4 * h + k], x[4 * h + k + 2], w[1 + h], wp[1 + h], p);
        }
        if (STAGES >= 3) {
            tfhe_amd::reduce_masked<pre_mask<G, 2>(), false>(x, p, pinv);
#pragma unroll
            for (int k = 0; k < 4; k++) bf_gs(x[k], x[k + 4], w[0], wp[0], p);
        }
    }

    // M forward transforms group by group, so that every group's twiddle row is loaded once for
    // all of them.  In: group-0 layout (position k*T + t), |x| <= 2^31 (SMALL = 1: |x| <= 2^7, 8-bit
    // gadget digits; SMALL = 2: |x| <= 2^15, 16-bit digits, where only the first stage's twiddle
    // zeta^2 < 2^26 is small enough to need no reduction).
    // Out: last group's layout (position 8t + k, bit-reversed spectrum order), |x| <= 3.2 p.
    // Additive inputs are reduced at the first stage of groups >= 2 (SMALL) or >= 1; the bound
    // then stays below 4.2 p at every stage for LOGN <= 12.
    // The growth stated above, recomputed by the compiler: largest output bound of forward_many<SMALL>, or -1
    // if the operand of some twiddle product could pass 2^53.  Group 0 has 3 - SMALL' full butterfly stages
    // (SMALL = 1 or 2 leave two), a later group reduces the additive inputs of its first stage if it is the
    // last one, if g >= 2, or if SMALL = 0.
    static constexpr double forward_bound(int small) {
        double b = 0.000002;  // inputs up to 2^31
        bool ok = true;
        for (int g = 0; g < GROUPS; g++) {
            const int st = g == 0 ? (small ? 2 : 3) : stages(g);
            const bool reduce_a = g > 0 && (g == GROUPS - 1 || g >= 2 || small == 0);
            for (int s = 0; s < st; s++) {
                ok = ok && b <= kGsTwo53;
                const double a = (s == 0 && reduce_a) ? 0.5 + kGsEps : b;
                b = a + 0.5 + 0.75 * b / kGsTwo53 + kGsEps;
            }
        }
        return ok ? b : -1.0;
    }
    static_assert(forward_bound(0) >= 0 && forward_bound(0) <= 3.2 && forward_bound(1) >= 0 && forward_bound(1) <= 3.2 &&
                      forward_bound(2) >= 0 && forward_bound(2) <= 3.2,
                  "forward_many: outputs above 3.2 p, or an operand above 2^53");
    template <int SMALL, int M, int LDS_LEVEL = 0>
    __device__ static __forceinline__ void forward_many(double (&xs)[M][8], const NttUniform& u,
                                                        const d2* __restrict__ table, double* bufs, int& parity, int t,
                                                        const d2* lds_rows = nullptr) {
        const double p = u.p, pinv = u.pinv;
#pragma unroll
        for (int m = 0; m < M; m++) {
            double (&x)[8] = xs[m];
            if (SMALL == 1) {
                small_first_two_stages(x, u);  // no reduction in either stage (ntt512.hpp)
            } else if (SMALL == 2) {
                // |x * zeta^2| < 2^41.3: exact, far below p
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const double a = x[k], b = x[k + 4];
                    x[k] = __builtin_fma(b, u.zI, a);
                    x[k + 4] = __builtin_fma(-b, u.zI, a);
                }
#pragma unroll
                for (int h = 0; h < 2; h++)
#pragma unroll
                    for (int k = 0; k < 2; k++) bf_ct(x[4 * h + k], x[4 * h + k + 2], u.fw[2 + h], u.fwp[2 + h], p);
            } else {
#pragma unroll
                for (int k = 0; k < 4; k++) bf_ct(x[k], x[k + 4], u.fw[1], u.fwp[1], p);
#pragma unroll
                for (int h = 0; h < 2; h++)
#pragma unroll
                    for (int k = 0; k < 2; k++) bf_ct(x[4 * h + k], x[4 * h + k + 2], u.fw[2 + h], u.fwp[2 + h], p);
            }
#pragma unroll
            for (int g = 0; g < 4; g++) bf_ct(x[2 * g], x[2 * g + 1], u.fw[4 + g], u.fwp[4 + g], p);
        }
#pragma unroll
        for (int g = 1; g < GROUPS; g++) {
            // keeps the compiler from hoisting every group's twiddle loads (28 registers each) to
            // the top of the transform
            __builtin_amdgcn_sched_barrier(0);
            Tw7 tw;
            load_row(tw, group_rows<LDS_LEVEL>(table, lds_rows, 0, g), opaque(t >> base(g)), u);
            // Pipelined form: the exchange of transform m + 1 is issued before the butterflies of transform m,
            // which then cover it (a wave's LDS instructions complete in issue order, so one buffer serves
            // back-to-back exchanges; the block-level ones alternate two buffers as before).  Measured on
            // 4,096 / 2,048 gates: N = 1024, K = 1, L = 2 47.4 -> 43.9 ms; N = 1024 with L = 3 or K = 2 and
            // N = 2048 1 - 2 % slower (the other workgroup of the CU already covers the exchange there, and
            // the longer live ranges cost registers), so only pairs of N = 1024 transforms use it.
            constexpr bool kPipe = TFHE_AMD_WG_PIPE_EXCHANGE < 0 ? (LOGN == 10 && M == 2) : TFHE_AMD_WG_PIPE_EXCHANGE != 0;
            if (kPipe) transpose(xs[0], bufs, parity, g - 1, g, t);
#pragma unroll
            for (int m = 0; m < M; m++) {
                if (kPipe) {
                    if (m + 1 < M) transpose(xs[m + 1], bufs, parity, g - 1, g, t);
                } else {
                    transpose(xs[m], bufs, parity, g - 1, g, t);
                }
                constexpr bool kAlways = SMALL == 0;
                if (g == GROUPS - 1) {
                    ct_group<LAST, true>(xs[m], tw, p, pinv);
                } else if (g >= 2 || kAlways) {
                    ct_group<3, true>(xs[m], tw, p, pinv);
                } else {
                    ct_group<3, false>(xs[m], tw, p, pinv);
                }
            }
        }
    }

    template <int SMALL>
    __device__ static __forceinline__ void forward(double (&x)[8], const NttUniform& u, const d2* __restrict__ table,
                                                   double* bufs, int& parity, int t) {
        double (&one)[1][8] = reinterpret_cast<double (&)[1][8]>(x);
        forward_many<SMALL, 1>(one, u, table, bufs, parity, t);
    }

    // M inverse transforms without the 1/N factor, group by group, so that every group's twiddle
    // row is loaded once for all of them.  In: last group's layout, |x| < 2^53.  Out: group-0
    // layout, |x| <= 4 p (kInversePlan.out).  Reductions: kInversePlan.
    template <int G, int M, int LDS_LEVEL>
    __device__ static __forceinline__ void inverse_group(double (&x)[M][8], const NttUniform& u,
                                                         const d2* __restrict__ table, const d2* lds_rows, double* bufs,
                                                         int& parity, int t) {
        const double p = u.p, pinv = u.pinv;
        __builtin_amdgcn_sched_barrier(0);
        Tw7 tw;
        load_row(tw, group_rows<LDS_LEVEL>(table, lds_rows, 1, G), opaque(t >> base(G)), u);
#pragma unroll
        for (int m = 0; m < M; m++) {
            gs_group<G, stages(G)>(x[m], tw.w, tw.wp, p, pinv);
            tfhe_amd::reduce_masked<carry_mask<G>(), false>(x[m], p, pinv);
            transpose(x[m], bufs, parity, G, G - 1, t);
        }
    }
    template <int STAGES>
    __device__ static __forceinline__ void gs_group_plain(double (&x)[8], const Tw7& tw, double p) {
#pragma unroll
        for (int g = 0; g < 4; g++) bf_gs(x[2 * g], x[2 * g + 1], tw.w[3 + g], tw.wp[3 + g], p);
        if (STAGES >= 2) {
#pragma unroll
            for (int h = 0; h < 2; h++)
#pragma unroll
                for (int k = 0; k < 2; k++) bf_gs(x[4 * h + k], x[4 * h + k + 2], tw.w[1 + h], tw.wp[1 + h], p);
        }
        if (STAGES >= 3) {
#pragma unroll
            for (int k = 0; k < 4; k++) bf_gs(x[k], x[k + 4], tw.w[0], tw.wp[0], p);
        }
    }
    template <int M, int LDS_LEVEL>
    __device__ static __forceinline__ void inverse_many_plain(double (&x)[M][8], const NttUniform& u,
                                                              const d2* __restrict__ table, double* bufs, int& parity,
                                                              int t, const d2* lds_rows) {
        const double p = u.p, pinv = u.pinv;
#pragma unroll
        for (int g = GROUPS - 1; g >= 1; g--) {
            __builtin_amdgcn_sched_barrier(0);
            Tw7 tw;
            load_row(tw, group_rows<LDS_LEVEL>(table, lds_rows, 1, g), opaque(t >> base(g)), u);
#pragma unroll
            for (int m = 0; m < M; m++) {
#pragma unroll
                for (int k = 0; k < 8; k++) x[m][k] = reduce(x[m][k], p, pinv);
                if (g == GROUPS - 1) gs_group_plain<LAST>(x[m], tw, p);
                else                 gs_group_plain<3>(x[m], tw, p);
                transpose(x[m], bufs, parity, g, g - 1, t);
            }
        }
#pragma unroll
        for (int m = 0; m < M; m++) {
#pragma unroll
            for (int k = 0; k < 8; k++) x[m][k] = reduce(x[m][k], p, pinv);
#pragma unroll
            for (int g = 0; g < 4; g++) bf_gs(x[m][2 * g], x[m][2 * g + 1], u.iw[4 + g], u.iwp[4 + g], p);
#pragma unroll
            for (int h = 0; h < 2; h++)
#pragma unroll
                for (int k = 0; k < 2; k++) bf_gs(x[m][4 * h + k], x[m][4 * h + k + 2], u.iw[2 + h], u.iwp[2 + h], p);
#pragma unroll
            for (int k = 0; k < 4; k++) bf_gs(x[m][k], x[m][k + 4], u.iw[1], u.iwp[1], p);
        }
    }
    template <int M, int LDS_LEVEL = 0>
    __device__ static __forceinline__ void inverse_many(double (&x)[M][8], const NttUniform& u,
                                                        const d2* __restrict__ table, double* bufs, int& parity, int t,
                                                        const d2* lds_rows = nullptr) {
        if constexpr (!kPlanned) {
            inverse_many_plain<M, LDS_LEVEL>(x, u, table, bufs, parity, t, lds_rows);
            return;
        }
        static_assert(GROUPS == 3 || GROUPS == 4, "groups 1 .. GROUPS-1 are written out below");
        if constexpr (GROUPS == 4) inverse_group<3, M, LDS_LEVEL>(x, u, table, lds_rows, bufs, parity, t);
        inverse_group<2, M, LDS_LEVEL>(x, u, table, lds_rows, bufs, parity, t);
        inverse_group<1, M, LDS_LEVEL>(x, u, table, lds_rows, bufs, parity, t);
        // group 0: thread-uniform twiddles u.iw[1 .. 7], laid out like a table row shifted by one
#pragma unroll
        for (int m = 0; m < M; m++) gs_group<0, 3>(x[m], u.iw + 1, u.iwp + 1, u.p, u.pinv);
    }

    __device__ static __forceinline__ void inverse(double (&x)[8], const NttUniform& u, const d2* __restrict__ table,
                                                   double* bufs, int& parity, int t) {
        double (&one)[1][8] = reinterpret_cast<double (&)[1][8]>(x);
        inverse_many<1>(one, u, table, bufs, parity, t);
    }
};

}  // namespace tfhe_amd
