// Device side of the exact negacyclic NTT of length 512 over F_p, p = 6432^4 + 1, evaluated
// on the FP64 vector pipe of gfx950.
//
// Representation: a field element is an integer held exactly in a double, lazily reduced
// (|x| may grow to a few p between reductions; every bound is stated where it is used and
// all of them stay below 2^53 = 5.26 p).  A product x*w is reduced with one rounded
// multiply, one FMA that recovers the rounding error of that multiply exactly, a quotient
// estimate, and one FMA that subtracts quotient*p exactly:
//     h = RN(x*w)   l = x*w - h (exact)   q = rint(x * RN(w/p))   r = (h - q*p) + l
// h - q*p is an integer below 2^53 in magnitude, so both FMAs are exact and r = x*w - q*p.
//
// One wave transforms one polynomial: 8 coefficients per lane, three groups of three
// radix-2 stages done in registers, two transposes through a private LDS buffer between
// them.  Lane/register <-> position maps (j = position, l = lane, k = register):
//     layout A: j = 64 k + l                  (stages with distance 256, 128, 64)
//     layout B: j = 64 (l >> 3) + 8 k + (l & 7)   (distance 32, 16, 8)
//     layout C: j = 8 l + k                   (distance 4, 2, 1)
// The forward transform is the merged-twiddle Cooley-Tukey form (natural order in,
// bit-reversed order out, twiddle psi^bitrev(m+i) for group i of stage m); the inverse is
// the matching Gentleman-Sande form.  The 1/N factor is folded into the bootstrapping key.
#pragma once

#include <utility>

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace tfhe_amd {

typedef double d2 __attribute__((ext_vector_type(2)));

// Wave-uniform constants, passed by value as a kernel argument (scalar registers).
struct NttUniform {
    double p, pinv;
    double zI, z;          // zeta^2 = sqrt(-1) and zeta, exact small integers
    double fw[8], fwp[8];  // forward twiddles of the three layout-A stages, index m+i in 1..7
    double iw[8], iwp[8];  // inverse twiddles of the three layout-A stages
    double ninv, ninvp;    // 1/N and RN((1/N)/p), used when the key is transformed
};

// Per-lane twiddles of one register-resident group of three stages.
struct Tw7 {
    double w[7], wp[7];  // [0] distance-4 stage, [1..2] distance-2 stage, [3..6] distance-1 stage
};

// LDS twiddle block: four sections of 7 (w, wp) pairs per row.
//   section 0: forward layout B, row = l >> 3   (8 rows)
//   section 1: forward layout C, row = l        (64 rows)
//   section 2: inverse layout B                 (8 rows)
//   section 3: inverse layout C                 (64 rows)
constexpr int kTwRows = 8 + 64 + 8 + 64;
constexpr int kTwPairs = kTwRows * 7;  // 1008 pairs = 16128 bytes
constexpr int kTwFwdB = 0, kTwFwdC = 8 * 7, kTwInvB = (8 + 64) * 7, kTwInvC = (8 + 64 + 8) * 7;

constexpr int kN = 512;
constexpr int kTrPad = kN + kN / 8;  // transpose buffer, one pad slot per 8 positions

__device__ __forceinline__ int pad_pos(int j) { return j + (j >> 3); }

// x*w mod p, lazily: result = x*w - q*p with |result| <= (0.5 + 0.75 |x| / 2^53) p.
// Needs |x| < 2^53, |w| <= p/2, wp = RN(w/p).
__device__ __forceinline__ double mulmod_tw(double x, double w, double wp, double p) {
    const double q = __builtin_rint(x * wp);
    const double h = x * w;
    const double l = __builtin_fma(x, w, -h);
    const double r = __builtin_fma(-q, p, h);
    return r + l;
}

// x*y mod p with both operands variable: result = x*y - q*p, |result| <= (0.5 + 1.5 |x| / 2^53) p.
// Needs |y| <= p/2 and |x| < 2^53.
__device__ __forceinline__ double mulmod_var(double x, double y, double p, double pinv) {
    const double h = x * y;
    const double l = __builtin_fma(x, y, -h);
    const double q = __builtin_rint(h * pinv);
    const double r = __builtin_fma(-q, p, h);
    return r + l;
}

// centred residue: |result| <= p/2 (+ one unit when x/p is within rounding of a half-integer)
__device__ __forceinline__ double reduce(double x, double p, double pinv) {
    const double q = __builtin_rint(x * pinv);
    return __builtin_fma(-q, p, x);
}

__device__ __forceinline__ void bf_ct(double& a, double& b, double w, double wp, double p) {
    const double t = mulmod_tw(b, w, wp, p);
    b = a - t;
    a = a + t;
}

__device__ __forceinline__ void bf_gs(double& a, double& b, double w, double wp, double p) {
    const double s = a + b;
    const double d = a - b;
    a = s;
    b = mulmod_tw(d, w, wp, p);
}

__device__ __forceinline__ void load_tw7(Tw7& t, const d2* row) {
#pragma unroll
    for (int i = 0; i < 7; i++) {
        const d2 v = row[i];
        t.w[i] = v.x;
        t.wp[i] = v.y;
    }
}

// three Cooley-Tukey stages on the 8 registers of a lane: pairs (k, k+4), (k, k+2), (k, k+1)
template <bool REDUCE_A>
__device__ __forceinline__ void ct8(double (&x)[8], const Tw7& t, double p, double pinv) {
#pragma unroll
    for (int k = 0; k < 4; k++) {
        if (REDUCE_A) x[k] = reduce(x[k], p, pinv);
        bf_ct(x[k], x[k + 4], t.w[0], t.wp[0], p);
    }
#pragma unroll
    for (int h = 0; h < 2; h++)
#pragma unroll
        for (int k = 0; k < 2; k++) bf_ct(x[4 * h + k], x[4 * h + k + 2], t.w[1 + h], t.wp[1 + h], p);
#pragma unroll
    for (int g = 0; g < 4; g++) bf_ct(x[2 * g], x[2 * g + 1], t.w[3 + g], t.wp[3 + g], p);
}

// three Gentleman-Sande stages: pairs (k, k+1), (k, k+2), (k, k+4)
__device__ __forceinline__ void gs8(double (&x)[8], const Tw7& t, double p) {
#pragma unroll
    for (int g = 0; g < 4; g++) bf_gs(x[2 * g], x[2 * g + 1], t.w[3 + g], t.wp[3 + g], p);
#pragma unroll
    for (int h = 0; h < 2; h++)
#pragma unroll
        for (int k = 0; k < 2; k++) bf_gs(x[4 * h + k], x[4 * h + k + 2], t.w[1 + h], t.wp[1 + h], p);
#pragma unroll
    for (int k = 0; k < 4; k++) bf_gs(x[k], x[k + 4], t.w[0], t.wp[0], p);
}

// ---- phase-ordered forms -------------------------------------------------------------------
// A wave issues a dependent FP64 instruction only every ~9 cycles but independent ones every
// ~5.5 (measured, tools/fp64_latency.hip), and the compiler emits each butterfly as one serial
// chain.  These forms evaluate NB independent chains phase by phase and fence every phase with
// a scheduling barrier, so that dependent instructions end up NB apart.
__device__ __forceinline__ void pin() { __builtin_amdgcn_sched_barrier(0); }

template <int NB>
__device__ __forceinline__ void mulmod_tw_n(double (&t)[NB], const double (&b)[NB], const double (&w)[NB],
                                            const double (&wp)[NB], double p) {
    double q[NB], h[NB];
#pragma unroll
    for (int i = 0; i < NB; i++) q[i] = b[i] * wp[i];
#pragma unroll
    for (int i = 0; i < NB; i++) h[i] = b[i] * w[i];
    pin();
#pragma unroll
    for (int i = 0; i < NB; i++) q[i] = __builtin_rint(q[i]);
#pragma unroll
    for (int i = 0; i < NB; i++) t[i] = __builtin_fma(b[i], w[i], -h[i]);
    pin();
#pragma unroll
    for (int i = 0; i < NB; i++) h[i] = __builtin_fma(-q[i], p, h[i]);
    pin();
#pragma unroll
    for (int i = 0; i < NB; i++) t[i] = h[i] + t[i];
    pin();
}

template <int NB>
__device__ __forceinline__ void mulmod_var_n(double (&t)[NB], const double (&x)[NB], const double (&y)[NB], double p,
                                             double pinv) {
    double q[NB], h[NB];
#pragma unroll
    for (int i = 0; i < NB; i++) h[i] = x[i] * y[i];
    pin();
#pragma unroll
    for (int i = 0; i < NB; i++) q[i] = h[i] * pinv;
#pragma unroll
    for (int i = 0; i < NB; i++) t[i] = __builtin_fma(x[i], y[i], -h[i]);
    pin();
#pragma unroll
    for (int i = 0; i < NB; i++) q[i] = __builtin_rint(q[i]);
    pin();
#pragma unroll
    for (int i = 0; i < NB; i++) h[i] = __builtin_fma(-q[i], p, h[i]);
    pin();
#pragma unroll
    for (int i = 0; i < NB; i++) t[i] = h[i] + t[i];
    pin();
}

template <int NB>
__device__ __forceinline__ void reduce_n(double (&x)[NB], double p, double pinv) {
    double q[NB];
#pragma unroll
    for (int i = 0; i < NB; i++) q[i] = x[i] * pinv;
    pin();
#pragma unroll
    for (int i = 0; i < NB; i++) q[i] = __builtin_rint(q[i]);
    pin();
#pragma unroll
    for (int i = 0; i < NB; i++) x[i] = __builtin_fma(-q[i], p, x[i]);
    pin();
}

// ---- lazy reductions of an inverse transform, planned at compile time -----------------------
// Bounds in units of p.  A Gentleman-Sande butterfly maps (a, b) to (a + b, mulmod(a - b)); the
// difference must stay below 2^53 = 5.26 p, and a product is at most (0.5 + 0.75 |d| / 5.26) p.  Only
// the sum outputs grow, so after a group of stages few registers are large: the bound of register k
// depends on k alone, and a register is reduced (to p/2) only where the plan says so:
//   * before a stage, both operands of every pair whose bounds add up to more than kGsPairLimit;
//   * after a group, before its transposition, every register above kGsCarryLimit.  After the
//     transposition a lane holds eight values of one such class, so the next group starts from the
//     largest bound that was left.
// The transform's inputs are taken as the full 5.26 p, which makes the first stage reduce them all.
// Groups are numbered like the stage groups of a transform: the inverse runs from group `groups - 1`
// (`last` stages, 1 to 3) down to group 0 (three stages); stage s of a group pairs registers 2^s apart.
constexpr double kGsPairLimit = 5.2, kGsCarryLimit = 1.3, kGsTwo53 = 5.26, kGsEps = 1e-6;
struct GsPlan {
    unsigned pre[4][3];  // [group][stage]: registers reduced before the stage (bit k)
    unsigned carry[4];   // [group]: registers reduced after the group
    double out;          // bound of the transform's outputs
};
constexpr GsPlan make_gs_plan(int groups, int last, bool classic = false) {
    GsPlan pl{};
    if (classic) {  // every group starts from eight reduced registers: outputs of a group <= 4 p
        for (int g = 0; g < groups; g++) pl.pre[g][0] = 0xFFu;
        pl.out = 4.0;
        return pl;
    }
    double b[8] = {kGsTwo53, kGsTwo53, kGsTwo53, kGsTwo53, kGsTwo53, kGsTwo53, kGsTwo53, kGsTwo53};
    for (int g = groups - 1; g >= 0; g--) {
        const int st = g == groups - 1 ? last : 3;
        for (int s = 0; s < st; s++) {
            const int dist = 1 << s;
            for (int i = 0; i < 4; i++) {
                const int lo = (i / dist) * 2 * dist + (i % dist), hi = lo + dist;
                if (b[lo] + b[hi] > kGsPairLimit) {
                    pl.pre[g][s] |= (1u << lo) | (1u << hi);
                    b[lo] = b[hi] = 0.5 + kGsEps;
                }
                const double d = b[lo] + b[hi];
                b[lo] = d;
                b[hi] = 0.5 + 0.75 * d / kGsTwo53 + kGsEps;
            }
        }
        if (g > 0) {
            double top = 0.5 + kGsEps;
            for (int k = 0; k < 8; k++) {
                if (b[k] > kGsCarryLimit) pl.carry[g] |= 1u << k;
                else if (b[k] > top) top = b[k];
            }
            for (int k = 0; k < 8; k++) b[k] = top;
        }
    }
    pl.out = 0.0;
    for (int k = 0; k < 8; k++) pl.out = b[k] > pl.out ? b[k] : pl.out;
    return pl;
}

constexpr int gs_popcount(unsigned m) { return m ? (int)(m & 1u) + gs_popcount(m >> 1) : 0; }
constexpr int gs_nth_bit(unsigned m, int n) {  // index of the n-th set bit
    int k = 0;
    for (;; k++)
        if ((m >> k) & 1u) {
            if (n == 0) break;
            n--;
        }
    return k;
}

// centred residues of the registers in MASK, phase-ordered like reduce_n
template <unsigned MASK, int I>
struct GsNthBit {
    static constexpr int value = gs_nth_bit(MASK, I);
};
template <unsigned MASK, bool PIN, int... I>
__device__ __forceinline__ void reduce_masked_impl(double (&x)[8], double p, double pinv,
                                                   std::integer_sequence<int, I...>) {
    if constexpr (PIN) {
        double v[sizeof...(I)] = {x[GsNthBit<MASK, I>::value]...};
        reduce_n<(int)sizeof...(I)>(v, p, pinv);
        ((x[GsNthBit<MASK, I>::value] = v[I]), ...);
    } else {
        ((x[GsNthBit<MASK, I>::value] = reduce(x[GsNthBit<MASK, I>::value], p, pinv)), ...);
    }
}
// PIN: phase by phase behind scheduling barriers (reduce_n); otherwise left to the compiler's scheduler
template <unsigned MASK, bool PIN = true>
__device__ __forceinline__ void reduce_masked(double (&x)[8], double p, double pinv) {
    if constexpr (MASK != 0u)
        reduce_masked_impl<MASK, PIN>(x, p, pinv, std::make_integer_sequence<int, gs_popcount(MASK)>{});
}

// Register pairs and twiddle slots of one radix-2 stage on the 8 registers of a lane.
// D = 4, 2, 1 is the register distance; butterfly j in 0..3 pairs (lo(j), lo(j) + D) and uses
// slot(j) of a Tw7-ordered twiddle array: [0] for D = 4, [1 + h] for D = 2, [3 + g] for D = 1.
template <int D>
__device__ __forceinline__ constexpr int stage_lo(int j) { return (j / D) * 2 * D + (j % D); }
template <int D>
__device__ __forceinline__ constexpr int stage_slot(int j) { return D == 4 ? 0 : (D == 2 ? 1 + j / 2 : 3 + j); }

// one Cooley-Tukey stage (4 butterflies), phase-ordered
template <int D, bool REDUCE_A>
__device__ __forceinline__ void ct_stage_pinned(double (&x)[8], const double* w, const double* wp, double p,
                                                double pinv) {
    double a[4], b[4], ww[4], wwp[4], t[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        a[j] = x[stage_lo<D>(j)];
        b[j] = x[stage_lo<D>(j) + D];
        ww[j] = w[stage_slot<D>(j)];
        wwp[j] = wp[stage_slot<D>(j)];
    }
    if (REDUCE_A) reduce_n<4>(a, p, pinv);
    mulmod_tw_n<4>(t, b, ww, wwp, p);
#pragma unroll
    for (int j = 0; j < 4; j++) {
        x[stage_lo<D>(j) + D] = a[j] - t[j];
        x[stage_lo<D>(j)] = a[j] + t[j];
    }
    pin();
}

// one Gentleman-Sande stage, phase-ordered
template <int D>
__device__ __forceinline__ void gs_stage_pinned(double (&x)[8], const double* w, const double* wp, double p) {
    double d[4], ww[4], wwp[4], t[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const double lo = x[stage_lo<D>(j)], hi = x[stage_lo<D>(j) + D];
        x[stage_lo<D>(j)] = lo + hi;
        d[j] = lo - hi;
        ww[j] = w[stage_slot<D>(j)];
        wwp[j] = wp[stage_slot<D>(j)];
    }
    pin();
    mulmod_tw_n<4>(t, d, ww, wwp, p);
#pragma unroll
    for (int j = 0; j < 4; j++) x[stage_lo<D>(j) + D] = t[j];
}

template <bool REDUCE_A>
__device__ __forceinline__ void ct8_pinned(double (&x)[8], const double* w, const double* wp, double p, double pinv) {
    ct_stage_pinned<4, REDUCE_A>(x, w, wp, p, pinv);
    ct_stage_pinned<2, false>(x, w, wp, p, pinv);
    ct_stage_pinned<1, false>(x, w, wp, p, pinv);
}

__device__ __forceinline__ void gs8_pinned(double (&x)[8], const double* w, const double* wp, double p) {
    gs_stage_pinned<1>(x, w, wp, p);
    gs_stage_pinned<2>(x, w, wp, p);
    gs_stage_pinned<4>(x, w, wp, p);
}

// Group G (2 = layout C, 1 = layout B, 0 = layout A) of the length-512 inverse transform with the
// reductions of its plan: those before each stage and, for G > 0, those before the transposition
// that follows.  Outputs of the transform: below 3.2 p.
constexpr GsPlan kGsPlan512 = make_gs_plan(3, 3);
static_assert(kGsPlan512.out <= 4.0, "to_torus assumes outputs below 4 p");
template <int G>
__device__ __forceinline__ void gs8_planned(double (&x)[8], const double* w, const double* wp, double p, double pinv) {
    reduce_masked<kGsPlan512.pre[G][0]>(x, p, pinv);
    gs_stage_pinned<1>(x, w, wp, p);
    reduce_masked<kGsPlan512.pre[G][1]>(x, p, pinv);
    gs_stage_pinned<2>(x, w, wp, p);
    reduce_masked<kGsPlan512.pre[G][2]>(x, p, pinv);
    gs_stage_pinned<4>(x, w, wp, p);
    reduce_masked<kGsPlan512.carry[G]>(x, p, pinv);
}

__device__ __forceinline__ void wave_lds_fence() {
    // LDS instructions of one wave execute in issue order; this only stops the compiler
    // from moving them across the exchange point.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// -DTFHE_AMD_TR_SETPRIO=1: raised issue priority (s_setprio 3) from the first store of an exchange to its
// last load.  Measured on 8,192 gates, same box, alternating runs: 44.53 / 44.18 ms against 44.16 / 44.00 ms
// without, i.e. nothing gained: the partner wave's vector instructions do not compete for the LDS port.
#ifndef TFHE_AMD_TR_SETPRIO
#define TFHE_AMD_TR_SETPRIO 0
#endif
struct TrPrio {
    __device__ __forceinline__ TrPrio() { if (TFHE_AMD_TR_SETPRIO) __builtin_amdgcn_s_setprio(3); }
    __device__ __forceinline__ ~TrPrio() { if (TFHE_AMD_TR_SETPRIO) __builtin_amdgcn_s_setprio(0); }
};

// Slots of the exchange between layouts A and B.  Default (0): the pad slot per 8 positions of pad_pos,
// shared with the exchange between B and C; it puts lanes 0..2 and 29..31 of a layout-A group on the
// same banks (2-way conflicts: 240 extra LDS cycles per gate-step, the 1.25e9 SQ_LDS_BANK_CONFLICT per
// 8,192 gates).  -DTFHE_AMD_TR_PAD_AB=1: a pad run of 8 slots per 64 positions (slot = j + 8 (j >> 6)),
// free of conflicts on both sides.  Measured, same box: 8,192 gates 43.3 vs 43.5 ms (equal), 2,048 gates
// 11.9 vs 11.7 ms, 1,024 gates (one wave per SIMD) 7.5 vs 7.0 ms -- the conflict-free form is slower
// where a wave has its SIMD to itself, so the older pad stays: the conflicts cost LDS cycles the array
// has to spare, while the other form changes how the compiler pairs the 8-byte accesses.
#ifndef TFHE_AMD_TR_PAD_AB
#define TFHE_AMD_TR_PAD_AB 0
#endif
__device__ __forceinline__ int ab_slot_a(int k, int lane) {
    return TFHE_AMD_TR_PAD_AB ? 72 * k + lane : pad_pos(64 * k + lane);
}
__device__ __forceinline__ int ab_slot_b(int k, int lane) {
    return 72 * (lane >> 3) + (lane & 7) + (TFHE_AMD_TR_PAD_AB ? 8 : 9) * k;
}

__device__ __forceinline__ void transpose_a_to_b(double (&x)[8], double* tr, int lane) {
#ifdef TFHE_AMD_EXPERIMENT_NO_TRANSPOSE
    return;  // timing experiment (wrong results): what the exchanges cost
#endif
    TrPrio prio;
#pragma unroll
    for (int k = 0; k < 8; k++) tr[ab_slot_a(k, lane)] = x[k];
    wave_lds_fence();
#pragma unroll
    for (int k = 0; k < 8; k++) x[k] = tr[ab_slot_b(k, lane)];
    wave_lds_fence();
}
__device__ __forceinline__ void transpose_b_to_c(double (&x)[8], double* tr, int lane) {
#ifdef TFHE_AMD_EXPERIMENT_NO_TRANSPOSE
    return;  // timing experiment (wrong results): what the exchanges cost
#endif
    TrPrio prio;
    const int base = 72 * (lane >> 3) + (lane & 7);
#pragma unroll
    for (int k = 0; k < 8; k++) tr[base + 9 * k] = x[k];
    wave_lds_fence();
#pragma unroll
    for (int k = 0; k < 8; k++) x[k] = tr[9 * lane + k];
    wave_lds_fence();
}
__device__ __forceinline__ void transpose_c_to_b(double (&x)[8], double* tr, int lane) {
#ifdef TFHE_AMD_EXPERIMENT_NO_TRANSPOSE
    return;  // timing experiment (wrong results): what the exchanges cost
#endif
    TrPrio prio;
#pragma unroll
    for (int k = 0; k < 8; k++) tr[9 * lane + k] = x[k];
    wave_lds_fence();
    const int base = 72 * (lane >> 3) + (lane & 7);
#pragma unroll
    for (int k = 0; k < 8; k++) x[k] = tr[base + 9 * k];
    wave_lds_fence();
}
__device__ __forceinline__ void transpose_b_to_a(double (&x)[8], double* tr, int lane) {
#ifdef TFHE_AMD_EXPERIMENT_NO_TRANSPOSE
    return;  // timing experiment (wrong results): what the exchanges cost
#endif
    TrPrio prio;
#pragma unroll
    for (int k = 0; k < 8; k++) tr[ab_slot_b(k, lane)] = x[k];
    wave_lds_fence();
#pragma unroll
    for (int k = 0; k < 8; k++) x[k] = tr[ab_slot_a(k, lane)];
    wave_lds_fence();
}

// The first two stages of a forward transform of 8-bit gadget digits (|x| <= 2^7), on the eight
// registers of a lane (pairs (k, k + 4), then (k, k + 2)), with no modular reduction at all.  With
// I = zeta^2 and zeta^4 = -1 the twiddles are I (stage 1), zeta (stage 2, registers 0..3) and zeta^3
// (stage 2, registers 4..7).  Stage 1 gives x_k +- I x_(k+4), below 2^32.4.  The zeta^3 products are
// taken from the original digits: zeta^3 (x_2 - I x_6) = zeta (I x_2 + x_6), so every value stays an
// exact integer below 2^45.2 and the two butterflies that would need a full modular product (six
// instructions each) cost one fused multiply-add per output instead.
__device__ __forceinline__ void small_first_two_stages(double (&x)[8], const NttUniform& u) {
    const double d2 = x[2], d3 = x[3], d6 = x[6], d7 = x[7];
#pragma unroll
    for (int k = 0; k < 2; k++) {
        const double a = x[k], b = x[k + 4];
        x[k] = __builtin_fma(b, u.zI, a);
        x[k + 4] = __builtin_fma(-b, u.zI, a);
    }
    x[2] = __builtin_fma(d6, u.zI, d2);
    x[3] = __builtin_fma(d7, u.zI, d3);
    const double u2 = __builtin_fma(d2, u.zI, d6), u3 = __builtin_fma(d3, u.zI, d7);
    pin();
#pragma unroll
    for (int k = 0; k < 2; k++) {
        const double a = x[k], b = x[k + 2];
        x[k] = __builtin_fma(b, u.z, a);
        x[k + 2] = __builtin_fma(-b, u.z, a);
    }
    const double a4 = x[4], a5 = x[5];
    x[4] = __builtin_fma(u2, u.z, a4);
    x[6] = __builtin_fma(-u2, u.z, a4);
    x[5] = __builtin_fma(u3, u.z, a5);
    x[7] = __builtin_fma(-u3, u.z, a5);
    pin();
}

// Forward transform.  In: layout A, |x| <= 2^31 (SMALL: |x| <= 2^7, 8-bit gadget digits).
// Out: layout C (bit-reversed spectrum order); |x| <= p/2 with FINAL_REDUCE, else |x| <= 3.2 p.
//
// Growth between reductions, in units of p (a CT butterfly maps bound B to
// B + 0.5 + 0.75 B / 5.26): after the layout-A stages <= 1.72, after layout B <= 4.29,
// the distance-4 stage reduces its additive inputs, after layout C <= 3.2 < 5.26.
// The same growth recomputed by the compiler, with the bounds the callers rely on: outputs <= 3.2 p, a
// product with a key value (<= p/2) <= 1.41 p, three products on top of a reduced accumulator <= 4.74 p.
struct ForwardBounds {
    double after_a, after_b, out, product, mac;
    bool ok;
};
constexpr ForwardBounds forward_bounds() {
    ForwardBounds r{};
    r.ok = true;
    auto tw = [&](double d) { r.ok = r.ok && d <= kGsTwo53; return 0.5 + 0.75 * d / kGsTwo53 + kGsEps; };
    double b = 0.000002;  // inputs up to 2^31 = 1.3e-6 p
    for (int stage = 1; stage <= 9; stage++) {
        const double a = stage == 7 ? 0.5 + kGsEps : b;  // ct8<true>: the distance-4 stage reduces its additive inputs
        b = a + tw(b);
        if (stage == 3) r.after_a = b;
        if (stage == 6) r.after_b = b;
    }
    r.out = b;
    r.product = 0.5 + 1.5 * r.out / kGsTwo53 + kGsEps;
    r.mac = 0.5 + kGsEps + 3.0 * r.product;
    r.ok = r.ok && r.out <= kGsTwo53 && r.mac <= kGsTwo53;
    return r;
}
constexpr ForwardBounds kForwardBounds = forward_bounds();
static_assert(kForwardBounds.ok && kForwardBounds.after_a <= 1.73 && kForwardBounds.after_b <= 4.3 &&
                  kForwardBounds.out <= 3.2 && kForwardBounds.product <= 1.42 && kForwardBounds.mac <= 4.75,
              "the growth quoted for the forward transform and the multiply-accumulates no longer holds");
template <bool SMALL, bool FINAL_REDUCE, bool PIN = false>
__device__ __forceinline__ void ntt512_forward(double (&x)[8], const NttUniform& u, const d2* s_tw,
                                               double* tr, int lane) {
    const double p = u.p, pinv = u.pinv;
    if (PIN) {
        // same arithmetic as below, butterflies issued phase by phase
        if (SMALL) {
            small_first_two_stages(x, u);
            ct_stage_pinned<1, false>(x, u.fw + 1, u.fwp + 1, p, pinv);
        } else {
            ct8_pinned<false>(x, u.fw + 1, u.fwp + 1, p, pinv);
        }
        transpose_a_to_b(x, tr, lane);
        Tw7 t;
        load_tw7(t, s_tw + kTwFwdB + 7 * (lane >> 3));
        ct8_pinned<false>(x, t.w, t.wp, p, pinv);
        transpose_b_to_c(x, tr, lane);
        load_tw7(t, s_tw + kTwFwdC + 7 * lane);
        ct8_pinned<true>(x, t.w, t.wp, p, pinv);
        if (FINAL_REDUCE) reduce_n<8>(x, p, pinv);
        return;
    }
    if (SMALL) {
        small_first_two_stages(x, u);
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

    transpose_a_to_b(x, tr, lane);
    Tw7 t;
    load_tw7(t, s_tw + kTwFwdB + 7 * (lane >> 3));
    ct8<false>(x, t, p, pinv);

    transpose_b_to_c(x, tr, lane);
    load_tw7(t, s_tw + kTwFwdC + 7 * lane);
    ct8<true>(x, t, p, pinv);
    if (FINAL_REDUCE) {
#pragma unroll
        for (int k = 0; k < 8; k++) x[k] = reduce(x[k], p, pinv);
    }
}

#ifndef TFHE_AMD_TW_EARLY
#define TFHE_AMD_TW_EARLY 1
#endif

// First stage group of the forward transform of 8-bit digits (layout A), phase-ordered: the
// distance-256 and distance-128 stages need no reduction (small_first_two_stages).
__device__ __forceinline__ void ct8_small_first(double (&x)[8], const NttUniform& u) {
    small_first_two_stages(x, u);
    ct_stage_pinned<1, false>(x, u.fw + 1, u.fwp + 1, u.p, u.pinv);
}

// Two forward transforms of 8-bit digit polynomials, interleaved group by group so that one's
// butterflies run while the other's transposition is in flight (tra, trb: a buffer each).  Same
// arithmetic and bounds as ntt512_forward<true, false, true>.  Between the last group of a and the
// last group of b the caller's `between(a)` runs: a is complete there, b's transposition in flight.
template <typename Between>
__device__ __forceinline__ void ntt512_forward2_small(double (&a)[8], double (&b)[8], const NttUniform& u, const d2* s_tw,
                                                      double* tra, double* trb, int lane, Between between) {
    const double p = u.p, pinv = u.pinv;
    // LDS instructions of a wave complete in issue order.  A stage group's twiddle row is therefore
    // requested BEFORE the partner's exchange is issued (TFHE_AMD_TW_EARLY): requested behind it, the
    // row would arrive only after that exchange has completed, and the group that is meant to cover
    // the exchange would wait for it instead.
    ct8_small_first(a, u);
    transpose_a_to_b(a, tra, lane);
    Tw7 t, t2;
    if (TFHE_AMD_TW_EARLY) load_tw7(t, s_tw + kTwFwdB + 7 * (lane >> 3));
    ct8_small_first(b, u);
    transpose_a_to_b(b, trb, lane);
    if (!TFHE_AMD_TW_EARLY) load_tw7(t, s_tw + kTwFwdB + 7 * (lane >> 3));
    ct8_pinned<false>(a, t.w, t.wp, p, pinv);
    transpose_b_to_c(a, tra, lane);
    if (TFHE_AMD_TW_EARLY) load_tw7(t2, s_tw + kTwFwdC + 7 * lane);
    ct8_pinned<false>(b, t.w, t.wp, p, pinv);
    transpose_b_to_c(b, trb, lane);
    if (!TFHE_AMD_TW_EARLY) load_tw7(t2, s_tw + kTwFwdC + 7 * lane);
    ct8_pinned<true>(a, t2.w, t2.wp, p, pinv);
    between();
    ct8_pinned<true>(b, t2.w, t2.wp, p, pinv);
}

// Two inverse transforms interleaved the same way; arithmetic and bounds of ntt512_inverse<true>.
__device__ __forceinline__ void ntt512_inverse2(double (&a)[8], double (&b)[8], const NttUniform& u, const d2* s_tw,
                                                double* tra, double* trb, int lane) {
    const double p = u.p, pinv = u.pinv;
    Tw7 t, t2;
    load_tw7(t, s_tw + kTwInvC + 7 * lane);
    gs8_planned<2>(a, t.w, t.wp, p, pinv);
    transpose_c_to_b(a, tra, lane);
    if (TFHE_AMD_TW_EARLY) load_tw7(t2, s_tw + kTwInvB + 7 * (lane >> 3));  // ahead of b's exchange (see above)
    gs8_planned<2>(b, t.w, t.wp, p, pinv);
    transpose_c_to_b(b, trb, lane);
    if (!TFHE_AMD_TW_EARLY) load_tw7(t2, s_tw + kTwInvB + 7 * (lane >> 3));
    gs8_planned<1>(a, t2.w, t2.wp, p, pinv);
    transpose_b_to_a(a, tra, lane);
    gs8_planned<1>(b, t2.w, t2.wp, p, pinv);
    transpose_b_to_a(b, trb, lane);
    gs8_planned<0>(a, u.iw + 1, u.iwp + 1, p, pinv);
    gs8_planned<0>(b, u.iw + 1, u.iwp + 1, p, pinv);
}

// ntt512_inverse<true> with two pieces of the caller's independent work placed where a
// transposition's loads are in flight (after1 behind the first transposition, after2 behind the
// second), so that a lone transform does not sit out the exchange latency.
template <typename A1, typename A2>
__device__ __forceinline__ void ntt512_inverse_overlapped(double (&x)[8], const NttUniform& u, const d2* s_tw, double* tr,
                                                          int lane, A1 after1, A2 after2) {
    const double p = u.p, pinv = u.pinv;
    Tw7 t;
    load_tw7(t, s_tw + kTwInvC + 7 * lane);
    gs8_planned<2>(x, t.w, t.wp, p, pinv);
    transpose_c_to_b(x, tr, lane);
    load_tw7(t, s_tw + kTwInvB + 7 * (lane >> 3));
    after1();
    gs8_planned<1>(x, t.w, t.wp, p, pinv);
    transpose_b_to_a(x, tr, lane);
    after2();
    gs8_planned<0>(x, u.iw + 1, u.iwp + 1, p, pinv);
}

// Inverse transform without the 1/N factor.  In: layout C, |x| < 2^53.  Out: layout A,
// |x| <= 4 p.  Every group of three GS stages starts from |x| <= p/2 and doubles the bound
// of the sum outputs at most three times.
template <bool PIN = false>
__device__ __forceinline__ void ntt512_inverse(double (&x)[8], const NttUniform& u, const d2* s_tw,
                                               double* tr, int lane) {
    const double p = u.p, pinv = u.pinv;
    Tw7 t;
    if (PIN) {
        load_tw7(t, s_tw + kTwInvC + 7 * lane);
        gs8_planned<2>(x, t.w, t.wp, p, pinv);
        transpose_c_to_b(x, tr, lane);
        load_tw7(t, s_tw + kTwInvB + 7 * (lane >> 3));
        gs8_planned<1>(x, t.w, t.wp, p, pinv);
        transpose_b_to_a(x, tr, lane);
        gs8_planned<0>(x, u.iw + 1, u.iwp + 1, p, pinv);
        return;
    }
#pragma unroll
    for (int k = 0; k < 8; k++) x[k] = reduce(x[k], p, pinv);
    load_tw7(t, s_tw + kTwInvC + 7 * lane);
    gs8(x, t, p);

    transpose_c_to_b(x, tr, lane);
#pragma unroll
    for (int k = 0; k < 8; k++) x[k] = reduce(x[k], p, pinv);
    load_tw7(t, s_tw + kTwInvB + 7 * (lane >> 3));
    gs8(x, t, p);

    transpose_b_to_a(x, tr, lane);
#pragma unroll
    for (int k = 0; k < 8; k++) x[k] = reduce(x[k], p, pinv);
#pragma unroll
    for (int g = 0; g < 4; g++) bf_gs(x[2 * g], x[2 * g + 1], u.iw[4 + g], u.iwp[4 + g], p);
#pragma unroll
    for (int h = 0; h < 2; h++)
#pragma unroll
        for (int k = 0; k < 2; k++) bf_gs(x[4 * h + k], x[4 * h + k + 2], u.iw[2 + h], u.iwp[2 + h], p);
#pragma unroll
    for (int k = 0; k < 4; k++) bf_gs(x[k], x[k + 4], u.iw[1], u.iwp[1], p);
}

// Centred residue of x (|x| <= 4p, true value known to lie at least 2^43 away from +-p/2)
// reduced mod 2^32: adding 1.5 * 2^52 leaves the two's-complement integer in the mantissa.
__device__ __forceinline__ uint32_t to_torus(double x, double p, double pinv) {
    const double r = reduce(x, p, pinv);
    return (uint32_t)__double2loint(r + 6755399441055744.0);
}

}  // namespace tfhe_amd
