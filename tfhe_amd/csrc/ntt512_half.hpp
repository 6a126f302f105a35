// The length-512 negacyclic NTT of ntt512.hpp split over TWO waves, for the latency form of the
// blind rotation.
// Forward: after the first radix-2 stage (distance 256) the two halves of the vector are
// independent length-256 transforms, so wave h in {0, 1} computes stage 0 for its own half from
// the (cheap, 8-bit) inputs and then owns positions 256 h .. 256 h + 255: 4 coefficients per lane,
// four groups of two radix-2 stages in registers, two transposes through a private LDS buffer and
// one in registers.  Inverse: split by the parity of the position instead (second part of this file).
// Same field, same modular arithmetic, same spectrum order as ntt512.hpp: the half-h wave ends
// with spectrum positions 256 h + 4 lane + k, which in the full transform's layout C
// (position 8 l + k) are lane l = 32 h + (lane >> 1), registers 4 (lane & 1) + k.
//
// Lane/register <-> position maps inside a half (j' = 8-bit position, l = lane, k = register):
//     layout 1: j' = 64 k + l                          (distance 128, 64)
//     layout 2: j' = 64 (l >> 4) + 16 k + (l & 15)     (distance 32, 16)
//     layout 3: j' = 16 (l >> 2) + 4 k + (l & 3)       (distance 8, 4)
//     layout 4: j' = 4 l + k                           (distance 2, 1)
// Twiddles: group g = 1..4 starts at stage s = 2g - 1 with m = 2^s blocks; a lane reads row
// U = (block index of its positions at stage s) = m h / 2 + (lane bits above the group's register
// bits) of a table of rows [tw[m + U], tw[2m + 2U], tw[2m + 2U + 1]].
#pragma once

#include "ntt512.hpp"

namespace tfhe_amd {

struct Tw3 {
    double w[3], wp[3];  // [0] first stage of the group (both butterflies), [1..2] second stage
};

constexpr int kHalfRows = 2 + 8 + 32 + 128;  // groups 1..4, both halves
constexpr int kHalfTrWords = 320;            // private transpose buffer, doubles per wave

__host__ __device__ constexpr int half_row_offset(int g) { return g == 1 ? 0 : (g == 2 ? 2 : (g == 3 ? 10 : 42)); }

__device__ __forceinline__ void load_tw3(Tw3& t, const d2* row) {
#pragma unroll
    for (int i = 0; i < 3; i++) {
        const d2 v = row[i];
        t.w[i] = v.x;
        t.wp[i] = v.y;
    }
}

// table row of a lane for group g
template <int G>
__device__ __forceinline__ int half_row(int h, int lane) {
    if (G == 1) return h;
    if (G == 2) return 4 * h + (lane >> 4);
    if (G == 3) return 16 * h + (lane >> 2);
    return 64 * h + lane;
}

template <int LAYOUT>
__host__ __device__ constexpr int half_pos(int l, int k) {
    if (LAYOUT == 1) return 64 * k + l;
    if (LAYOUT == 2) return 64 * (l >> 4) + 16 * k + (l & 15);
    if (LAYOUT == 3) return 16 * (l >> 2) + 4 * k + (l & 3);
    return 4 * l + k;
}

// Padding of the transpose between layouts T and T + 1, chosen so that the 32 lanes of a pass
// of an 8-byte LDS access fall on 32 different bank pairs in both layouts.
template <int T>
__host__ __device__ constexpr int half_pad(int j) {
    if (T == 1) return j + 16 * (j >> 6);
    if (T == 2) return j + 4 * (j >> 4);
    return j + (j >> 2);
}

template <int FROM, int TO>
__device__ __forceinline__ void half_transpose(double (&x)[4], double* tr, int lane) {
    constexpr int T = FROM < TO ? FROM : TO;
#pragma unroll
    for (int k = 0; k < 4; k++) tr[half_pad<T>(half_pos<FROM>(lane, k))] = x[k];
    wave_lds_fence();
#pragma unroll
    for (int k = 0; k < 4; k++) x[k] = tr[half_pad<T>(half_pos<TO>(lane, k))];
    wave_lds_fence();
}

// Layout 1 <-> layout 2 without the LDS: the element of lane L, register K goes to lane
// 16 K + (L & 15), register L >> 4, a 4 x 4 transposition between the two register bits and lane
// bits 5 and 4, which the half-wave and row swaps of gfx950 do in place (one instruction per 32-bit
// register pair).  The transposition is its own inverse.
__device__ __forceinline__ void swap_half_waves(double& a, double& b) {  // a's lanes 32..63 <-> b's lanes 0..31
    const auto lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
    const auto hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
    a = __hiloint2double((int)hi[0], (int)lo[0]);
    b = __hiloint2double((int)hi[1], (int)lo[1]);
}
__device__ __forceinline__ void swap_rows(double& a, double& b) {  // a's odd 16-lane rows <-> b's even rows
    const auto lo = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
    const auto hi = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
    a = __hiloint2double((int)hi[0], (int)lo[0]);
    b = __hiloint2double((int)hi[1], (int)lo[1]);
}
__device__ __forceinline__ void half_transpose_1_2(double (&x)[4]) {
    swap_half_waves(x[0], x[2]);
    swap_half_waves(x[1], x[3]);
    swap_rows(x[0], x[1]);
    swap_rows(x[2], x[3]);
}

// The two butterflies of a stage are independent: they are issued phase by phase (ntt512.hpp) so
// that a wave does not run them as two serial chains -- the latency form has one or two waves per
// SIMD in its inverse phase and is bound by the dependent-issue latency there.
#ifndef TFHE_AMD_HALF_PIN
#define TFHE_AMD_HALF_PIN 1
#endif
#ifndef TFHE_AMD_HALF_TW_AHEAD
#define TFHE_AMD_HALF_TW_AHEAD 1
#endif
__device__ __forceinline__ void bf_ct2(double& a0, double& b0, double w0, double wp0, double& a1, double& b1, double w1,
                                       double wp1, double p) {
    if (!TFHE_AMD_HALF_PIN) {
        bf_ct(a0, b0, w0, wp0, p);
        bf_ct(a1, b1, w1, wp1, p);
        return;
    }
    const double b[2] = {b0, b1}, w[2] = {w0, w1}, wp[2] = {wp0, wp1};
    double t[2];
    mulmod_tw_n<2>(t, b, w, wp, p);
    b0 = a0 - t[0];
    b1 = a1 - t[1];
    a0 = a0 + t[0];
    a1 = a1 + t[1];
    pin();
}
__device__ __forceinline__ void bf_gs2(double& a0, double& b0, double w0, double wp0, double& a1, double& b1, double w1,
                                       double wp1, double p) {
    if (!TFHE_AMD_HALF_PIN) {
        bf_gs(a0, b0, w0, wp0, p);
        bf_gs(a1, b1, w1, wp1, p);
        return;
    }
    const double d[2] = {a0 - b0, a1 - b1}, w[2] = {w0, w1}, wp[2] = {wp0, wp1};
    a0 = a0 + b0;
    a1 = a1 + b1;
    pin();
    double t[2];
    mulmod_tw_n<2>(t, d, w, wp, p);
    b0 = t[0];
    b1 = t[1];
}

// two Cooley-Tukey stages on 4 registers: pairs (0,2), (1,3), then (0,1), (2,3)
template <bool REDUCE_A>
__device__ __forceinline__ void ct4(double (&x)[4], const Tw3& t, double p, double pinv) {
    if (REDUCE_A) {
        double a[2] = {x[0], x[1]};
        reduce_n<2>(a, p, pinv);
        x[0] = a[0];
        x[1] = a[1];
    }
    bf_ct2(x[0], x[2], t.w[0], t.wp[0], x[1], x[3], t.w[0], t.wp[0], p);
    bf_ct2(x[0], x[1], t.w[1], t.wp[1], x[2], x[3], t.w[2], t.wp[2], p);
}

__device__ __forceinline__ void gs4_near(double (&x)[4], const Tw3& t, double p) {  // pairs (0,1), (2,3)
    bf_gs2(x[0], x[1], t.w[1], t.wp[1], x[2], x[3], t.w[2], t.wp[2], p);
}
__device__ __forceinline__ void gs4_far(double (&x)[4], const Tw3& t, double p) {  // pairs (0,2), (1,3)
    bf_gs2(x[0], x[2], t.w[0], t.wp[0], x[1], x[3], t.w[0], t.wp[0], p);
}

// Forward transform of 8-bit digits, half h.  d[k] = digit at position 64 k + lane of the full
// vector (k = 0..7, the full transform's layout A).  Out: layout 4 of half h, |x| <= 2.2 p.
//
// Stages 0 and 1 need no reduction: with I = zeta^2 and the stage-1 twiddle W = zeta (h = 0) or
// zeta^3 (h = 1), u_k = d_k +- I d_(k+4) and W u_(k+2) = c1 d_(k+2) + c2 d_(k+6) with
// (c1, c2) = (zeta, zeta^3) or (zeta^3, zeta) (zeta^4 = -1), all below 2^46.
// Growth afterwards in units of p (a butterfly maps bound B to B + 0.5 + 0.75 B / 5.26):
// 0.55, 1.13, 1.79, 2.55, 3.41 after stages 2..6; stage 7 reduces its additive inputs: 1.49, 2.2.
__device__ __forceinline__ void half_forward_small(double (&x)[4], const double (&d)[8], int h, const NttUniform& u,
                                                   const d2* s_tw, double* tr, int lane) {
    const double p = u.p, pinv = u.pinv;
    const double z3 = u.z * u.zI;
    const double cA = h ? -u.zI : u.zI, c1 = h ? z3 : u.z, c2 = h ? u.z : z3;
#pragma unroll
    for (int k = 0; k < 2; k++) {
        const double a = __builtin_fma(d[k + 4], cA, d[k]);
        const double t = __builtin_fma(d[k + 2], c1, c2 * d[k + 6]);
        x[k] = a + t;
        x[k + 2] = a - t;
    }
    Tw3 t, t2;
    load_tw3(t, s_tw + 3 * (half_row_offset(1) + half_row<1>(h, lane)));
    if (TFHE_AMD_HALF_TW_AHEAD) {
        // LDS instructions complete in issue order: a group's twiddle row is requested one group ahead,
        // in front of the exchange that precedes the group, so that the group waits for the exchange only
        load_tw3(t2, s_tw + 3 * (half_row_offset(2) + half_row<2>(h, lane)));
        bf_ct2(x[0], x[1], t.w[1], t.wp[1], x[2], x[3], t.w[2], t.wp[2], p);
        half_transpose_1_2(x);
        load_tw3(t, s_tw + 3 * (half_row_offset(3) + half_row<3>(h, lane)));
        ct4<false>(x, t2, p, pinv);
        half_transpose<2, 3>(x, tr, lane);
        load_tw3(t2, s_tw + 3 * (half_row_offset(4) + half_row<4>(h, lane)));
        ct4<false>(x, t, p, pinv);
        half_transpose<3, 4>(x, tr, lane);
        ct4<true>(x, t2, p, pinv);
        return;
    }
    bf_ct2(x[0], x[1], t.w[1], t.wp[1], x[2], x[3], t.w[2], t.wp[2], p);

    half_transpose_1_2(x);
    load_tw3(t, s_tw + 3 * (half_row_offset(2) + half_row<2>(h, lane)));
    ct4<false>(x, t, p, pinv);

    half_transpose<2, 3>(x, tr, lane);
    load_tw3(t, s_tw + 3 * (half_row_offset(3) + half_row<3>(h, lane)));
    ct4<false>(x, t, p, pinv);

    half_transpose<3, 4>(x, tr, lane);
    load_tw3(t, s_tw + 3 * (half_row_offset(4) + half_row<4>(h, lane)));
    ct4<true>(x, t, p, pinv);
}

// ---- inverse transform split by the parity of the position -----------------------------------------
// A Gentleman-Sande inverse on the spectrum order above pairs positions (2i, 2i + 1) in its first stage
// (distance 1); every later stage has an even distance, so the sums (even positions) and the twiddled
// differences (odd positions) never meet again.  Wave e in {0, 1} therefore computes its own output
// of every first-stage butterfly from all 512 inputs and then runs the remaining eight stages
// (distances 2 .. 256) alone, on the 256 positions P = 2i + e; it ends with the finished time-domain
// coefficients of its parity, and no stage is left that pairs the two waves.
// Lane/register maps: the layouts 4 .. 1 above applied to i (layout 4: i = 4 lane + k, i.e. a lane
// takes the eight inputs 8 lane .. 8 lane + 7; layout 1: i = 64 k + lane, output position
// 128 k + 2 lane + e).  A stage of distance t has 256 / t blocks and block b uses inverse twiddle
// tab[256 / t + b], the same for both parities.  Table: [4][64] first-stage twiddles
// tab[256 + 4 lane + k] stored [k][lane], then rows [tab[m + U], tab[2m + 2U], tab[2m + 2U + 1]] for
// m = 64 (U = lane), 16 (U = lane >> 2), 4 (U = lane >> 4), 1 (U = 0).
constexpr int kParityRows = 64 + 16 + 4 + 1;
constexpr int kParityPairs = 256 + 3 * kParityRows;
constexpr int kHalfPairs = 3 * kHalfRows + kParityPairs;  // the whole table: forward rows, then the inverse part
__host__ __device__ constexpr int parity_row_offset(int m) { return m == 64 ? 0 : (m == 16 ? 64 : (m == 4 ? 80 : 84)); }

// Slot (in doubles) of first-stage pair i in a row of 256 partial sums or differences kept in LDS.  A
// forward wave stores i = 128 h + 2 lane + {0, 1} with one 16-byte access, an inverse wave loads
// i = 4 lane + {0, 1} and {2, 3} with two; two pad slots per 32 keep the loads of lanes l and l + 8,
// which are 256 bytes apart without them, on different banks.
__host__ __device__ constexpr int pair_slot(int i) { return i + 2 * (i >> 5); }
constexpr int kPairSlots = 256 + 16;

// Stages 2 .. 9 of the parity-split inverse.  In: layout 4 of i, |x| <= 1.22 p.  Out: layout 1,
// |x| <= 2.21 p.  Bounds in units of p (a butterfly maps (a, b) to (a + b, 0.5 + 0.75 (a + b) / 5.26);
// a transposition may hand a lane any register's value, so every group starts from the largest bound
// left): [1.22 x 4] -> near [2.44, .85, 2.44, .85] -> far [4.88, 1.7, 1.2, .74], register 0 reduced
// -> 1.7 -> near [3.4, .98, ..], registers 0 and 2 reduced -> far [1.0, 1.96, .64, .78] -> 1.96 ->
// near [3.92, 1.06, ..], reduced -> far [1.0, 2.1, .64, .8] -> 2.1 -> near [4.2, 1.1, ..], reduced ->
// far [1.0, 2.21, .64, .81].  (Recomputed at compile time: half_bounds() at the end of this file.)
__device__ __forceinline__ void reduce_0_2(double (&x)[4], double p, double pinv) {
    double a[2] = {x[0], x[2]};
    reduce_n<2>(a, p, pinv);
    x[0] = a[0];
    x[2] = a[1];
}
__device__ __forceinline__ void parity_inverse(double (&x)[4], const NttUniform& u, const d2* s_rows, double* tr,
                                               int lane) {
    const double p = u.p, pinv = u.pinv;
    Tw3 t;
    load_tw3(t, s_rows + 3 * (parity_row_offset(64) + lane));
    if (TFHE_AMD_HALF_TW_AHEAD) {  // every row one group ahead, as in half_forward_small
        Tw3 t2;
        load_tw3(t2, s_rows + 3 * (parity_row_offset(16) + (lane >> 2)));
        gs4_near(x, t, p);
        gs4_far(x, t, p);
        x[0] = reduce(x[0], p, pinv);
        half_transpose<4, 3>(x, tr, lane);
        load_tw3(t, s_rows + 3 * (parity_row_offset(4) + (lane >> 4)));
        gs4_near(x, t2, p);
        reduce_0_2(x, p, pinv);
        gs4_far(x, t2, p);
        half_transpose<3, 2>(x, tr, lane);
        load_tw3(t2, s_rows + 3 * parity_row_offset(1));
        gs4_near(x, t, p);
        reduce_0_2(x, p, pinv);
        gs4_far(x, t, p);
        half_transpose_1_2(x);
        gs4_near(x, t2, p);
        reduce_0_2(x, p, pinv);
        gs4_far(x, t2, p);
        return;
    }
    gs4_near(x, t, p);
    gs4_far(x, t, p);
    x[0] = reduce(x[0], p, pinv);

    half_transpose<4, 3>(x, tr, lane);
    load_tw3(t, s_rows + 3 * (parity_row_offset(16) + (lane >> 2)));
    gs4_near(x, t, p);
    reduce_0_2(x, p, pinv);
    gs4_far(x, t, p);

    half_transpose<3, 2>(x, tr, lane);
    load_tw3(t, s_rows + 3 * (parity_row_offset(4) + (lane >> 4)));
    gs4_near(x, t, p);
    reduce_0_2(x, p, pinv);
    gs4_far(x, t, p);

    half_transpose_1_2(x);
    load_tw3(t, s_rows + 3 * parity_row_offset(1));
    gs4_near(x, t, p);
    reduce_0_2(x, p, pinv);
    gs4_far(x, t, p);
}

// ---- the bounds quoted above, recomputed by the compiler -----------------------------------------------
// Units of p; kGsTwo53 = 5.26 stands for 2^53 / p (a little below it).  A twiddle product of an operand d
// leaves 0.5 + 0.75 d / 5.26, a product of two variables (one at most p/2) 0.5 + 1.5 x / 5.26, a
// reduction 0.5.  `ok` turns false as soon as a sum or a difference of two registers could pass 2^53.
struct HalfBounds {
    double forward_out, product, partial, running_sum, inverse_in, inverse_out;
    bool ok;
};
constexpr HalfBounds half_bounds() {
    HalfBounds r{};
    r.ok = true;
    const double eps = kGsEps;
    auto tw = [&](double d) { r.ok = r.ok && d <= kGsTwo53; return 0.5 + 0.75 * d / kGsTwo53 + eps; };
    // forward half: stages 0 and 1 stay below 2^46 (0.0001 p); stages 2 .. 8 are butterflies a +- tw(b);
    // stage 7 (the first of the last group, ct4<true>) reduces its additive inputs
    double b = 0.0001;
    for (int stage = 2; stage <= 8; stage++) {
        const double a = stage == 7 ? 0.5 + eps : b;
        b = a + tw(b);
    }
    r.forward_out = b;
    r.ok = r.ok && b <= kGsTwo53;
    r.product = 0.5 + 1.5 * b / kGsTwo53 + eps;
    r.partial = 2.0 * r.product;  // sum or difference of two partial products
    // phase 2: six partials, a reduction after every second one
    r.running_sum = 0.5 + eps + 2.0 * r.partial;
    r.ok = r.ok && r.running_sum <= kGsTwo53;
    r.inverse_in = tw(r.running_sum) > 0.5 + eps ? tw(r.running_sum) : 0.5 + eps;  // e = 1: twiddle product; e = 0: reduced
    // parity_inverse: four groups (near, far) on registers {0, 1, 2, 3}; reductions as in the code
    double x[4] = {r.inverse_in, r.inverse_in, r.inverse_in, r.inverse_in};
    for (int g = 0; g < 4; g++) {
        // near: pairs (0, 1), (2, 3)
        const double n01 = x[0] + x[1], n23 = x[2] + x[3];
        x[1] = tw(n01);
        x[3] = tw(n23);
        x[0] = n01;
        x[2] = n23;
        if (g > 0) x[0] = x[2] = 0.5 + eps;  // reduce_0_2
        // far: pairs (0, 2), (1, 3)
        const double f02 = x[0] + x[2], f13 = x[1] + x[3];
        x[2] = tw(f02);
        x[3] = tw(f13);
        x[0] = f02;
        x[1] = f13;
        if (g == 0) x[0] = 0.5 + eps;  // the one reduction before the first transposition
        double top = x[0];
        for (int k = 1; k < 4; k++) top = x[k] > top ? x[k] : top;
        if (g < 3)
            for (int k = 0; k < 4; k++) x[k] = top;  // a transposition may hand a lane any register's value
        r.inverse_out = top;
    }
    r.ok = r.ok && r.inverse_out <= kGsTwo53;
    return r;
}
constexpr HalfBounds kHalfBounds = half_bounds();
static_assert(kHalfBounds.ok, "a sum in the two-wave transforms of the latency form could pass 2^53");
static_assert(kHalfBounds.forward_out <= 2.2 && kHalfBounds.product <= 1.13 && kHalfBounds.running_sum <= 5.02 &&
                  kHalfBounds.inverse_in <= 1.22 && kHalfBounds.inverse_out <= 2.25,
              "the bounds quoted in the comments of the latency form no longer hold");

}  // namespace tfhe_amd
