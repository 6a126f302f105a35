// One wave evaluates the 256-point complex transform of the reference's FFT product
// (reference src/types.hpp:152-197) with exactly the reference's floating-point operations:
// stage i multiplies the upper half of block j by omega_table[bitrev(2j)] as (ac - bd, ad + bc) and
// forms even + odd, even - odd.  The butterflies of a stage touch disjoint pairs, so the order in
// which lanes run them does not matter; only the operations inside one butterfly do.
//
// 4 complex points per lane, held as separate real and imaginary registers; the position maps and
// the padded wave-private LDS transposes are those of ntt512_half.hpp (a 256-point, 8-stage
// network of the same shape): layouts 1..4 cover stages (0,1), (2,3), (4,5), (6,7).  After the last
// stage the element at position m is the output of natural index bitrev8(m) (the reference ends
// with that permutation; here it is folded into what reads the result).
//
// Twiddle rows: group g = 1..4 starts at stage i0 = 2 (g - 1); a lane's block there is
// U = its position >> (8 - i0); row U = [T[brev8(2U)], T[brev8(4U)], T[brev8(4U + 2)]] for the
// direction's table T (the omega powers, or the omega powers reversed from index 1).
#pragma once

#include "ntt512_half.hpp"

namespace tfhe_amd {

struct Cx4 {
    double re[4], im[4];
};

constexpr int kFftRows = 1 + 4 + 16 + 64;  // rows per direction
__host__ __device__ constexpr int fft_row_offset(int g) { return g == 1 ? 0 : (g == 2 ? 1 : (g == 3 ? 5 : 21)); }
// table block appended to the [4][256] base tables of the replica mode, in complex (d2) units:
//   forward rows [85][3] | reversed-omega rows [85][3]
constexpr int kFftFwdRows = 0, kFftInvRows = kFftRows * 3;
constexpr int kFftFastPairs = 2 * kFftRows * 3;

template <int G>
__device__ __forceinline__ int fft_row(int lane) {
    if (G == 1) return 0;
    if (G == 2) return lane >> 4;
    if (G == 3) return lane >> 2;
    return lane;
}

// reference: odd *= w; lo = even + odd; hi = even - odd
__device__ __forceinline__ void cx_butterfly(double& ar, double& ai, double& br, double& bi, d2 w) {
    const double ac = br * w.x, bd = bi * w.y, ad = br * w.y, bc = bi * w.x;
    const double orr = ac - bd, oi = ad + bc;
    br = ar - orr;
    bi = ai - oi;
    ar = ar + orr;
    ai = ai + oi;
}

// Two butterflies at once, written phase by phase (all eight products, then the four sums of
// products, then the eight outputs) and fenced with scheduling barriers so that dependent FP64
// instructions end up four or eight apart (see the phase-ordered forms in ntt512.hpp).  The
// operations of each butterfly are the ones cx_butterfly performs, in the same order.
__device__ __forceinline__ void cx_butterfly2(double& a0r, double& a0i, double& b0r, double& b0i, d2 w0, double& a1r,
                                              double& a1i, double& b1r, double& b1i, d2 w1) {
    const double ac0 = b0r * w0.x, bd0 = b0i * w0.y, ad0 = b0r * w0.y, bc0 = b0i * w0.x;
    const double ac1 = b1r * w1.x, bd1 = b1i * w1.y, ad1 = b1r * w1.y, bc1 = b1i * w1.x;
    pin();
    const double or0 = ac0 - bd0, oi0 = ad0 + bc0, or1 = ac1 - bd1, oi1 = ad1 + bc1;
    pin();
    b0r = a0r - or0; b0i = a0i - oi0; b1r = a1r - or1; b1i = a1i - oi1;
    a0r = a0r + or0; a0i = a0i + oi0; a1r = a1r + or1; a1i = a1i + oi1;
    pin();
}

// two stages on the 4 points of a lane: pairs (0,2), (1,3) with row[0], then (0,1) with row[1], (2,3) with row[2]
__device__ __forceinline__ void cx_group(Cx4& x, const d2* row) {
    const d2 w0 = row[0], w1 = row[1], w2 = row[2];
    cx_butterfly2(x.re[0], x.im[0], x.re[2], x.im[2], w0, x.re[1], x.im[1], x.re[3], x.im[3], w0);
    cx_butterfly2(x.re[0], x.im[0], x.re[1], x.im[1], w1, x.re[2], x.im[2], x.re[3], x.im[3], w2);
}

// The first group (stages 0 and 1): block 0 of every table entry is T[0] = (1, 0) exactly, and
// multiplying by it returns the operand unchanged (b.re * 1 - b.im * 0, b.re * 0 + b.im * 1; only
// the sign of a zero can differ, which no later operation or the final conversion can see), so
// three of the four butterflies are additions only.  w23 = T[bitrev8(2)] = T[64], wave-uniform.
__device__ __forceinline__ void cx_group_first(Cx4& x, d2 w23) {
    double s, d;
#define TFHE_AMD_ADDSUB(a, b) s = a + b; d = a - b; a = s; b = d;
    TFHE_AMD_ADDSUB(x.re[0], x.re[2]) TFHE_AMD_ADDSUB(x.im[0], x.im[2])
    TFHE_AMD_ADDSUB(x.re[1], x.re[3]) TFHE_AMD_ADDSUB(x.im[1], x.im[3])
    TFHE_AMD_ADDSUB(x.re[0], x.re[1]) TFHE_AMD_ADDSUB(x.im[0], x.im[1])
#undef TFHE_AMD_ADDSUB
    cx_butterfly(x.re[2], x.im[2], x.re[3], x.im[3], w23);
}

// Layout 1 <-> layout 2 without the LDS (swap_half_waves, swap_rows: ntt512_half.hpp), real and
// imaginary parts alike.
__device__ __forceinline__ void cx_transpose_1_2(Cx4& x) {
    swap_half_waves(x.re[0], x.re[2]); swap_half_waves(x.im[0], x.im[2]);
    swap_half_waves(x.re[1], x.re[3]); swap_half_waves(x.im[1], x.im[3]);
    swap_rows(x.re[0], x.re[1]); swap_rows(x.im[0], x.im[1]);
    swap_rows(x.re[2], x.re[3]); swap_rows(x.im[2], x.im[3]);
}

// Transposition between layouts through the wave's LDS buffer tr (kHalfTrWords complex slots, real and
// imaginary part side by side: one 16-byte access per point, which a 16-byte read moves at twice the
// bytes per LDS cycle of the paired 8-byte reads the compiler makes of two planes; 69.7 -> 68.6 ms).
// The pads of half_pad also keep the 16-lane groups of a 16-byte read and the 8-lane groups of a
// 16-byte write on different slots.  The compiler waits for the loads only where their values are
// first used, so independent work placed after the call overlaps the exchange.
template <int FROM, int TO>
__device__ __forceinline__ void cx_transpose(Cx4& x, d2* tr, int lane) {
    constexpr int T = FROM < TO ? FROM : TO;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        d2 v;
        v.x = x.re[k];
        v.y = x.im[k];
        tr[half_pad<T>(half_pos<FROM>(lane, k))] = v;
    }
    wave_lds_fence();
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const d2 v = tr[half_pad<T>(half_pos<TO>(lane, k))];
        x.re[k] = v.x;
        x.im[k] = v.y;
    }
    wave_lds_fence();
}

// Two independent transforms of the same direction, interleaved group by group: while one's
// transposition is in flight in the LDS the other's butterflies run, so the wave does not sit out
// the exchange latency.  In: layout 1 (position 64 k + lane).  Out: layout 4 (position 4 lane + k),
// i.e. natural index bitrev8(4 lane + k).  rows: this direction's [85][3] table in LDS; w23 = its
// row 0, entry 2; row2 = the lane's row of group 2; tra, trb: one transposition buffer each.
__device__ __forceinline__ void fft256_wave2(Cx4& a, Cx4& b, const d2* rows, d2 w23, const d2* row2, d2* tra,
                                             d2* trb, int lane) {
    const d2* p3 = rows + 3 * (fft_row_offset(3) + fft_row<3>(lane));
    const d2* p4 = rows + 3 * (fft_row_offset(4) + fft_row<4>(lane));
    cx_group_first(a, w23);
    cx_transpose_1_2(a);
    const d2 r2[3] = {row2[0], row2[1], row2[2]};
    cx_group_first(b, w23);
    cx_transpose_1_2(b);
    cx_group(a, r2);
    cx_transpose<2, 3>(a, tra, lane);
    const d2 r3[3] = {p3[0], p3[1], p3[2]};
    cx_group(b, r2);
    cx_transpose<2, 3>(b, trb, lane);
    cx_group(a, r3);
    cx_transpose<3, 4>(a, tra, lane);
    const d2 r4[3] = {p4[0], p4[1], p4[2]};
    cx_group(b, r3);
    cx_transpose<3, 4>(b, trb, lane);
    cx_group(a, r4);
    cx_group(b, r4);
}

// The same network run from the other end of the lane index, for a transform whose INPUT sits in
// layout 4 with position p holding input index bitrev8(p) -- which is how fft256_wave2 leaves its
// output, so the second transform of a product needs no permutation in between.  Stage i pairs
// input indices that differ in bit 7 - i, i.e. positions that differ in bit i: stages 0, 1 are
// register bits of layout 4, stages 2, 3 of layout 3, 4, 5 of layout 2, 6, 7 of layout 1, the low
// register bit first.  A lane's block at the first stage of group g is the reversal of the low
// 2 (g - 1) bits of its position.  In place, the output with index o ends at input index
// bitrev8(o), i.e. at position o: the result is in natural order in layout 1 (position 64 k + lane).
__device__ __forceinline__ void cx_group_rev(Cx4& x, const d2* row) {
    const d2 w0 = row[0], w1 = row[1], w2 = row[2];
    cx_butterfly2(x.re[0], x.im[0], x.re[1], x.im[1], w0, x.re[2], x.im[2], x.re[3], x.im[3], w0);
    cx_butterfly2(x.re[0], x.im[0], x.re[2], x.im[2], w1, x.re[1], x.im[1], x.re[3], x.im[3], w2);
}

__device__ __forceinline__ void cx_group_first_rev(Cx4& x, d2 w23) {
    double s, d;
#define TFHE_AMD_ADDSUB(a, b) s = a + b; d = a - b; a = s; b = d;
    TFHE_AMD_ADDSUB(x.re[0], x.re[1]) TFHE_AMD_ADDSUB(x.im[0], x.im[1])
    TFHE_AMD_ADDSUB(x.re[2], x.re[3]) TFHE_AMD_ADDSUB(x.im[2], x.im[3])
    TFHE_AMD_ADDSUB(x.re[0], x.re[2]) TFHE_AMD_ADDSUB(x.im[0], x.im[2])
#undef TFHE_AMD_ADDSUB
    cx_butterfly(x.re[1], x.im[1], x.re[3], x.im[3], w23);
}

__device__ __forceinline__ void fft256_wave2_rev(Cx4& a, Cx4& b, const d2* rows, d2 w23, d2* tra, d2* trb, int lane) {
    const int u2 = (int)(__brev((unsigned)lane & 3u) >> 30);
    const int u3 = (int)(__brev((unsigned)lane & 15u) >> 28);
    const int u4 = (int)(__brev((unsigned)lane) >> 26);
    const d2* p2 = rows + 3 * (fft_row_offset(2) + u2);
    const d2* p3 = rows + 3 * (fft_row_offset(3) + u3);
    const d2* p4 = rows + 3 * (fft_row_offset(4) + u4);
    cx_group_first_rev(a, w23);
    cx_transpose<4, 3>(a, tra, lane);
    const d2 r2[3] = {p2[0], p2[1], p2[2]};
    cx_group_first_rev(b, w23);
    cx_transpose<4, 3>(b, trb, lane);
    cx_group_rev(a, r2);
    cx_transpose<3, 2>(a, tra, lane);
    const d2 r3[3] = {p3[0], p3[1], p3[2]};
    cx_group_rev(b, r2);
    cx_transpose<3, 2>(b, trb, lane);
    cx_group_rev(a, r3);
    cx_transpose_1_2(a);  // its own inverse: layout 2 -> layout 1
    const d2 r4[3] = {p4[0], p4[1], p4[2]};
    cx_group_rev(b, r3);
    cx_transpose_1_2(b);
    cx_group_rev(a, r4);
    cx_group_rev(b, r4);
}

__device__ __forceinline__ int brev8(int v) { return (int)(__brev((unsigned)v) >> 24); }

// reference src/types.hpp:28-33 for a finite d: fmod(d, 1), plus 1 if negative, times 2^32,
// truncated through 64 bits (a value that rounds up to exactly 2^32 wraps to 0).  d - floor(d) is
// the same real number as fmod(d, 1) (+ 1 for negative d) rounded once, like the reference's sum;
// for d >= 0 and for negative integers both are exact.
__device__ __forceinline__ uint32_t torus_from_double_fast(double d) {
    const double f = d - __builtin_floor(d);
    const double t = f * 4294967296.0;
    return t >= 4294967296.0 ? 0u : (uint32_t)t;
}

// Sixteen conversions at once.  In full: f = d - floor(d) as above (round to nearest), then
// f + 1.5 * 2^20 with the FP64 rounding mode set to "toward zero" for just those additions: the sum
// lies in [1.5 * 2^20, 1.5 * 2^20 + 1], where a double's unit is 2^-32, so its low mantissa word is
// floor(f * 2^32) mod 2^32 (f = 1 gives 2^32, i.e. 0, like the reference's conversion through 64
// bits).  The mode is bits 3:2 of the MODE register and is restored inside the same instruction block.
//
// f itself is needed only for -1/2 < d < 0: everywhere else d - floor(d) is exact (for |d| >= 1/2 the
// fraction has no bit below 2^-53; for d >= 0 below 1 it is d), and then the truncated sum
// d + 1.5 * 2^20 already has floor(d * 2^32) mod 2^32 = floor(f * 2^32) in its low word (|d| < 2^19 here:
// at most N * 2^7 = 2^16).  Only in (-1/2, 0) can the reference's rounded 1 + d land on the next multiple
// of 2^-32.  Such values are rare (d spreads over +-2^15), so a wave tests for them first -- the high
// word of a double in [-0, -1/2) is, as a signed integer, below INT_MIN + 0x3FE00000 -- and takes the
// subtraction only when one of its 1,024 values needs it: one FP64 instruction per value instead of
// three.  Measured: 8,192 gates 68.5 -> 67.0 ms, 2,048 gates 19.3 -> 18.25 ms.  Checked against the
// reference's formula on the device by tests/hip/convert_check.hip.
// -DTFHE_AMD_REPLICA_FAST_CONVERT=0: the subtraction always.
#ifndef TFHE_AMD_REPLICA_FAST_CONVERT
#define TFHE_AMD_REPLICA_FAST_CONVERT 1
#endif
__device__ __forceinline__ void rtz_add_8(double (&d)[8], double c) {
    asm volatile(
        "s_setreg_imm32_b32 hwreg(HW_REG_MODE, 2, 2), 3\n\t"
        "s_nop 1\n\t"
        "v_add_f64 %0, %0, %8\n\t"
        "v_add_f64 %1, %1, %8\n\t"
        "v_add_f64 %2, %2, %8\n\t"
        "v_add_f64 %3, %3, %8\n\t"
        "v_add_f64 %4, %4, %8\n\t"
        "v_add_f64 %5, %5, %8\n\t"
        "v_add_f64 %6, %6, %8\n\t"
        "v_add_f64 %7, %7, %8\n\t"
        "s_setreg_imm32_b32 hwreg(HW_REG_MODE, 2, 2), 0\n\t"
        "s_nop 1"
        : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]), "+v"(d[6]), "+v"(d[7])
        : "s"(c));
}
// sixteen conversions (both parts of eight products); a and b are overwritten
__device__ __forceinline__ void torus_from_double_16(uint32_t (&oa)[8], uint32_t (&ob)[8], double (&a)[8], double (&b)[8]) {
    if (TFHE_AMD_REPLICA_FAST_CONVERT) {
        int m = min(__double2hiint(a[0]), __double2hiint(b[0]));
#pragma unroll
        for (int i = 1; i < 8; i++) m = min(m, min(__double2hiint(a[i]), __double2hiint(b[i])));
        // The rare path sits inside one instruction block, behind a branch the compiler does not see
        // (as control flow of the kernel it cost 108 bytes of spills at this kernel's 254 registers):
        // skipped unless some lane of the wave holds a value in [-0, -1/2).
        double t;
        asm volatile(
            "v_cmp_gt_i32 vcc, %18, %17\n\t"
            "s_nop 1\n\t"
            "s_cbranch_vccz .Ltfhe_cvt_skip_%=\n\t"
            "v_floor_f64 %16, %0\n\tv_add_f64 %0, %0, -%16\n\t"
            "v_floor_f64 %16, %1\n\tv_add_f64 %1, %1, -%16\n\t"
            "v_floor_f64 %16, %2\n\tv_add_f64 %2, %2, -%16\n\t"
            "v_floor_f64 %16, %3\n\tv_add_f64 %3, %3, -%16\n\t"
            "v_floor_f64 %16, %4\n\tv_add_f64 %4, %4, -%16\n\t"
            "v_floor_f64 %16, %5\n\tv_add_f64 %5, %5, -%16\n\t"
            "v_floor_f64 %16, %6\n\tv_add_f64 %6, %6, -%16\n\t"
            "v_floor_f64 %16, %7\n\tv_add_f64 %7, %7, -%16\n\t"
            "v_floor_f64 %16, %8\n\tv_add_f64 %8, %8, -%16\n\t"
            "v_floor_f64 %16, %9\n\tv_add_f64 %9, %9, -%16\n\t"
            "v_floor_f64 %16, %10\n\tv_add_f64 %10, %10, -%16\n\t"
            "v_floor_f64 %16, %11\n\tv_add_f64 %11, %11, -%16\n\t"
            "v_floor_f64 %16, %12\n\tv_add_f64 %12, %12, -%16\n\t"
            "v_floor_f64 %16, %13\n\tv_add_f64 %13, %13, -%16\n\t"
            "v_floor_f64 %16, %14\n\tv_add_f64 %14, %14, -%16\n\t"
            "v_floor_f64 %16, %15\n\tv_add_f64 %15, %15, -%16\n\t"
            ".Ltfhe_cvt_skip_%=:"
            : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]),
              "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(b[4]), "+v"(b[5]), "+v"(b[6]), "+v"(b[7]), "=&v"(t)
            : "v"(m), "s"((int)(0x80000000u + 0x3FE00000u))
            : "vcc");
    } else {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            a[i] = a[i] - __builtin_floor(a[i]);
            b[i] = b[i] - __builtin_floor(b[i]);
        }
    }
    const double c = 1572864.0;  // 1.5 * 2^20
    rtz_add_8(a, c);
    rtz_add_8(b, c);
#pragma unroll
    for (int i = 0; i < 8; i++) {
        oa[i] = (uint32_t)__double2loint(a[i]);
        ob[i] = (uint32_t)__double2loint(b[i]);
    }
}

}  // namespace tfhe_amd
