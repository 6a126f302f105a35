/*
 * TEST INFRASTRUCTURE ONLY -- exact-integer CPU restatement of KowerKoint/TFHE's
 * HomNAND::nand path.  See tfhe_oracle.h for the rules on who may load it, the flat
 * layouts and how the restatement is pinned to the reference.
 *
 * Two interchangeable negacyclic multipliers are provided:
 *   TFO_MUL_SCHOOLBOOK  O(N^2) wrap-around arithmetic, the definition itself
 *                       (what the reference executes for N < 32, src/types.hpp:225-236);
 *   TFO_MUL_NTT         number-theoretic transform over p = 2^64 - 2^32 + 1, exact because
 *                       every true coefficient is far below p/2; checked against the
 *                       schoolbook form by the tests.
 */
#include "tfhe_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------ parameters */

void tfo_params_default(tfo_params* p) {
    p->n = 636;  p->N = 512;  p->K = 2;  p->bg_bit = 8;  p->L = 2;  p->T = 5;  p->base_bit = 2;
    p->alpha_lv0 = 0.0000925119974676756;   /* reference src/tlwe.hpp:13 */
    p->alpha_lv1 = 2.9802322387695312e-08;  /* reference src/trlwe.hpp:12 */
}

void tfo_params_n2048(tfo_params* p) {
    p->n = 800;  p->N = 2048;  p->K = 1;  p->bg_bit = 8;  p->L = 3;  p->T = 8;  p->base_bit = 2;
    p->alpha_lv0 = 0.0000925119974676756;
    p->alpha_lv1 = 2.9802322387695312e-08;
}

static int ilog2(int v) { int b = 0; while ((1 << b) < v) b++; return b; }

int tfo_params_valid(const tfo_params* p) {
    if (p->n < 1 || p->K < 1 || p->L < 1 || p->T < 1) return 0;
    if (p->N < 2 || (p->N & (p->N - 1))) return 0;                 /* reference src/trgsw.hpp:142-143 */
    if (p->bg_bit != 8 && p->bg_bit != 16) return 0;               /* reference src/types.hpp:11-19  */
    if (p->L * p->bg_bit > 31 || p->T * p->base_bit > 31) return 0; /* reference src/trgsw.hpp:28-29 */
    if (p->base_bit < 1) return 0;                                  /* reference src/tlwe_key_switching.hpp:52-53 */
    return 1;
}

size_t tfo_bk_words(const tfo_params* p) {
    return (size_t)p->n * (p->K + 1) * p->L * (p->K + 1) * p->N;
}
size_t tfo_ks_words(const tfo_params* p) {
    return (size_t)p->N * p->K * p->T * ((size_t)1 << p->base_bit) * (p->n + 1);
}

/* ------------------------------------------------------------------ primitives */

/* reference src/types.hpp:299-312: p(X) * X^n mod X^N + 1, n reduced into [0, 2N) */
void tfo_poly_mul_x_exp(uint32_t* out, const uint32_t* in, int N, int n) {
    n %= 2 * N;
    if (n < 0) n += 2 * N;
    for (int i = 0; i < N; i++) {
        int t = i + n;
        if (t >= 2 * N)      out[t - 2 * N] = in[i];
        else if (t >= N)     out[t - N] = 0u - in[i];
        else                 out[t] = in[i];
    }
}

/* reference src/trgsw.hpp:26-50: signed gadget digits, most significant level first */
void tfo_decompose(int32_t* digits, const uint32_t* poly, const tfo_params* p) {
    const int N = p->N, L = p->L, bg = p->bg_bit;
    const uint32_t round_offset = 1u << (32 - L * bg - 1);
    const uint32_t mask = (1u << bg) - 1;
    const int32_t half = 1 << (bg - 1), full = 1 << bg;
    for (int j = 0; j < N; j++) {
        const uint32_t t = poly[j] + round_offset;
        int32_t carry = 0;
        for (int l = L - 1; l >= 0; l--) {
            int32_t hat = (int32_t)((t >> (32 - bg * (l + 1))) & mask) + carry;
            if (hat >= half) { digits[(size_t)l * N + j] = hat - full; carry = 1; }
            else             { digits[(size_t)l * N + j] = hat;        carry = 0; }
        }
        /* the carry out of level 0 is dropped (reference src/trgsw.hpp:43 `if (i)`) */
    }
}

/* reference src/types.hpp:225-236: acc += d (*) b, negacyclic, wrap-around */
void tfo_negacyclic_mac_schoolbook(uint32_t* acc, const int32_t* d, const uint32_t* b, int N) {
    for (int i = 0; i < N; i++) {
        const uint32_t di = (uint32_t)d[i];
        if (di == 0) continue;
        for (int j = 0; j < N - i; j++) acc[i + j] += di * b[j];
        for (int j = N - i; j < N; j++) acc[i + j - N] -= di * b[j];
    }
}

/* ---- Goldilocks field, p = 2^64 - 2^32 + 1 (oracle-side fast multiplier) ---- */
#define GP 0xFFFFFFFF00000001ULL
typedef unsigned __int128 u128;

static inline uint64_t g_red(u128 x) {
    const uint64_t lo = (uint64_t)x, hi = (uint64_t)(x >> 64);
    const uint64_t hh = hi >> 32, hl = hi & 0xFFFFFFFFULL;
    uint64_t t0 = lo - hh;                       /* 2^96 = -1 (mod p) */
    if (lo < hh) t0 -= 0xFFFFFFFFULL;
    const uint64_t t1 = hl * 0xFFFFFFFFULL;      /* 2^64 = 2^32 - 1 (mod p) */
    uint64_t r = t0 + t1;
    if (r < t1) r += 0xFFFFFFFFULL;
    if (r >= GP) r -= GP;
    return r;
}
static inline uint64_t g_mul(uint64_t a, uint64_t b) { return g_red((u128)a * b); }
static inline uint64_t g_add(uint64_t a, uint64_t b) {
    uint64_t s = a + b;
    if (s < a) s += 0xFFFFFFFFULL;
    if (s >= GP) s -= GP;
    return s;
}
static inline uint64_t g_sub(uint64_t a, uint64_t b) { return a >= b ? a - b : a + (GP - b); }
static uint64_t g_pow(uint64_t a, uint64_t e) {
    uint64_t r = 1;
    while (e) { if (e & 1) r = g_mul(r, a); a = g_mul(a, a); e >>= 1; }
    return r;
}

typedef struct {
    int N;
    uint64_t* fwd_tw;  /* [N] psi^bitrev(i), merged negacyclic forward twiddles */
    uint64_t* inv_tw;  /* [N] psi^-bitrev(i) */
    uint64_t n_inv;
} g_plan;

#define MAX_PLANS 16
static g_plan g_plans[MAX_PLANS];
static int g_nplans = 0;

static const g_plan* g_get_plan(int N) {
    const g_plan* found = NULL;
#ifdef _OPENMP
#pragma omp critical(tfo_plan)
#endif
    {
        for (int i = 0; i < g_nplans; i++) if (g_plans[i].N == N) found = &g_plans[i];
        if (!found && g_nplans < MAX_PLANS) {
            g_plan* pl = &g_plans[g_nplans];
            const int lg = ilog2(N);
            const uint64_t psi = g_pow(7, (GP - 1) / (2 * (uint64_t)N)); /* 7 generates F_p^* */
            const uint64_t psi_inv = g_pow(psi, GP - 2);
            pl->N = N;
            pl->fwd_tw = (uint64_t*)malloc(sizeof(uint64_t) * N);
            pl->inv_tw = (uint64_t*)malloc(sizeof(uint64_t) * N);
            for (int i = 0; i < N; i++) {
                int r = 0;
                for (int b = 0; b < lg; b++) if (i & (1 << b)) r |= 1 << (lg - 1 - b);
                pl->fwd_tw[i] = g_pow(psi, (uint64_t)r);
                pl->inv_tw[i] = g_pow(psi_inv, (uint64_t)r);
            }
            pl->n_inv = g_pow((uint64_t)N, GP - 2);
            g_nplans++;
            found = pl;
        }
    }
    return found;
}

/* in-place forward negacyclic NTT, natural order in, bit-reversed order out */
static void g_ntt(uint64_t* a, const g_plan* pl) {
    const int N = pl->N;
    int t = N;
    for (int m = 1; m < N; m <<= 1) {
        t >>= 1;
        for (int i = 0; i < m; i++) {
            const uint64_t w = pl->fwd_tw[m + i];
            uint64_t* x = a + 2 * i * t;
            for (int j = 0; j < t; j++) {
                const uint64_t u = x[j], v = g_mul(x[j + t], w);
                x[j] = g_add(u, v);
                x[j + t] = g_sub(u, v);
            }
        }
    }
}

/* in-place inverse: bit-reversed in, natural out, scaled by 1/N */
static void g_intt(uint64_t* a, const g_plan* pl) {
    const int N = pl->N;
    int t = 1;
    for (int m = N >> 1; m >= 1; m >>= 1) {
        for (int i = 0; i < m; i++) {
            const uint64_t w = pl->inv_tw[m + i];
            uint64_t* x = a + 2 * i * t;
            for (int j = 0; j < t; j++) {
                const uint64_t u = x[j], v = x[j + t];
                x[j] = g_add(u, v);
                x[j + t] = g_mul(g_sub(u, v), w);
            }
        }
        t <<= 1;
    }
    for (int j = 0; j < N; j++) a[j] = g_mul(a[j], pl->n_inv);
}

static inline uint64_t g_from_i64(int64_t v) { return v >= 0 ? (uint64_t)v : GP - (uint64_t)(-v); }
/* centred lift of a field element back to Z, then reduce mod 2^32 */
static inline uint32_t g_to_torus(uint64_t v) {
    if (v > GP / 2) return (uint32_t)(0u - (uint32_t)(GP - v));
    return (uint32_t)v;
}

/* ---- replay of the reference's FFT product (reference src/types.hpp:237-292, fft :152-197) ----
 * Restated from its definition: fold the N reals into M = N/2 complex numbers (first half real
 * parts, second half imaginary parts), twist by successive powers of e^(i pi / N) built by
 * repeated multiplication, an M-point decimation-in-frequency style FFT whose stage i multiplies
 * the upper half of block j by omega^bitrev(2j) and ends with a bit-reversal permutation,
 * pointwise product, the same FFT with the omega table reversed from index 1, scale by 1/M, untwist,
 * and convert each part with the truncating double -> torus rule.  Every floating-point operation
 * is a plain IEEE double multiply, add or subtract in the reference's order (complex products as
 * (ac - bd, ad + bc)); the three std::polar values the tables start from are constants (kUnitRoot).
 * Build without -ffast-math and without FMA contraction (the oracle's Makefile does). */
typedef struct { double re, im; } cplx;
static inline cplx c_mul(cplx x, cplx y) {
    const double ac = x.re * y.re, bd = x.im * y.im, ad = x.re * y.im, bc = x.im * y.re;
    cplx r = { ac - bd, ad + bc };
    return r;
}
/* cos and sin of pi / 2^j for j = 3 .. 11, correctly rounded (computed in 80-digit arithmetic from the
 * double nearest to pi / 2^j; x86-64 glibc, whose std::polar the reference's recorded outputs come from,
 * returns the same values).  The power tables of the reference's FFT product (reference
 * src/types.hpp:257-268,276-283) start from these, so they do not depend on the math library of
 * the machine that runs this code. */
static const double kUnitRoot[9][2] = {
    {0x1.d906bcf328d46p-1, 0x1.87de2a6aea963p-2},  {0x1.f6297cff75cb0p-1, 0x1.8f8b83c69a60ap-3},
    {0x1.fd88da3d12526p-1, 0x1.917a6bc29b42cp-4},  {0x1.ff621e3796d7ep-1, 0x1.91f65f10dd814p-5},
    {0x1.ffd886084cd0dp-1, 0x1.92155f7a3667ep-6},  {0x1.fff62169b92dbp-1, 0x1.921d1fcdec784p-7},
    {0x1.fffd8858e8a92p-1, 0x1.921f0fe670071p-8},  {0x1.ffff621621d02p-1, 0x1.921f8becca4bap-9},
    {0x1.ffffd88586ee6p-1, 0x1.921faaee6472dp-10}};

/* std::polar(1.0, sign * pi / 2^j) */
static inline cplx c_polar_pow2(int j, int sign) {
    cplx r;
    if (j >= 3 && j <= 11) { r.re = kUnitRoot[j - 3][0]; r.im = sign * kUnitRoot[j - 3][1]; }
    else { const double theta = sign * M_PI / (double)(1u << j); r.re = 1.0 * cos(theta); r.im = 1.0 * sin(theta); }
    return r;
}
static inline int log2_exact(int v) { int b = 0; while ((1 << b) < v) b++; return b; }

static void ref_fft(cplx* a, cplx* tmp, int M, const cplx* omega_pow) {
    int bits = 0;
    while ((1 << bits) < M) bits++;
    cplx* cur = a;
    cplx* nxt = tmp;
    for (int i = 0; i < bits; i++) {
        const int span = 1 << (bits - i), half = span >> 1;
        for (int j = 0; j < (1 << i); j++) {
            int rev = 0;
            const int v = j << 1;
            for (int t = 0; t < bits; t++) if (v & (1 << t)) rev |= 1 << (bits - t - 1);
            for (int k = 0; k < half; k++) {
                const cplx even = cur[j * span + k];
                const cplx odd = c_mul(cur[j * span + k + half], omega_pow[rev]);
                nxt[j * span + k].re = even.re + odd.re;
                nxt[j * span + k].im = even.im + odd.im;
                nxt[j * span + k + half].re = even.re - odd.re;
                nxt[j * span + k + half].im = even.im - odd.im;
            }
        }
        cplx* sw = cur; cur = nxt; nxt = sw;
    }
    for (int i = 0; i < M; i++) {
        int rev = 0;
        for (int t = 0; t < bits; t++) if (i & (1 << t)) rev |= 1 << (bits - t - 1);
        nxt[rev] = cur[i];
    }
    if (nxt != a) memcpy(a, nxt, sizeof(cplx) * M);
}

static void reference_fft_mul(uint32_t* out, const int32_t* d, const uint32_t* b, int N) {
    const int M = N >> 1;
    cplx* buf = (cplx*)malloc(sizeof(cplx) * (size_t)M * 5);
    cplx *lhs = buf, *rhs = buf + M, *tmp = buf + 2 * M, *omega_pow = buf + 3 * M, *prod = buf + 4 * M;
    for (int i = 0; i < M; i++) {
        lhs[i].re = (double)d[i];
        lhs[i].im = (double)d[i + M];
        rhs[i].re = (double)b[i] / 4294967296.0;       /* operator double: raw / 2^32 */
        rhs[i].im = (double)b[i + M] / 4294967296.0;
    }
    const cplx w = c_polar_pow2(log2_exact(M) + 1, 1);         /* M_PI / (M * 2) */
    cplx w_pow = { 1.0, 0.0 };
    for (int i = 0; i < M; i++) {
        lhs[i] = c_mul(lhs[i], w_pow);
        rhs[i] = c_mul(rhs[i], w_pow);
        w_pow = c_mul(w_pow, w);
    }
    const cplx omega = c_polar_pow2(log2_exact(M) - 1, -1);    /* -M_PI * 2 / M */
    omega_pow[0].re = 1.0; omega_pow[0].im = 0.0;
    for (int i = 0; i < M - 1; i++) omega_pow[i + 1] = c_mul(omega_pow[i], omega);
    ref_fft(lhs, tmp, M, omega_pow);
    ref_fft(rhs, tmp, M, omega_pow);
    for (int i = 0; i < M; i++) prod[i] = c_mul(lhs[i], rhs[i]);
    for (int i = 1, j = M - 1; i < j; i++, j--) { cplx t = omega_pow[i]; omega_pow[i] = omega_pow[j]; omega_pow[j] = t; }
    ref_fft(prod, tmp, M, omega_pow);
    const cplx w_inv = c_polar_pow2(log2_exact(M) + 1, -1);    /* -M_PI / (M * 2) */
    cplx w_inv_pow = { 1.0, 0.0 };
    const double scale = 1. / M;
    for (int i = 0; i < M; i++) {
        prod[i].re *= scale;
        prod[i].im *= scale;
        prod[i] = c_mul(prod[i], w_inv_pow);
        w_inv_pow = c_mul(w_inv_pow, w_inv);
    }
    for (int i = 0; i < M; i++) {
        out[i] = tfo_torus_from_double(prod[i].re);
        out[i + M] = tfo_torus_from_double(prod[i].im);
    }
    free(buf);
}

void tfo_negacyclic_mul(uint32_t* out, const int32_t* d, const uint32_t* b, int N, int mode) {
    if (mode == TFO_MUL_REFERENCE_FFT && N >= 32) {  /* below 32 the reference multiplies exactly */
        reference_fft_mul(out, d, b, N);
        return;
    }
    if (mode == TFO_MUL_SCHOOLBOOK || mode == TFO_MUL_REFERENCE_FFT) {
        memset(out, 0, sizeof(uint32_t) * N);
        tfo_negacyclic_mac_schoolbook(out, d, b, N);
        return;
    }
    const g_plan* pl = g_get_plan(N);
    uint64_t* x = (uint64_t*)malloc(sizeof(uint64_t) * 2 * N);
    uint64_t* y = x + N;
    for (int i = 0; i < N; i++) { x[i] = g_from_i64(d[i]); y[i] = g_from_i64((int32_t)b[i]); }
    g_ntt(x, pl); g_ntt(y, pl);
    for (int i = 0; i < N; i++) x[i] = g_mul(x[i], y[i]);
    g_intt(x, pl);
    for (int i = 0; i < N; i++) out[i] = g_to_torus(x[i]);
    free(x);
}

/* reference src/trgsw.hpp:100-110 with src/types.hpp:433-445:
 * digit vector index r = comp*L + level; out[c] = sum_r dec[r] (*) C[r][c] */
void tfo_external_product(uint32_t* out, const uint32_t* trgsw, const uint32_t* trlwe,
                          const tfo_params* p, int mode) {
    const int N = p->N, K1 = p->K + 1, L = p->L, R = K1 * L;
    int32_t* dec = (int32_t*)malloc(sizeof(int32_t) * (size_t)R * N);
    for (int comp = 0; comp < K1; comp++)
        tfo_decompose(dec + (size_t)comp * L * N, trlwe + (size_t)comp * N, p);
    if (mode == TFO_MUL_REFERENCE_FFT && N >= 32) {
        /* every product rounded on its own, then wrap-around sums (reference src/types.hpp:433-445) */
        uint32_t* prod = (uint32_t*)malloc(sizeof(uint32_t) * N);
        memset(out, 0, sizeof(uint32_t) * (size_t)K1 * N);
        for (int r = 0; r < R; r++)
            for (int c = 0; c < K1; c++) {
                reference_fft_mul(prod, dec + (size_t)r * N, trgsw + ((size_t)r * K1 + c) * N, N);
                for (int i = 0; i < N; i++) out[(size_t)c * N + i] += prod[i];
            }
        free(prod);
    } else if (mode == TFO_MUL_SCHOOLBOOK || mode == TFO_MUL_REFERENCE_FFT) {
        memset(out, 0, sizeof(uint32_t) * (size_t)K1 * N);
        for (int r = 0; r < R; r++)
            for (int c = 0; c < K1; c++)
                tfo_negacyclic_mac_schoolbook(out + (size_t)c * N, dec + (size_t)r * N,
                                              trgsw + ((size_t)r * K1 + c) * N, N);
    } else {
        const g_plan* pl = g_get_plan(N);
        uint64_t* dx = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)(R + 2) * N);
        uint64_t* acc = dx + (size_t)R * N;
        uint64_t* y = acc + N;
        for (int r = 0; r < R; r++) {
            for (int i = 0; i < N; i++) dx[(size_t)r * N + i] = g_from_i64(dec[(size_t)r * N + i]);
            g_ntt(dx + (size_t)r * N, pl);
        }
        for (int c = 0; c < K1; c++) {
            memset(acc, 0, sizeof(uint64_t) * N);
            for (int r = 0; r < R; r++) {
                const uint32_t* b = trgsw + ((size_t)r * K1 + c) * N;
                for (int i = 0; i < N; i++) y[i] = g_from_i64((int32_t)b[i]);
                g_ntt(y, pl);
                for (int i = 0; i < N; i++) acc[i] = g_add(acc[i], g_mul(dx[(size_t)r * N + i], y[i]));
            }
            g_intt(acc, pl);
            for (int i = 0; i < N; i++) out[(size_t)c * N + i] = g_to_torus(acc[i]);
        }
        free(dx);
    }
    free(dec);
}

/* reference src/trgsw.hpp:112-117: external_product(C, d1 - d0) + d0 */
void tfo_cmux(uint32_t* out, const uint32_t* trgsw, const uint32_t* d0, const uint32_t* d1,
              const tfo_params* p, int mode) {
    const size_t W = (size_t)(p->K + 1) * p->N;
    uint32_t* diff = (uint32_t*)malloc(sizeof(uint32_t) * 2 * W);
    uint32_t* prod = diff + W;
    for (size_t i = 0; i < W; i++) diff[i] = d1[i] - d0[i];
    tfo_external_product(prod, trgsw, diff, p, mode);
    for (size_t i = 0; i < W; i++) out[i] = prod[i] + d0[i];
    free(diff);
}

/* rotation amounts, reference src/trgsw.hpp:155-162 */
static inline int rot_floor(uint32_t v, int N) { return (int)(v >> (32 - (ilog2(N) + 1))); }
static inline int rot_round(uint32_t v, int N) {
    const int sh = 32 - (ilog2(N) + 1);
    return (int)((v + (1u << (sh - 1))) >> sh);
}

/* reference src/trgsw.hpp:131-170 */
void tfo_blind_rotate(uint32_t* acc, const uint32_t* c, const uint32_t* bk, const uint32_t* test_vector,
                      const tfo_params* p, int mode, tfo_trace_fn trace, void* user) {
    const int N = p->N, K1 = p->K + 1;
    const size_t W = (size_t)K1 * N, row = (size_t)K1 * p->L * K1 * N;
    uint32_t* rotated = (uint32_t*)malloc(sizeof(uint32_t) * 2 * W);
    uint32_t* next = rotated + W;
    const int b2n = rot_floor(c[0], N);
    for (int q = 0; q < K1; q++) tfo_poly_mul_x_exp(acc + (size_t)q * N, test_vector + (size_t)q * N, N, -b2n);
    if (trace) trace(user, -1, acc);
    for (int i = 0; i < p->n; i++) {
        const int a2n = rot_round(c[1 + i], N);
        for (int q = 0; q < K1; q++) tfo_poly_mul_x_exp(rotated + (size_t)q * N, acc + (size_t)q * N, N, a2n);
        tfo_cmux(next, bk + (size_t)i * row, acc, rotated, p, mode);
        memcpy(acc, next, sizeof(uint32_t) * W);
        if (trace) trace(user, i, acc);
    }
    free(rotated);
}

/* reference src/trlwe.hpp:109-122 */
void tfo_sample_extract_index(uint32_t* lv1, const uint32_t* trlwe, int x, const tfo_params* p) {
    const int N = p->N;
    lv1[0] = trlwe[x];
    for (int j = 0; j < p->K; j++) {
        const uint32_t* a = trlwe + (size_t)(1 + j) * N;
        for (int i = 0; i <= x; i++) lv1[1 + j * N + i] = a[x - i];
        for (int i = x + 1; i < N; i++) lv1[1 + j * N + i] = 0u - a[N + x - i];
    }
}

/* reference src/trgsw.hpp:172-182: test vector b-poly = +1/8 everywhere, mask polys 0 */
void tfo_gate_bootstrapping_tlwe_to_tlwe(uint32_t* lv1, const uint32_t* c, const uint32_t* bk,
                                         const tfo_params* p, int mode) {
    const int N = p->N, K1 = p->K + 1;
    uint32_t* tv = (uint32_t*)calloc((size_t)2 * K1 * N, sizeof(uint32_t));
    uint32_t* acc = tv + (size_t)K1 * N;
    for (int i = 0; i < N; i++) tv[i] = 1u << 29;
    tfo_blind_rotate(acc, c, bk, tv, p, mode, NULL, NULL);
    tfo_sample_extract_index(lv1, acc, 0, p);
    free(tv);
}

/* reference src/tlwe_key_switching.hpp:47-65 */
void tfo_identity_key_switch(uint32_t* lv0, const uint32_t* lv1, const uint32_t* ks, const tfo_params* p) {
    const int n1 = p->n + 1, T = p->T, bb = p->base_bit, nlv1 = p->N * p->K;
    const uint32_t round_offset = 1u << (32 - T * bb - 1);
    const uint32_t mask = (1u << bb) - 1;
    memset(lv0, 0, sizeof(uint32_t) * n1);
    lv0[0] = lv1[0];
    for (int i = 0; i < nlv1; i++) {
        const uint32_t t = lv1[1 + i] + round_offset;
        for (int m = 0; m < T; m++) {
            const uint32_t o = (t >> (32 - (m + 1) * bb)) & mask;
            if (!o) continue;
            const uint32_t* row = ks + (((size_t)i * T + m) * ((size_t)1 << bb) + o) * n1;
            for (int w = 0; w < n1; w++) lv0[w] -= row[w];
        }
    }
}

/* reference src/hom_nand.hpp:76-78: c = (1/8, 0, ..) - (a + b) */
void tfo_nand_combine(uint32_t* c, const uint32_t* a, const uint32_t* b, const tfo_params* p) {
    c[0] = (1u << 29) - (a[0] + b[0]);
    for (int j = 1; j <= p->n; j++) c[j] = 0u - (a[j] + b[j]);
}

/* Generalisation of the line above to the other two-input gates (the reference has none). */
int tfo_gate_coefficients(int op, uint32_t* offset, int32_t* sa, int32_t* sb) {
    const uint32_t mu = 1u << 29, q = 1u << 30;
    switch (op) {
    case TFO_NAND:  *offset = mu;      *sa = -1; *sb = -1; return 0;
    case TFO_AND:   *offset = 0u - mu; *sa = 1;  *sb = 1;  return 0;
    case TFO_OR:    *offset = mu;      *sa = 1;  *sb = 1;  return 0;
    case TFO_NOR:   *offset = 0u - mu; *sa = -1; *sb = -1; return 0;
    case TFO_XOR:   *offset = q;       *sa = 2;  *sb = 2;  return 0;
    case TFO_XNOR:  *offset = 0u - q;  *sa = -2; *sb = -2; return 0;
    case TFO_ANDNY: *offset = 0u - mu; *sa = -1; *sb = 1;  return 0;
    case TFO_ANDYN: *offset = 0u - mu; *sa = 1;  *sb = -1; return 0;
    case TFO_ORNY:  *offset = mu;      *sa = -1; *sb = 1;  return 0;
    case TFO_ORYN:  *offset = mu;      *sa = 1;  *sb = -1; return 0;
    default: return -1;
    }
}

int tfo_gate_combine(uint32_t* c, int op, const uint32_t* a, const uint32_t* b, const tfo_params* p) {
    uint32_t off;
    int32_t sa, sb;
    if (tfo_gate_coefficients(op, &off, &sa, &sb)) return -1;
    for (int j = 0; j <= p->n; j++) c[j] = (uint32_t)sa * a[j] + (uint32_t)sb * b[j];
    c[0] += off;
    return 0;
}

void tfo_not(uint32_t* out, const uint32_t* a, const tfo_params* p) {
    for (int j = 0; j <= p->n; j++) out[j] = 0u - a[j];
}

/* reference src/hom_nand.hpp:74-81 */
void tfo_nand(uint32_t* out, const uint32_t* a, const uint32_t* b, const uint32_t* bk, const uint32_t* ks,
              const tfo_params* p, int mode) {
    uint32_t* c = (uint32_t*)malloc(sizeof(uint32_t) * ((size_t)p->n + 1 + (size_t)p->N * p->K + 1));
    uint32_t* lv1 = c + p->n + 1;
    tfo_nand_combine(c, a, b, p);
    tfo_gate_bootstrapping_tlwe_to_tlwe(lv1, c, bk, p, mode);
    tfo_identity_key_switch(out, lv1, ks, p);
    free(c);
}

/* ------------------------------------------------------------------ fast context */

struct tfo_ctx {
    tfo_params p;
    const uint32_t* ks;  /* borrowed */
    uint64_t* bk_ntt;    /* [n][(K+1)L][K+1][N], bit-reversed transform order */
    const g_plan* plan;
};

tfo_ctx* tfo_ctx_create(const tfo_params* p, const uint32_t* bk, const uint32_t* ks) {
    if (!tfo_params_valid(p)) return NULL;
    tfo_ctx* ctx = (tfo_ctx*)calloc(1, sizeof(tfo_ctx));
    ctx->p = *p;
    ctx->ks = ks;
    ctx->plan = g_get_plan(p->N);
    const size_t words = tfo_bk_words(p);
    const size_t polys = words / p->N;
    ctx->bk_ntt = (uint64_t*)malloc(sizeof(uint64_t) * words);
    const int N = p->N;
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
    for (long q = 0; q < (long)polys; q++) {
        uint64_t* y = ctx->bk_ntt + (size_t)q * N;
        for (int i = 0; i < N; i++) y[i] = g_from_i64((int32_t)bk[(size_t)q * N + i]);
        g_ntt(y, ctx->plan);
    }
    return ctx;
}

void tfo_ctx_destroy(tfo_ctx* ctx) {
    if (!ctx) return;
    free(ctx->bk_ntt);
    free(ctx);
}

/* same computation as tfo_blind_rotate with the test vector of
 * tfo_gate_bootstrapping_tlwe_to_tlwe, against the pre-transformed key */
void tfo_ctx_blind_rotate(const tfo_ctx* ctx, uint32_t* acc, const uint32_t* c) {
    const tfo_params* p = &ctx->p;
    const int N = p->N, K1 = p->K + 1, L = p->L, R = K1 * L;
    const size_t W = (size_t)K1 * N;
    uint32_t* diff = (uint32_t*)malloc(sizeof(uint32_t) * 2 * W + sizeof(int32_t) * (size_t)L * N);
    uint32_t* tmp = diff + W;
    int32_t* dec = (int32_t*)(tmp + W);
    uint64_t* dx = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)(R + 1) * N);
    uint64_t* sum = dx + (size_t)R * N;

    memset(tmp, 0, sizeof(uint32_t) * W);
    for (int i = 0; i < N; i++) tmp[i] = 1u << 29;
    const int b2n = rot_floor(c[0], N);
    for (int q = 0; q < K1; q++) tfo_poly_mul_x_exp(acc + (size_t)q * N, tmp + (size_t)q * N, N, -b2n);

    for (int i = 0; i < p->n; i++) {
        const int a2n = rot_round(c[1 + i], N);
        for (int q = 0; q < K1; q++) {
            tfo_poly_mul_x_exp(tmp + (size_t)q * N, acc + (size_t)q * N, N, a2n);
            for (int j = 0; j < N; j++) diff[(size_t)q * N + j] = tmp[(size_t)q * N + j] - acc[(size_t)q * N + j];
            tfo_decompose(dec, diff + (size_t)q * N, p);
            for (int l = 0; l < L; l++) {
                uint64_t* x = dx + ((size_t)q * L + l) * N;
                for (int j = 0; j < N; j++) x[j] = g_from_i64(dec[(size_t)l * N + j]);
                g_ntt(x, ctx->plan);
            }
        }
        const uint64_t* row = ctx->bk_ntt + (size_t)i * R * K1 * N;
        for (int col = 0; col < K1; col++) {
            memset(sum, 0, sizeof(uint64_t) * N);
            for (int r = 0; r < R; r++) {
                const uint64_t* y = row + ((size_t)r * K1 + col) * N;
                const uint64_t* x = dx + (size_t)r * N;
                for (int j = 0; j < N; j++) sum[j] = g_add(sum[j], g_mul(x[j], y[j]));
            }
            g_intt(sum, ctx->plan);
            for (int j = 0; j < N; j++) acc[(size_t)col * N + j] += g_to_torus(sum[j]);
        }
    }
    free(dx);
    free(diff);
}

void tfo_ctx_bootstrap(const tfo_ctx* ctx, uint32_t* lv0_out, const uint32_t* c) {
    const tfo_params* p = &ctx->p;
    const size_t W = (size_t)(p->K + 1) * p->N;
    uint32_t* acc = (uint32_t*)malloc(sizeof(uint32_t) * (W + (size_t)p->N * p->K + 1));
    uint32_t* lv1 = acc + W;
    tfo_ctx_blind_rotate(ctx, acc, c);
    tfo_sample_extract_index(lv1, acc, 0, p);
    tfo_identity_key_switch(lv0_out, lv1, ctx->ks, p);
    free(acc);
}

int tfo_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

void tfo_ctx_bootstrap_batch(const tfo_ctx* ctx, uint32_t* out, const uint32_t* c, size_t n_gates, int threads) {
    const size_t n1 = (size_t)ctx->p.n + 1;
    if (threads <= 0) threads = tfo_max_threads();
    (void)threads;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads)
#endif
    for (long g = 0; g < (long)n_gates; g++)
        tfo_ctx_bootstrap(ctx, out + (size_t)g * n1, c + (size_t)g * n1);
}

void tfo_ctx_nand_batch(const tfo_ctx* ctx, uint32_t* out, const uint32_t* a, const uint32_t* b,
                        size_t n_gates, int threads) {
    const size_t n1 = (size_t)ctx->p.n + 1;
    if (threads <= 0) threads = tfo_max_threads();
    (void)threads;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads)
#endif
    for (long g = 0; g < (long)n_gates; g++) {
        uint32_t* c = (uint32_t*)malloc(sizeof(uint32_t) * n1);
        tfo_nand_combine(c, a + (size_t)g * n1, b + (size_t)g * n1, &ctx->p);
        tfo_ctx_bootstrap(ctx, out + (size_t)g * n1, c);
        free(c);
    }
}

int tfo_ctx_gate_batch(const tfo_ctx* ctx, int op, const uint8_t* ops, uint32_t* out, const uint32_t* a,
                       const uint32_t* b, size_t n_gates, int threads) {
    const size_t n1 = (size_t)ctx->p.n + 1;
    uint32_t off;
    int32_t sa, sb;
    for (size_t g = 0; g < n_gates; g++)
        if (tfo_gate_coefficients(ops ? ops[g] : op, &off, &sa, &sb)) return -1;
    if (threads <= 0) threads = tfo_max_threads();
    (void)threads;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads)
#endif
    for (long g = 0; g < (long)n_gates; g++) {
        uint32_t* c = (uint32_t*)malloc(sizeof(uint32_t) * n1);
        tfo_gate_combine(c, ops ? ops[g] : op, a + (size_t)g * n1, b + (size_t)g * n1, &ctx->p);
        tfo_ctx_bootstrap(ctx, out + (size_t)g * n1, c);
        free(c);
    }
    return 0;
}

void tfo_ctx_mux_batch(const tfo_ctx* ctx, uint32_t* out, const uint32_t* sel, const uint32_t* x1,
                       const uint32_t* x0, size_t n_gates, int threads) {
    const tfo_params* p = &ctx->p;
    const size_t n1 = (size_t)p->n + 1, W = (size_t)(p->K + 1) * p->N, L1 = (size_t)p->N * p->K + 1;
    if (threads <= 0) threads = tfo_max_threads();
    (void)threads;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads)
#endif
    for (long g = 0; g < (long)n_gates; g++) {
        uint32_t* c = (uint32_t*)malloc(sizeof(uint32_t) * (n1 + W + 2 * L1));
        uint32_t* acc = c + n1;
        uint32_t* u1 = acc + W;
        uint32_t* u0 = u1 + L1;
        tfo_gate_combine(c, TFO_AND, sel + (size_t)g * n1, x1 + (size_t)g * n1, p);
        tfo_ctx_blind_rotate(ctx, acc, c);
        tfo_sample_extract_index(u1, acc, 0, p);
        tfo_gate_combine(c, TFO_ANDNY, sel + (size_t)g * n1, x0 + (size_t)g * n1, p);
        tfo_ctx_blind_rotate(ctx, acc, c);
        tfo_sample_extract_index(u0, acc, 0, p);
        for (size_t j = 0; j < L1; j++) u1[j] += u0[j];
        u1[0] += 1u << 29;
        tfo_identity_key_switch(out + (size_t)g * n1, u1, ctx->ks, p);
        free(c);
    }
}

/* ------------------------------------------------------------------ seeded client side */
/* The reference draws from std::random_device (src/random.hpp:10) and cannot be seeded;
 * "same keys" therefore means same key MATERIAL.  This generator is xoshiro256** seeded
 * through splitmix64; the distributions follow the reference: uniform 32-bit masks
 * (src/random.hpp:39-41), binary secrets (:16-25), Gaussian noise scaled by alpha and
 * converted with the truncating double->torus rule (src/tlwe.hpp:30, src/types.hpp:28-33). */

static inline uint64_t rotl64(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }

void tfo_rng_seed(tfo_rng* r, uint64_t seed) {
    uint64_t z = seed;
    for (int i = 0; i < 4; i++) {
        z += 0x9E3779B97F4A7C15ULL;
        uint64_t x = z;
        x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
        x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL;
        r->s[i] = x ^ (x >> 31);
    }
    r->has_spare = 0;
    r->spare = 0.0;
}

uint64_t tfo_rng_u64(tfo_rng* r) {
    uint64_t* s = r->s;
    const uint64_t result = rotl64(s[1] * 5, 7) * 9;
    const uint64_t t = s[1] << 17;
    s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3];
    s[2] ^= t;
    s[3] = rotl64(s[3], 45);
    return result;
}

static inline double rng_unit(tfo_rng* r) { return (double)(tfo_rng_u64(r) >> 11) * (1.0 / 9007199254740992.0); }

/* Approximately standard normal, from additions only: the sum of twelve uniforms minus six
 * (Irwin-Hall; mean 0, variance 1, support [-6, 6]).  No logarithm or trigonometric function, so the
 * seeded key generator returns the same bytes on every machine, whatever its math library -- the
 * recorded known answers of the compiled reference (tests/golden/refnand512.npz) depend on that. */
double tfo_rng_normal(tfo_rng* r) {
    double s = 0.0;
    for (int i = 0; i < 12; i++) s += rng_unit(r);
    return s - 6.0;
}

/* reference src/types.hpp:28-33 */
uint32_t tfo_torus_from_double(double d) {
    d = fmod(d, 1.0);
    if (d < 0.0) d += 1.0;
    d *= 4294967296.0;
    return (uint32_t)(uint64_t)d;
}

void tfo_keygen_secret(uint8_t* s_lv0, uint8_t* s_ring, const tfo_params* p, tfo_rng* r) {
    for (int i = 0; i < p->n; i++) s_lv0[i] = (uint8_t)(tfo_rng_u64(r) & 1);
    for (int i = 0; i < p->K * p->N; i++) s_ring[i] = (uint8_t)(tfo_rng_u64(r) & 1);
}

/* reference src/tlwe.hpp:55-61: b = a.s + m + e, output [b, a...] */
void tfo_tlwe_encrypt(uint32_t* ct, uint32_t m, const uint8_t* s, int n, double alpha, tfo_rng* r) {
    uint32_t b = m + tfo_torus_from_double(tfo_rng_normal(r) * alpha);
    for (int i = 0; i < n; i++) {
        const uint32_t a = (uint32_t)tfo_rng_u64(r);
        ct[1 + i] = a;
        if (s[i]) b += a;
    }
    ct[0] = b;
}

/* reference src/tlwe.hpp:67-71: b - a.s */
uint32_t tfo_tlwe_phase(const uint32_t* ct, const uint8_t* s, int n) {
    uint32_t ph = ct[0];
    for (int i = 0; i < n; i++) if (s[i]) ph -= ct[1 + i];
    return ph;
}

/* reference src/types.hpp:25,34 and src/tlwe.hpp:62-65 */
void tfo_encrypt_bit(uint32_t* ct, int bit, const uint8_t* s_lv0, const tfo_params* p, tfo_rng* r) {
    const uint32_t mu = 1u << 29;
    tfo_tlwe_encrypt(ct, bit ? mu : 0u - mu, s_lv0, p->n, p->alpha_lv0, r);
}

/* reference src/types.hpp:46 and src/tlwe.hpp:72-75: true iff the top bit of the phase is clear */
int tfo_decrypt_bit(const uint32_t* ct, const uint8_t* s_lv0, const tfo_params* p) {
    return !((tfo_tlwe_phase(ct, s_lv0, p->n) >> 31) & 1u);
}

static void binary_poly_mac(uint32_t* acc, const uint32_t* a, const uint8_t* s, int N, int sign) {
    /* acc +/-= a (*) s with s binary, negacyclic */
    for (int i = 0; i < N; i++) {
        if (!s[i]) continue;
        for (int j = 0; j < N; j++) {
            const int t = i + j;
            uint32_t v = a[j];
            if (t >= N) v = 0u - v;
            if (sign < 0) v = 0u - v;
            acc[t >= N ? t - N : t] += v;
        }
    }
}

/* reference src/trlwe.hpp:75-82 */
void tfo_trlwe_encrypt(uint32_t* ct, const uint32_t* m, const uint8_t* s_ring, const tfo_params* p, tfo_rng* r) {
    const int N = p->N;
    for (int k = 0; k < p->K; k++)
        for (int j = 0; j < N; j++) ct[(size_t)(1 + k) * N + j] = (uint32_t)tfo_rng_u64(r);
    for (int j = 0; j < N; j++) ct[j] = m[j] + tfo_torus_from_double(tfo_rng_normal(r) * p->alpha_lv1);
    for (int k = 0; k < p->K; k++) binary_poly_mac(ct, ct + (size_t)(1 + k) * N, s_ring + (size_t)k * N, N, +1);
}

/* reference src/trlwe.hpp:91-97 */
void tfo_trlwe_phase(uint32_t* m, const uint32_t* ct, const uint8_t* s_ring, const tfo_params* p) {
    const int N = p->N;
    memcpy(m, ct, sizeof(uint32_t) * N);
    for (int k = 0; k < p->K; k++) binary_poly_mac(m, ct + (size_t)(1 + k) * N, s_ring + (size_t)k * N, N, -1);
}

/* reference src/trgsw.hpp:61-82: gadget on the block diagonal plus a fresh zero encryption per row */
void tfo_trgsw_encrypt_int_poly(uint32_t* trgsw, const int32_t* mu, const uint8_t* s_ring,
                                const tfo_params* p, tfo_rng* r) {
    const int N = p->N, K1 = p->K + 1, L = p->L;
    uint32_t* zero = (uint32_t*)calloc(N, sizeof(uint32_t));
    for (int comp = 0; comp < K1; comp++)
        for (int l = 0; l < L; l++) {
            uint32_t* row = trgsw + ((size_t)(comp * L + l) * K1) * N;
            tfo_trlwe_encrypt(row, zero, s_ring, p, r);
            uint32_t* diag = row + (size_t)comp * N;
            for (int j = 0; j < N; j++) diag[j] += (uint32_t)mu[j] << (32 - p->bg_bit * (l + 1));
        }
    free(zero);
}

/* reference src/trgsw.hpp:93-98,119-129 */
void tfo_make_bk(uint32_t* bk, const uint8_t* s_lv0, const uint8_t* s_ring, const tfo_params* p, tfo_rng* r) {
    const size_t row = (size_t)(p->K + 1) * p->L * (p->K + 1) * p->N;
    int32_t* mu = (int32_t*)calloc(p->N, sizeof(int32_t));
    for (int i = 0; i < p->n; i++) {
        mu[0] = s_lv0[i] ? 1 : 0;
        tfo_trgsw_encrypt_int_poly(bk + (size_t)i * row, mu, s_ring, p, r);
    }
    free(mu);
}

/* reference src/tlwe_key_switching.hpp:26-45; the lv1 key is the ring key's coefficients
 * concatenated (reference src/trlwe.hpp:124-131) */
void tfo_make_ks(uint32_t* ks, const uint8_t* s_lv0, const uint8_t* s_ring, const tfo_params* p, tfo_rng* r) {
    const int n1 = p->n + 1, T = p->T, bb = p->base_bit, nlv1 = p->N * p->K;
    memset(ks, 0, sizeof(uint32_t) * tfo_ks_words(p));
    for (int i = 0; i < nlv1; i++)
        for (int m = 0; m < T; m++)
            for (int o = 1; o < (1 << bb); o++) {
                const uint32_t msg = s_ring[i] ? (uint32_t)o << (32 - (m + 1) * bb) : 0u;
                uint32_t* row = ks + (((size_t)i * T + m) * ((size_t)1 << bb) + o) * n1;
                tfo_tlwe_encrypt(row, msg, s_lv0, p->n, p->alpha_lv0, r);
            }
}

/* ---- restatement of the GPU key generator (tfhe_amd/csrc/keygen_kernels.hip) -------------------------
 * Same key structure as tfo_make_bk / tfo_make_ks (reference src/trgsw.hpp:61-82,119-129,
 * src/tlwe_key_switching.hpp:26-45), with every mask and noise word taken from the ChaCha20 block
 * function (RFC 8439: 20 rounds, feed-forward) under a 256-bit seed, block position (index, stream,
 * row low, row high) in the counter and nonce words.  Masks are pure generator words, so they are
 * identical on every machine; the Gaussian noise goes through log, sqrt, sin and cos, whose last-bit
 * rounding may differ between this libm and the GPU's, so a noise word may differ by one unit where the
 * scaled normal lies within rounding of an integer. */
static uint32_t tfo_rotl32(uint32_t v, int c) { return (v << c) | (v >> (32 - c)); }
#define TFO_QR(a, b, c, d)                   \
    a += b; d ^= a; d = tfo_rotl32(d, 16);   \
    c += d; b ^= c; b = tfo_rotl32(b, 12);   \
    a += b; d ^= a; d = tfo_rotl32(d, 8);    \
    c += d; b ^= c; b = tfo_rotl32(b, 7);

void tfo_chacha20_block(uint32_t out[16], const uint32_t seed[8], uint32_t stream, uint64_t row, uint32_t index) {
    uint32_t in[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u};
    uint32_t x[16];
    for (int i = 0; i < 8; i++) in[4 + i] = seed[i];
    in[12] = index; in[13] = stream; in[14] = (uint32_t)row; in[15] = (uint32_t)(row >> 32);
    memcpy(x, in, sizeof(x));
    for (int r = 0; r < 10; r++) {
        TFO_QR(x[0], x[4], x[8], x[12])
        TFO_QR(x[1], x[5], x[9], x[13])
        TFO_QR(x[2], x[6], x[10], x[14])
        TFO_QR(x[3], x[7], x[11], x[15])
        TFO_QR(x[0], x[5], x[10], x[15])
        TFO_QR(x[1], x[6], x[11], x[12])
        TFO_QR(x[2], x[7], x[8], x[13])
        TFO_QR(x[3], x[4], x[9], x[14])
    }
    for (int i = 0; i < 16; i++) out[i] = x[i] + in[i];
}

/* two standard normals from four words: Box-Muller on a 53-bit uniform in (0, 1] and a 52-bit one in [0, 1) */
static void tfo_normal2(const uint32_t* w, double* z0, double* z1) {
    const double u1 = ((double)(((uint64_t)w[0] << 21) ^ (w[1] >> 11)) + 1.0) * (1.0 / 9007199254740992.0);
    const double u2 = (double)(((uint64_t)w[2] << 20) ^ (w[3] >> 12)) * (1.0 / 4503599627370496.0);
    const double r = sqrt(-2.0 * log(u1));
    const double a = 2.0 * u2;          /* angle in units of pi, [0, 2) */
    /* sin(pi a), cos(pi a) with the argument reduced exactly to [-1/4, 1/4] around a multiple of 1/2 */
    const double k = floor(a * 2.0 + 0.5);          /* nearest multiple of 1/2 */
    const double f = (a - k * 0.5) * 3.14159265358979323846;
    const double sf = sin(f), cf = cos(f);
    double sv, cv;
    switch (((int)k) & 3) {
    case 0: sv = sf; cv = cf; break;
    case 1: sv = cf; cv = -sf; break;
    case 2: sv = -sf; cv = -cf; break;
    default: sv = -cf; cv = sf; break;
    }
    *z0 = r * cv;
    *z1 = r * sv;
}

enum { TFO_STREAM_BK_MASK = 1, TFO_STREAM_BK_NOISE = 2, TFO_STREAM_KS_MASK = 3, TFO_STREAM_KS_NOISE = 4 };

void tfo_keygen_chacha_bk(uint32_t* bk, const uint8_t* s_lv0, const uint8_t* s_ring, const tfo_params* p,
                          double alpha, const uint32_t seed[8]) {
    const int N = p->N, K = p->K, K1 = K + 1, L = p->L, R = K1 * L;
    uint32_t w[16];
    for (int64_t row = 0; row < (int64_t)p->n * R; row++) {
        const int i = (int)(row / R), r = (int)(row % R), comp = r / L, level = r % L;
        uint32_t* out = bk + (size_t)row * K1 * N;
        for (int j = 0; j < N; j += 8) {
            tfo_chacha20_block(w, seed, TFO_STREAM_BK_NOISE, (uint64_t)row, (uint32_t)(j >> 3));
            for (int e = 0; e < 4; e++) {
                double z0, z1;
                tfo_normal2(w + 4 * e, &z0, &z1);
                if (j + 2 * e < N) out[j + 2 * e] = tfo_torus_from_double(z0 * alpha);
                if (j + 2 * e + 1 < N) out[j + 2 * e + 1] = tfo_torus_from_double(z1 * alpha);
            }
        }
        for (int k = 0; k < K; k++) {
            uint32_t* a = out + (size_t)(1 + k) * N;
            for (int j = 0; j < N; j += 16) {
                tfo_chacha20_block(w, seed, TFO_STREAM_BK_MASK, (uint64_t)row * K + k, (uint32_t)(j >> 4));
                for (int e = 0; e < 16 && j + e < N; e++) a[j + e] = w[e];
            }
            /* b += a (*) s_k, negacyclic, s binary */
            const uint8_t* sk = s_ring + (size_t)k * N;
            for (int m = 0; m < N; m++) {
                if (!sk[m]) continue;
                for (int j = 0; j < N; j++) {
                    const int idx = j - m;
                    out[j] += idx < 0 ? 0u - a[idx + N] : a[idx];
                }
            }
        }
        if (s_lv0[i]) out[(size_t)comp * N] += 1u << (32 - p->bg_bit * (level + 1));
    }
}

void tfo_keygen_chacha_ks(uint32_t* ks, const uint8_t* s_lv0, const uint8_t* s_lv1, const tfo_params* p,
                          double alpha, const uint32_t seed[8]) {
    const int n = p->n, T = p->T, bb = p->base_bit;
    const size_t rows = ((size_t)p->N * p->K * T) << bb;
    uint32_t w[16];
    for (size_t row = 0; row < rows; row++) {
        const int o = (int)(row & ((1u << bb) - 1));
        uint32_t* out = ks + row * (size_t)(n + 1);
        if (o == 0) {
            memset(out, 0, sizeof(uint32_t) * (size_t)(n + 1));
            continue;
        }
        const size_t im = row >> bb;
        const int m = (int)(im % T);
        const size_t i = im / T;
        uint32_t acc = 0;
        for (int j = 0; j < n; j += 16) {
            tfo_chacha20_block(w, seed, TFO_STREAM_KS_MASK, (uint64_t)row, (uint32_t)(j >> 4));
            for (int e = 0; e < 16 && j + e < n; e++) {
                out[1 + j + e] = w[e];
                if (s_lv0[j + e]) acc += w[e];
            }
        }
        tfo_chacha20_block(w, seed, TFO_STREAM_KS_NOISE, (uint64_t)row, 0u);
        double z0, z1;
        tfo_normal2(w, &z0, &z1);
        const uint32_t msg = s_lv1[i] ? (uint32_t)o << (32 - (m + 1) * bb) : 0u;
        out[0] = acc + tfo_torus_from_double(z0 * alpha) + msg;
    }
}
