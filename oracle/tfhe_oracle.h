/*
 * TEST INFRASTRUCTURE ONLY -- exact-integer CPU restatement of the gate-bootstrapping
 * path of KowerKoint/TFHE (HomNAND::nand and everything below it).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library, and only as the checker.  The product path (tfhe_amd/) never links,
 * imports or calls anything in this directory.
 *
 * Parity pinning: the reference has no golden vectors of its own.  This restatement is
 * pinned against outputs of the reference itself, compiled in the development
 * container from /root/reference/src by oracle/Makefile (target `ref`), at toy ring
 * degrees (N < 32) where the reference multiplies polynomials exactly
 * (reference src/types.hpp:225-236), and per integer-only primitive at the default
 * sizes.  The resulting vectors are committed under tests/golden/.
 *
 * All arithmetic is on the 32-bit discretised torus: uint32_t, wrap-around mod 2^32.
 *
 * Flat layouts (row-major, little-endian uint32_t):
 *   lv0 ciphertext   [n+1]                    = [b, a_0 .. a_{n-1}]      (reference src/tlwe.hpp:33-39)
 *   lv1 ciphertext   [N*K+1]                  = [b, a'_0 ..]             (reference src/trlwe.hpp:109-122)
 *   TRLWE sample     [K+1][N]                 polynomial 0 is b(X)        (reference src/trlwe.hpp:45-52)
 *   bootstrapping key[n][(K+1)*L][K+1][N]     row = comp*L + level        (reference src/trgsw.hpp:61-82,119-129)
 *   key-switch key   [N*K][T][1<<base_bit][n+1]  slot o = 0 is all zero   (reference src/tlwe_key_switching.hpp:26-45)
 */
#ifndef TFHE_ORACLE_H
#define TFHE_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct tfo_params {
    int32_t n;        /* lv0 TLWE dimension            (reference src/tlwe.hpp:12)   */
    int32_t N;        /* ring degree, power of two     (reference src/trlwe.hpp:10)  */
    int32_t K;        /* TRLWE mask polynomials        (reference src/trlwe.hpp:11)  */
    int32_t bg_bit;   /* gadget base bits, 8 or 16     (reference src/trgsw.hpp:8)   */
    int32_t L;        /* gadget levels                 (reference src/trgsw.hpp:9)   */
    int32_t T;        /* key-switch levels             (reference src/tlwe_key_switching.hpp:7) */
    int32_t base_bit; /* key-switch base bits          (reference src/tlwe_key_switching.hpp:8) */
    double alpha_lv0; /* lv0 noise stddev (torus units)(reference src/tlwe.hpp:13)   */
    double alpha_lv1; /* ring noise stddev             (reference src/trlwe.hpp:12)  */
} tfo_params;

/* Multipliers.  SCHOOLBOOK and NTT are exact and interchangeable.  REFERENCE_FFT replays the
 * reference's own double-precision FFT product (src/types.hpp:152-197,237-292) operation by
 * operation, rounding included, so that the whole pipeline can be compared bit for bit with the
 * compiled reference at ring degrees where that product is inexact (N >= 32). */
enum { TFO_MUL_SCHOOLBOOK = 0, TFO_MUL_NTT = 1, TFO_MUL_REFERENCE_FFT = 2 };

/* parameter sets */
void tfo_params_default(tfo_params* p); /* n=636 N=512 K=2 bg=8 L=2 T=5 base=2 */
void tfo_params_n2048(tfo_params* p);   /* n=800 N=2048 K=1 bg=8 L=3 T=8 base=2 */
int tfo_params_valid(const tfo_params* p);
size_t tfo_bk_words(const tfo_params* p);
size_t tfo_ks_words(const tfo_params* p);

/* ---- integer primitives (each cites the reference lines it restates) ---- */
void tfo_poly_mul_x_exp(uint32_t* out, const uint32_t* in, int N, int n);
void tfo_decompose(int32_t* digits /*[L][N]*/, const uint32_t* poly, const tfo_params* p);
void tfo_negacyclic_mac_schoolbook(uint32_t* acc, const int32_t* d, const uint32_t* b, int N);
void tfo_negacyclic_mul(uint32_t* out, const int32_t* d, const uint32_t* b, int N, int mode);
void tfo_external_product(uint32_t* out /*[K+1][N]*/, const uint32_t* trgsw /*[(K+1)L][K+1][N]*/,
                          const uint32_t* trlwe /*[K+1][N]*/, const tfo_params* p, int mode);
void tfo_cmux(uint32_t* out, const uint32_t* trgsw, const uint32_t* d0, const uint32_t* d1,
              const tfo_params* p, int mode);
/* trace (may be NULL): receives the accumulator after the initial rotation (step = -1)
 * and after every CMUX step. */
typedef void (*tfo_trace_fn)(void* user, int step, const uint32_t* acc /*[K+1][N]*/);
void tfo_blind_rotate(uint32_t* acc /*[K+1][N]*/, const uint32_t* c /*[n+1]*/, const uint32_t* bk,
                      const uint32_t* test_vector /*[K+1][N]*/, const tfo_params* p, int mode,
                      tfo_trace_fn trace, void* user);
void tfo_sample_extract_index(uint32_t* lv1 /*[N*K+1]*/, const uint32_t* trlwe, int x, const tfo_params* p);
void tfo_gate_bootstrapping_tlwe_to_tlwe(uint32_t* lv1, const uint32_t* c, const uint32_t* bk,
                                         const tfo_params* p, int mode);
void tfo_identity_key_switch(uint32_t* lv0 /*[n+1]*/, const uint32_t* lv1 /*[N*K+1]*/, const uint32_t* ks,
                             const tfo_params* p);
void tfo_nand_combine(uint32_t* c, const uint32_t* a, const uint32_t* b, const tfo_params* p);
void tfo_nand(uint32_t* out, const uint32_t* a, const uint32_t* b, const uint32_t* bk, const uint32_t* ks,
              const tfo_params* p, int mode);

/* Two-input gates other than NAND.  The reference has NAND only; every other gate is the same
 * bootstrap and key switch behind another linear pre-combination (reference src/hom_nand.hpp:76-78
 * is the NAND instance): c = (offset, 0, ..) + sa * a + sb * b with true = +1/8, false = -1/8
 * (reference src/types.hpp:25,34).  Gate codes as in include/tfhe_amd.h. */
enum { TFO_NAND = 0, TFO_AND, TFO_OR, TFO_NOR, TFO_XOR, TFO_XNOR, TFO_ANDNY, TFO_ANDYN, TFO_ORNY, TFO_ORYN,
       TFO_GATE_COUNT };
/* returns 0 and fills offset / sa / sb, or -1 for an unknown code */
int tfo_gate_coefficients(int op, uint32_t* offset, int32_t* sa, int32_t* sb);
int tfo_gate_combine(uint32_t* c, int op, const uint32_t* a, const uint32_t* b, const tfo_params* p);
void tfo_not(uint32_t* out, const uint32_t* a, const tfo_params* p);

/* ---- evaluation context with the bootstrapping key pre-transformed (NTT mode only) ---- */
typedef struct tfo_ctx tfo_ctx;
tfo_ctx* tfo_ctx_create(const tfo_params* p, const uint32_t* bk, const uint32_t* ks);
void tfo_ctx_destroy(tfo_ctx* ctx);
void tfo_ctx_bootstrap(const tfo_ctx* ctx, uint32_t* lv0_out, const uint32_t* c);
void tfo_ctx_blind_rotate(const tfo_ctx* ctx, uint32_t* acc, const uint32_t* c);
/* out[g] = NAND(a[g], b[g]) for g < n_gates; rows are n+1 words; threads<=0 -> all cores */
void tfo_ctx_nand_batch(const tfo_ctx* ctx, uint32_t* out, const uint32_t* a, const uint32_t* b,
                        size_t n_gates, int threads);
/* out[g] = keyswitch(bootstrap(c[g])) on pre-combined inputs */
void tfo_ctx_bootstrap_batch(const tfo_ctx* ctx, uint32_t* out, const uint32_t* c, size_t n_gates, int threads);
/* out[g] = gate ops[g] (or `op` for all when ops == NULL) of a[g], b[g] */
int tfo_ctx_gate_batch(const tfo_ctx* ctx, int op, const uint8_t* ops, uint32_t* out, const uint32_t* a,
                       const uint32_t* b, size_t n_gates, int threads);
/* out[g] = sel[g] ? x1[g] : x0[g]: two bootstraps without key switch (AND(sel, x1), ANDNY(sel, x0)),
 * their lv1 sum plus (1/8, 0, ..), one key switch */
void tfo_ctx_mux_batch(const tfo_ctx* ctx, uint32_t* out, const uint32_t* sel, const uint32_t* x1,
                       const uint32_t* x0, size_t n_gates, int threads);
int tfo_max_threads(void);

/* ---- seeded client side: key generation, encryption, decryption ---- */
typedef struct tfo_rng { uint64_t s[4]; int has_spare; double spare; } tfo_rng;
void tfo_rng_seed(tfo_rng* r, uint64_t seed);
uint64_t tfo_rng_u64(tfo_rng* r);
double tfo_rng_normal(tfo_rng* r);
uint32_t tfo_torus_from_double(double d); /* reference src/types.hpp:28-33 */
void tfo_keygen_secret(uint8_t* s_lv0 /*[n]*/, uint8_t* s_ring /*[K][N]*/, const tfo_params* p, tfo_rng* r);
void tfo_tlwe_encrypt(uint32_t* ct, uint32_t m, const uint8_t* s, int n, double alpha, tfo_rng* r);
uint32_t tfo_tlwe_phase(const uint32_t* ct, const uint8_t* s, int n);
void tfo_encrypt_bit(uint32_t* ct, int bit, const uint8_t* s_lv0, const tfo_params* p, tfo_rng* r);
int tfo_decrypt_bit(const uint32_t* ct, const uint8_t* s_lv0, const tfo_params* p);
void tfo_trlwe_encrypt(uint32_t* ct /*[K+1][N]*/, const uint32_t* m /*[N]*/, const uint8_t* s_ring,
                       const tfo_params* p, tfo_rng* r);
void tfo_trlwe_phase(uint32_t* m /*[N]*/, const uint32_t* ct, const uint8_t* s_ring, const tfo_params* p);
void tfo_trgsw_encrypt_int_poly(uint32_t* trgsw, const int32_t* mu /*[N]*/, const uint8_t* s_ring,
                                const tfo_params* p, tfo_rng* r);
void tfo_make_bk(uint32_t* bk, const uint8_t* s_lv0, const uint8_t* s_ring, const tfo_params* p, tfo_rng* r);
void tfo_make_ks(uint32_t* ks, const uint8_t* s_lv0, const uint8_t* s_ring, const tfo_params* p, tfo_rng* r);
/* Restatement of the GPU key generator: same key structure, masks and noise from ChaCha20 under a
 * 256-bit seed (stream, row and block index in the counter / nonce words). */
void tfo_chacha20_block(uint32_t out[16], const uint32_t seed[8], uint32_t stream, uint64_t row, uint32_t index);
void tfo_keygen_chacha_bk(uint32_t* bk, const uint8_t* s_lv0, const uint8_t* s_ring, const tfo_params* p,
                          double alpha, const uint32_t seed[8]);
void tfo_keygen_chacha_ks(uint32_t* ks, const uint8_t* s_lv0, const uint8_t* s_lv1 /*[N*K]*/, const tfo_params* p,
                          double alpha, const uint32_t seed[8]);

#ifdef __cplusplus
}
#endif
#endif
