/*
 * tfhe_amd -- C ABI of the MI355X gate-bootstrapping library.
 *
 * The reference (KowerKoint/TFHE) is a header-only C++ template library with no FFI; the
 * entry points below are what its hot path, HomNAND::nand (reference src/hom_nand.hpp:74-81)
 * and the functions under it, bind to when evaluation is moved to the GPU.  All pointers are
 * plain arrays of uint32_t torus words in the flat layouts described next; no C++ or torch
 * type crosses this boundary.  Every call returns an int status (0 = success) and never
 * throws.  A context may be used from several host threads at once, as the reference's nand() may
 * be called concurrently on one shared EvaluateKey: every call that needs scratch memory takes one
 * of up to eight workspaces of the context for its duration (host-pointer calls) or while it
 * enqueues (device-pointer calls; calls that reuse a workspace on another stream are ordered one
 * behind the other).  tfhe_amd_ctx_destroy must not race with calls on the same context.
 *
 * Flat layouts (row-major):
 *   lv0 ciphertext    [n+1]                      = [b, a_0 .. a_{n-1}]   (reference src/hom_nand.hpp:43,
 *                                                                         src/tlwe.hpp:33-39)
 *   lv1 ciphertext    [N*K+1]                                            (reference src/trlwe.hpp:109-122)
 *   bootstrapping key [n][(K+1)*L][K+1][N]       row = comp*L + level    (reference src/hom_nand.hpp:37,
 *                                                                         src/trgsw.hpp:61-82,119-129)
 *   key-switching key [N*K][T][1<<base_bit][n+1] slot o = 0 is never read (reference src/hom_nand.hpp:38-41,
 *                                                                         src/tlwe_key_switching.hpp:26-45)
 */
#ifndef TFHE_AMD_H
#define TFHE_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Run-time mirror of the reference's compile-time parameter structs
 * (src/tlwe.hpp:11-14, src/trlwe.hpp:9-18, src/trgsw.hpp:6-10, src/tlwe_key_switching.hpp:6-9). */
typedef struct tfhe_amd_params {
    int32_t n;        /* lv0 TLWE dimension, default 636 */
    int32_t N;        /* ring degree, default 512 */
    int32_t K;        /* TRLWE mask polynomials, default 2 */
    int32_t bg_bit;   /* gadget base bits, default 8 */
    int32_t L;        /* gadget levels, default 2 */
    int32_t T;        /* key-switch levels, default 5 */
    int32_t base_bit; /* key-switch base bits, default 2 */
} tfhe_amd_params;

enum {
    TFHE_AMD_OK = 0,
    TFHE_AMD_ERR_ARGUMENT = 1,    /* null pointer, bad size */
    TFHE_AMD_ERR_UNSUPPORTED = 2, /* parameter set has no kernel */
    TFHE_AMD_ERR_HIP = 3,         /* a HIP call failed; see tfhe_amd_last_error */
    TFHE_AMD_ERR_MEMORY = 4       /* device allocation failed */
};

typedef struct tfhe_amd_ctx tfhe_amd_ctx;

/* `stream` arguments of the _device entry points: NULL = the context's own (non-blocking)
 * stream; TFHE_AMD_DEFAULT_STREAM = the device's default (null) stream, whose handle is itself
 * NULL and therefore needs a name of its own; anything else = a hipStream_t.  A call is ordered
 * only against work on the stream it is given: pass the stream that produces the input arrays, or
 * synchronise that stream before calling on another one. */
#define TFHE_AMD_DEFAULT_STREAM ((void*)(intptr_t)-1)

/* Fills *p with the reference's default parameter set. */
void tfhe_amd_params_default(tfhe_amd_params* p);
/* 0 if kernels exist for *p, else TFHE_AMD_ERR_UNSUPPORTED / TFHE_AMD_ERR_ARGUMENT. */
int tfhe_amd_params_supported(const tfhe_amd_params* p);
size_t tfhe_amd_bk_words(const tfhe_amd_params* p);
size_t tfhe_amd_ks_words(const tfhe_amd_params* p);

enum {
    TFHE_AMD_CTX_DEFAULT = 0,
    TFHE_AMD_CTX_KEYS_ON_DEVICE = 1,  /* bk / ks point to memory of GPU `device` (e.g. after a broadcast).  The
                                         context reads them on a stream of its own: whatever wrote them
                                         must have finished (synchronise the producing stream first) */
    TFHE_AMD_CTX_GENERIC_KERNELS = 2, /* use the parameter-generic exact O(N^2) kernels even where a
                                         specialised NTT kernel exists (cross-checking) */
    /* The specialised blind rotation has two forms with identical results: one gate per wave
     * (throughput) and one gate per workgroup (latency).  By default batches of at most 512 gates
     * take the latency form; these flags pin one form.  They apply to the "ntt512" kernel family
     * (tfhe_amd_ctx_kernel_family) and are ignored by the others, which have one form. */
    TFHE_AMD_CTX_LATENCY_FORM = 4,
    TFHE_AMD_CTX_THROUGHPUT_FORM = 8,
    /* Verification mode.  The reference multiplies polynomials with a double-precision FFT
     * (src/types.hpp:237-292) whose result is the exact product or one unit lower; by default this
     * library returns the exact product.  With this flag the blind rotation replays the reference's
     * FFT operation by operation, rounding included, and the output ciphertexts are bit-identical to
     * those of the reference compiled without FMA contraction (g++ -O2 on x86-64).  Ring degrees
     * 32 .. 1024; much slower than the default path. */
    TFHE_AMD_CTX_REFERENCE_FFT = 16
};

/* Replaces holding an EvaluateKey (reference src/hom_nand.hpp:36-42): uploads the keys to GPU
 * `device`, transforms the bootstrapping key to the NTT domain (specialised parameter sets) and
 * re-packs the key-switching key.  The caller's arrays may be freed after the call.  Either
 * key may be NULL; the stages that need it then return TFHE_AMD_ERR_ARGUMENT. */
int tfhe_amd_ctx_create(const tfhe_amd_params* p, const uint32_t* bk, const uint32_t* ks, int device,
                        unsigned flags, tfhe_amd_ctx** out);
void tfhe_amd_ctx_destroy(tfhe_amd_ctx* ctx);

/* Evaluation-key generation on the GPU: HomNAND::make_evaluate_key (reference src/hom_nand.hpp:56-64),
 * i.e. TRGSW::make_bk (src/trgsw.hpp:61-82,119-129) and TLWEKeySwitching::make_ks
 * (src/tlwe_key_switching.hpp:26-45).  s_lv0: [n] bytes 0/1 (SecretKey::tlwe_lv0_s); s_ring:
 * [K][N] bytes 0/1 (SecretKey::trlwe_s), both in HOST memory.  alpha_lv0 / alpha_lv1: noise standard
 * deviations of the lv0 TLWE and of the ring (reference src/tlwe.hpp:13, src/trlwe.hpp:12).  Every
 * mask word and every noise sample is derived from ChaCha20 output under the 256-bit key `seed` (eight
 * words), and the masks are as unpredictable as the seed, which must therefore be fresh secret
 * randomness (the reference draws every word from std::random_device, src/random.hpp:10,35-41); a
 * fixed seed is for tests.  On one GPU and library build the result is a pure function of (secret
 * key, seed).  The mask words are pure integer functions of the seed; a noise word goes through
 * log, sin and cos (Box-Muller) and may differ by one unit of 2^-32 between math libraries, so the
 * b words are reproducible across machines only up to that unit.  Generated keys cannot be compared
 * with reference bytes (the reference is not reproducible); they are checked against the CPU oracle's
 * restatement of the same generator (masks identical, noise words within one unit) and through
 * their decryption.
 *   _keygen:        raw keys to HOST arrays bk / ks (either may be NULL)
 *   _keygen_device: raw keys to DEVICE arrays on GPU `device`, synchronised before returning
 *   _ctx_create_from_secret: keys generated on the GPU and turned into a context without
 *                   ever leaving it */
int tfhe_amd_keygen(const tfhe_amd_params* p, const uint8_t* s_lv0, const uint8_t* s_ring, double alpha_lv0,
                    double alpha_lv1, const uint32_t* seed /*[8]*/, int device, uint32_t* bk, uint32_t* ks);
int tfhe_amd_keygen_device(const tfhe_amd_params* p, const uint8_t* s_lv0, const uint8_t* s_ring, double alpha_lv0,
                           double alpha_lv1, const uint32_t* seed /*[8]*/, int device, uint32_t* bk, uint32_t* ks,
                           void* stream);
int tfhe_amd_ctx_create_from_secret(const tfhe_amd_params* p, const uint8_t* s_lv0, const uint8_t* s_ring,
                                    double alpha_lv0, double alpha_lv1, const uint32_t* seed /*[8]*/, int device,
                                    unsigned flags, tfhe_amd_ctx** out);

/* HomNAND::nand over a batch (reference src/hom_nand.hpp:74-81): out[g] = NAND(a[g], b[g]).
 * a, b, out: [n_gates][n+1] in HOST memory.  Synchronous. */
int tfhe_amd_nand_batch(tfhe_amd_ctx* ctx, const uint32_t* a, const uint32_t* b, uint32_t* out, size_t n_gates);
/* Same with all three arrays in DEVICE memory; enqueued on `stream` (a hipStream_t, NULL =
 * the context's own stream) and not synchronised. */
int tfhe_amd_nand_batch_device(tfhe_amd_ctx* ctx, const uint32_t* a, const uint32_t* b, uint32_t* out,
                               size_t n_gates, void* stream);

/* No counterpart in the reference.  tfhe_amd_nand_batch_device for callers that cannot allocate:
 * code under HIP graph capture, serving loops and harnesses that forbid hipMalloc.  Same arguments,
 * same results, but the call performs no device or host allocation: its workspace (the combined
 * inputs and the lv1 ciphertexts, (n+1) + (N*K+1) words per gate) is a static device array of
 * TFHE_AMD_NOALLOC_SCRATCH_BYTES that every GPU the library runs on holds until the process ends.
 * Batches whose workspace exceeds it (more than 2,523 gates at the default parameters) are rejected
 * with TFHE_AMD_ERR_ARGUMENT, and so are parameter sets whose ring does not fit LDS (N > 2048: their
 * kernels take a workspace per launch; TFHE_AMD_ERR_UNSUPPORTED).  All contexts on one GPU share that GPU's array, which keeps the last
 * call's contents.  Calls through this entry point on one GPU are ordered one behind the other even
 * on different streams (an event created with the first context on that GPU: recording it and
 * waiting for it allocate nothing).  Two cases stay with the caller: a call on a stream under
 * capture records no such ordering into the graph, so launches of that graph must not overlap other
 * users of the array on the same GPU; and callers in different processes share nothing.  Kernel
 * timing (tfhe_amd_ctx_enable_timing) does not record these calls. */
#define TFHE_AMD_NOALLOC_SCRATCH_BYTES ((size_t)16 << 20)
int tfhe_amd_nand_batch_device_noalloc(tfhe_amd_ctx* ctx, const uint32_t* a, const uint32_t* b, uint32_t* out,
                                       size_t n_gates, void* stream);

/* The other two-input gates.  The reference has NAND only; each gate below is the same bootstrap
 * and key switch behind another linear pre-combination, c = (offset, 0, ..) + sa*a + sb*b, of which
 * reference src/hom_nand.hpp:76-78 is the NAND instance (true = +1/8, false = -1/8, reference
 * src/types.hpp:25,34).  ANDNY = (not a) and b, ANDYN = a and (not b), likewise ORNY / ORYN. */
enum {
    TFHE_AMD_GATE_NAND = 0, /* ( 1/8) - a - b */
    TFHE_AMD_GATE_AND = 1,  /* (-1/8) + a + b */
    TFHE_AMD_GATE_OR = 2,   /* ( 1/8) + a + b */
    TFHE_AMD_GATE_NOR = 3,  /* (-1/8) - a - b */
    TFHE_AMD_GATE_XOR = 4,  /* ( 1/4) + 2(a + b) */
    TFHE_AMD_GATE_XNOR = 5, /* (-1/4) - 2(a + b) */
    TFHE_AMD_GATE_ANDNY = 6,
    TFHE_AMD_GATE_ANDYN = 7,
    TFHE_AMD_GATE_ORNY = 8,
    TFHE_AMD_GATE_ORYN = 9,
    TFHE_AMD_GATE_COUNT = 10
};
/* out[g] = op(a[g], b[g]); same layouts and memory rules as tfhe_amd_nand_batch[_device]. */
int tfhe_amd_gate_batch(tfhe_amd_ctx* ctx, int op, const uint32_t* a, const uint32_t* b, uint32_t* out,
                        size_t n_gates);
int tfhe_amd_gate_batch_device(tfhe_amd_ctx* ctx, int op, const uint32_t* a, const uint32_t* b, uint32_t* out,
                               size_t n_gates, void* stream);
/* One gate code per gate: ops [n_gates] bytes in DEVICE memory (a dependency level of a circuit with
 * mixed gate types in one call).  A code outside the enum evaluates as NAND. */
int tfhe_amd_gates_batch_device(tfhe_amd_ctx* ctx, const uint8_t* ops, const uint32_t* a, const uint32_t* b,
                                uint32_t* out, size_t n_gates, void* stream);
/* out[g] = not a[g] = -a[g]; no bootstrap, noise unchanged. */
int tfhe_amd_not_batch(tfhe_amd_ctx* ctx, const uint32_t* a, uint32_t* out, size_t n_gates);
int tfhe_amd_not_batch_device(tfhe_amd_ctx* ctx, const uint32_t* a, uint32_t* out, size_t n_gates, void* stream);
/* out[g] = sel[g] ? x1[g] : x0[g]: the lv1 results of AND(sel, x1) and ANDNY(sel, x0) are added,
 * plus (1/8, 0, ..), and key-switched once: two blind rotations, one key switch per gate. */
int tfhe_amd_mux_batch(tfhe_amd_ctx* ctx, const uint32_t* sel, const uint32_t* x1, const uint32_t* x0, uint32_t* out,
                       size_t n_gates);
int tfhe_amd_mux_batch_device(tfhe_amd_ctx* ctx, const uint32_t* sel, const uint32_t* x1, const uint32_t* x0,
                              uint32_t* out, size_t n_gates, void* stream);

/* One dependency level of a circuit in one call: gates [0, n_gates - n_mux) are two-input gates
 * with codes ops[g] on rows a[g], b[g]; the last n_mux gates are MUXes with sel = a[g], x1 = b[g]
 * and x0 = x0[g - (n_gates - n_mux)] (x0: [n_mux][n+1]).  All n_gates + n_mux blind rotations run
 * in one launch, so a level costs one blind-rotation latency whatever its mix.  ops: device
 * memory, [n_gates - n_mux] bytes (may be NULL when every gate is a MUX). */
int tfhe_amd_level_batch_device(tfhe_amd_ctx* ctx, const uint8_t* ops, const uint32_t* a, const uint32_t* b,
                                const uint32_t* x0, uint32_t* out, size_t n_gates, size_t n_mux, void* stream);

/* gate_bootstrapping_tlwe_to_tlwe followed by identity_key_switch on pre-combined inputs
 * (reference src/hom_nand.hpp:79-80): out[g] = KS(bootstrap(c[g])).  The building block
 * of every other two-input gate. */
int tfhe_amd_bootstrap_batch(tfhe_amd_ctx* ctx, const uint32_t* c, uint32_t* out, size_t n_gates);
int tfhe_amd_bootstrap_batch_device(tfhe_amd_ctx* ctx, const uint32_t* c, uint32_t* out, size_t n_gates,
                                    void* stream);

/* TRGSW::gate_bootstrapping_tlwe_to_tlwe alone (reference src/trgsw.hpp:172-182):
 * lv1[g] = sample_extract_index(blind_rotate(c[g], bk, testvector), 0); lv1: [n_gates][N*K+1]. */
int tfhe_amd_gate_bootstrapping_batch(tfhe_amd_ctx* ctx, const uint32_t* c, uint32_t* lv1, size_t n_gates);
int tfhe_amd_gate_bootstrapping_batch_device(tfhe_amd_ctx* ctx, const uint32_t* c, uint32_t* lv1, size_t n_gates,
                                             void* stream);
/* TRGSW::blind_rotate (reference src/trgsw.hpp:131-170): acc[g] = CMUX chain over c[g] starting
 * from test_vector * X^(-floor(b 2N)).  test_vector: [K+1][N] shared by all gates, NULL = the
 * gate-bootstrapping one.  acc: [n_gates][K+1][N]. */
int tfhe_amd_blind_rotate_batch(tfhe_amd_ctx* ctx, const uint32_t* c, const uint32_t* test_vector, uint32_t* acc,
                                size_t n_gates);
int tfhe_amd_blind_rotate_batch_device(tfhe_amd_ctx* ctx, const uint32_t* c, const uint32_t* test_vector,
                                       uint32_t* acc, size_t n_gates, void* stream);
/* TLWEKeySwitching::identity_key_switch alone (reference src/tlwe_key_switching.hpp:47-65). */
int tfhe_amd_identity_key_switch_batch(tfhe_amd_ctx* ctx, const uint32_t* lv1, uint32_t* out, size_t n_gates);
int tfhe_amd_identity_key_switch_batch_device(tfhe_amd_ctx* ctx, const uint32_t* lv1, uint32_t* out,
                                              size_t n_gates, void* stream);

/* TRGSW::external_product / TRGSW::cmux on explicit operands (reference src/trgsw.hpp:100-117),
 * no context: trgsw [batch][(K+1)L][K+1][N], TRLWE operands and out [batch][K+1][N], all in HOST
 * memory; evaluated on GPU `device`.  cmux: out = trgsw (x) (d1 - d0) + d0.  Parameter sets with NTT
 * kernels (tfhe_amd_ctx_kernel_family) transform the samples and multiply in the NTT domain, the
 * others use the exact O(N^2) kernels.  Staging buffers and tables are kept per device for the life
 * of the process; calls on one device run one at a time. */
int tfhe_amd_external_product_batch(const tfhe_amd_params* p, const uint32_t* trgsw, const uint32_t* trlwe,
                                    uint32_t* out, size_t batch, int device);
int tfhe_amd_cmux_batch(const tfhe_amd_params* p, const uint32_t* trgsw, const uint32_t* d0, const uint32_t* d1,
                        uint32_t* out, size_t batch, int device);

/* Several GPUs of one node behind one handle, for a single-process caller (the reference's only
 * kind: test/main.cpp:180-224).  Gates share nothing but the read-only keys (reference
 * src/hom_nand.hpp:74-81), so a batch is split in contiguous blocks, device g of P evaluating gates
 * [g G / P, (g + 1) G / P), with no exchange between devices while evaluating.  device_ids lists the
 * GPUs (an id may appear more than once: each entry gets a context of its own).  The raw keys are
 * uploaded to the first device once (with TFHE_AMD_CTX_KEYS_ON_DEVICE they are already there),
 * copied device to device to the others, and every device transforms its own copy. */
typedef struct tfhe_amd_multi_ctx tfhe_amd_multi_ctx;
int tfhe_amd_multi_ctx_create(const tfhe_amd_params* p, const uint32_t* bk, const uint32_t* ks, const int* device_ids,
                              int n_devices, unsigned flags, tfhe_amd_multi_ctx** out);
void tfhe_amd_multi_ctx_destroy(tfhe_amd_multi_ctx* m);
int tfhe_amd_multi_ctx_device_count(const tfhe_amd_multi_ctx* m);
/* The context of entry `index` of the device list (owned by m), for the single-device calls above. */
tfhe_amd_ctx* tfhe_amd_multi_ctx_get(tfhe_amd_multi_ctx* m, int index);
/* The block of a batch of n_gates that entry `index` evaluates: gates [*begin, *end). */
void tfhe_amd_multi_shard(const tfhe_amd_multi_ctx* m, size_t n_gates, int index, size_t* begin, size_t* end);
/* HomNAND::nand / the other gates / the bare bootstrap over a batch in HOST memory, all devices
 * working side by side (one host thread per device inside the call).  Synchronous. */
int tfhe_amd_multi_nand_batch(tfhe_amd_multi_ctx* m, const uint32_t* a, const uint32_t* b, uint32_t* out,
                              size_t n_gates);
int tfhe_amd_multi_gate_batch(tfhe_amd_multi_ctx* m, int op, const uint32_t* a, const uint32_t* b, uint32_t* out,
                              size_t n_gates);
int tfhe_amd_multi_bootstrap_batch(tfhe_amd_multi_ctx* m, const uint32_t* c, uint32_t* out, size_t n_gates);
/* Device-resident form: a[g], b[g], out[g] are arrays of n_gates[g] ciphertexts in the memory of
 * entry g's GPU.  streams[g] is the stream entry g's share is enqueued on, with the meaning of the
 * `stream` argument of the single-device calls (NULL = that context's own stream,
 * TFHE_AMD_DEFAULT_STREAM, or a hipStream_t of entry g's GPU); streams == NULL means every context's
 * own stream.  A share is ordered only against work on the stream it is given: pass the streams that
 * produce a[g] and b[g] and consume out[g], or synchronise those before the call and
 * tfhe_amd_multi_ctx_synchronize (which waits for the contexts' own streams only) after it. */
int tfhe_amd_multi_nand_batch_device(tfhe_amd_multi_ctx* m, const uint32_t* const* a, const uint32_t* const* b,
                                     uint32_t* const* out, const size_t* n_gates, void* const* streams);
int tfhe_amd_multi_ctx_synchronize(tfhe_amd_multi_ctx* m);

/* Which blind-rotation kernels a context uses:
 *   "ntt512"         N = 512 with L = 2 (the default set, and K = 1): wave-per-gate and
 *                    workgroup-per-gate NTT kernels
 *   "ntt-workgroup"  N = 2048, N = 1024 and N = 512 with K = 1 or 2: 8-bit gadget with L = 2 or 3
 *                    (except N = 512, L = 2: "ntt512"), 16-bit gadget with L = 1
 *   "generic"        exact O(N^2) kernels, any parameter set
 *   "reference-fft-wave" / "reference-fft-workgroup"   verification mode
 *   "none"           the context holds no bootstrapping key
 * For logs and for tests that must not pass on a slower path by accident. */
const char* tfhe_amd_ctx_kernel_family(const tfhe_amd_ctx* ctx);

/* Blocks until everything enqueued on the context's own stream has finished. */
int tfhe_amd_ctx_synchronize(tfhe_amd_ctx* ctx);

/* Kernel timing with HIP events on the stream the kernels run on.  While enabled, every
 * batch call records events around its blind-rotation and key-switch kernels.
 * tfhe_amd_ctx_timing waits for the recorded calls and returns the sums since the last
 * query: milliseconds in each kernel and the number of batch calls. */
int tfhe_amd_ctx_enable_timing(tfhe_amd_ctx* ctx, int enable);
int tfhe_amd_ctx_timing(tfhe_amd_ctx* ctx, double* blind_rotate_ms, double* key_switch_ms, int64_t* calls);

/* Text of the last error recorded on this thread (empty string if none). */
const char* tfhe_amd_last_error(void);
/* Library version, "major.minor.patch". */
const char* tfhe_amd_version(void);

#ifdef __cplusplus
}
#endif
#endif
