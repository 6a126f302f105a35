// Launcher declarations shared by kernels.hip and the C ABI.
#pragma once

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "ntt512.hpp"

namespace tfhe_amd {

hipError_t launch_nand_combine(uint32_t* c, const uint32_t* a, const uint32_t* b, size_t n_gates, int n1,
                               hipStream_t stream);
// c = (offset, 0, ..) + sa * a + sb * b per gate; `ops` (device, one code per gate) overrides `op`
hipError_t launch_gate_combine(uint32_t* c, const uint32_t* a, const uint32_t* b, const uint8_t* ops, int op,
                               size_t n_gates, int n1, hipStream_t stream);
hipError_t launch_negate(uint32_t* out, const uint32_t* a, size_t words, hipStream_t stream);
// dst[j] += src[j] + (1/8, 0, ..) over n_rows rows of lv1_words words
hipError_t launch_mux_sum(uint32_t* dst, const uint32_t* src, size_t n_rows, int lv1_words, hipStream_t stream);
hipError_t launch_bk_transform_512(const NttUniform& u, const d2* tw, const uint32_t* bk, d2* bk_ntt, int n_polys,
                                   hipStream_t stream);
// tv: test vector [K+1][N] shared by all gates or nullptr (gate bootstrapping); extract != 0
// writes sample_extract_index(acc, 0), else the accumulator.  form: 0 = choose by batch size,
// 1 = one gate per workgroup (latency), 2 = one gate per wave (throughput)
hipError_t launch_blind_rotate_512(const NttUniform& u, const d2* tw, const d2* bk_ntt, const uint32_t* c,
                                   const uint32_t* tv, uint32_t* out, int n_gates, int n, int K, int L, int extract,
                                   int form, hipStream_t stream);
// blind rotation through a replay of the reference's FFT product (replica_kernels.hip); tables: [4][N/2]
// complex (twist, omega, untwist, reversed omega powers); spec: [polys][N/2] complex key spectra
bool replica_has_kernel(int N, int K, int L);
hipError_t launch_replica_key(const uint32_t* bk, const d2* tables, d2* spec, size_t n_polys, int N, int wave_order,
                              hipStream_t stream);
// generic != 0: the one-workgroup-per-gate kernel even where the wave-per-gate one exists (cross-checking)
bool replica_has_fast_kernel(int N, int K, int L, int bg);
hipError_t launch_blind_rotate_replica(const d2* spec, const d2* tables, const uint32_t* c, const uint32_t* tv,
                                       uint32_t* out, size_t n_gates, int n, int N, int K, int L, int bg, int extract,
                                       int generic, hipStream_t stream);
// evaluation-key generation (keygen_kernels.hip): raw keys in the flat layouts of tfhe_amd.h, device memory
struct KeygenSeed { uint32_t w[8]; };  // the 256-bit ChaCha20 key that every mask and noise word derives from
hipError_t launch_keygen_bk(uint32_t* bk, const uint8_t* s_lv0, const uint8_t* s_ring, int n, int N, int K, int L,
                            int bg_bit, double alpha, const KeygenSeed& seed, hipStream_t stream);
hipError_t launch_keygen_ks(uint32_t* ks, const uint8_t* s_lv0, const uint8_t* s_lv1, int n, int nlv1, int T, int bb,
                            double alpha, const KeygenSeed& seed, hipStream_t stream);
// stand-alone external product / cmux on the NTT kernels (kernels.hip, wg_kernels.hip); key_ntt: the
// batch's TRGSW samples transformed like a bootstrapping key of `batch` rows
bool external_product_512_has_kernel(int N, int K, int L, int bg);
hipError_t launch_external_product_512(const NttUniform& u, const d2* tw, const d2* key_ntt, const uint32_t* x0,
                                       const uint32_t* x1, uint32_t* out, size_t batch, int K, int L, int mode,
                                       hipStream_t stream);
hipError_t launch_external_product_wg(const NttUniform& u, const d2* table, const d2* key_ntt, const uint32_t* x0,
                                      const uint32_t* x1, uint32_t* out, size_t batch, int N, int K, int L, int bg,
                                      int shift, int pack_shift, int pack, int mode, hipStream_t stream);
// exact O(N^2) kernels for any parameter set (generic_kernels.hip).  Rings whose per-item state
// exceeds LDS need a workspace `ws` of generic_big_workspace_words() words (0 = none needed).
bool generic_has_kernel(int N, int K, int L);
size_t generic_big_workspace_words(size_t items, int N, int K, int L);
hipError_t launch_external_product_generic(const uint32_t* trgsw, const uint32_t* x0, const uint32_t* x1,
                                           uint32_t* out, size_t batch, int N, int K, int L, int bg, int mode,
                                           uint32_t* ws, hipStream_t stream);
hipError_t launch_blind_rotate_generic(const uint32_t* bk, const uint32_t* c, const uint32_t* tv, uint32_t* out,
                                       size_t n_gates, int n, int N, int K, int L, int bg, int extract,
                                       uint32_t* ws, hipStream_t stream);
bool key_switch_has_kernel(int n1, int T, int bb);
// the key-switching key is kept on the device without its digit-0 rows, rows padded to
// key_switch_row_words(n1) words
int key_switch_row_words(int n1);
size_t key_switch_packed_words(int n1, int nlv1, int T, int bb);
hipError_t launch_ks_pack(uint32_t* packed, const uint32_t* ks, int n1, int nlv1, int T, int bb, hipStream_t stream);
hipError_t launch_key_switch(const uint32_t* lv1, const uint32_t* ks_packed, uint32_t* out, int n_gates, int n1,
                             int nlv1, int T, int bb, hipStream_t stream);

// workgroup-cooperative NTT path for N >= 1024 (wg_kernels.hip)
bool wg_has_kernel(int N, int K, int L, int bg);
int wg_table_pairs(int N);
bool wg_split(int N, int K, int L, int bg, int* tiles, int* shift, int* pack_shift, int* pack);
hipError_t launch_bk_transform_wg(const NttUniform& u, const d2* table, const uint32_t* bk, d2* bk_ntt, size_t n_rows,
                                  int N, int K, int shift, int pack_shift, int pack, hipStream_t stream);
hipError_t launch_blind_rotate_wg(const NttUniform& u, const d2* table, const d2* bk_ntt, const uint32_t* c,
                                  const uint32_t* tv, uint32_t* out, size_t n_gates, int n, int N, int K, int L, int bg,
                                  int shift, int pack_shift, int pack, int extract, hipStream_t stream);

}  // namespace tfhe_amd
