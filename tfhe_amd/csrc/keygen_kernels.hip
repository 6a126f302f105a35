// Evaluation-key generation on the GPU: the bootstrapping key (reference TRGSW::make_bk,
// src/trgsw.hpp:61-82,119-129, over TRLWE::encrypt, src/trlwe.hpp:75-82) and the key-switching key
// (reference TLWEKeySwitching::make_ks, src/tlwe_key_switching.hpp:26-45, over TLWE::encrypt,
// src/tlwe.hpp:55-61) in the flat layouts of include/tfhe_amd.h.
//
// Randomness: the reference draws masks and noise from std::random_device (src/random.hpp:10) and
// cannot be reproduced; here every word comes from the ChaCha20 block function keyed by the
// caller's 256-bit seed, with the block position (block index, stream, row) in the counter and
// nonce words, so a key set is a pure function of (secret key, seed) whatever the launch geometry,
// and as unpredictable as the seed is.  Distributions follow the reference: uniform 32-bit masks
// (src/random.hpp:39-41), Gaussian noise of standard deviation alpha converted to the torus with
// the truncating double -> torus rule (src/types.hpp:28-33).
#include "kernels.hpp"

namespace tfhe_amd {

__device__ __forceinline__ uint32_t rotl32(uint32_t v, int c) { return (v << c) | (v >> (32 - c)); }

#define TFHE_AMD_CHACHA_QR(a, b, c, d) \
    a += b; d ^= a; d = rotl32(d, 16); \
    c += d; b ^= c; b = rotl32(b, 12); \
    a += b; d ^= a; d = rotl32(d, 8);  \
    c += d; b ^= c; b = rotl32(b, 7);

// streams: which part of the key set a block belongs to
enum : uint32_t { kStreamBkMask = 1, kStreamBkNoise = 2, kStreamKsMask = 3, kStreamKsNoise = 4 };

// One ChaCha20 block (RFC 8439 block function: 20 rounds, feed-forward): state = constants | key[8] |
// (index, stream, row low, row high).  16 output words.
__device__ __forceinline__ void chacha20_block(uint32_t (&out)[16], const KeygenSeed& seed, uint32_t stream, uint64_t row,
                                               uint32_t index) {
    uint32_t in[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u,
                       seed.w[0], seed.w[1], seed.w[2], seed.w[3], seed.w[4], seed.w[5], seed.w[6], seed.w[7],
                       index, stream, (uint32_t)row, (uint32_t)(row >> 32)};
    uint32_t x[16];
#pragma unroll
    for (int i = 0; i < 16; i++) x[i] = in[i];
#pragma unroll
    for (int r = 0; r < 10; r++) {
        TFHE_AMD_CHACHA_QR(x[0], x[4], x[8], x[12])
        TFHE_AMD_CHACHA_QR(x[1], x[5], x[9], x[13])
        TFHE_AMD_CHACHA_QR(x[2], x[6], x[10], x[14])
        TFHE_AMD_CHACHA_QR(x[3], x[7], x[11], x[15])
        TFHE_AMD_CHACHA_QR(x[0], x[5], x[10], x[15])
        TFHE_AMD_CHACHA_QR(x[1], x[6], x[11], x[12])
        TFHE_AMD_CHACHA_QR(x[2], x[7], x[8], x[13])
        TFHE_AMD_CHACHA_QR(x[3], x[4], x[9], x[14])
    }
#pragma unroll
    for (int i = 0; i < 16; i++) out[i] = x[i] + in[i];
}

// reference src/types.hpp:28-33: fmod, +1 if negative, times 2^32, truncate
__device__ __forceinline__ uint32_t torus_from_double(double d) {
    d = fmod(d, 1.0);
    if (d < 0.0) d += 1.0;
    return (uint32_t)(unsigned long long)(d * 4294967296.0);
}

// two standard normals from four words (Box-Muller on 53- and 52-bit uniforms)
__device__ __forceinline__ void normal2(const uint32_t* w, double& z0, double& z1) {
    const double u1 = ((double)(((uint64_t)w[0] << 21) ^ (w[1] >> 11)) + 1.0) * (1.0 / 9007199254740992.0);  // (0, 1]
    const double u2 = (double)(((uint64_t)w[2] << 20) ^ (w[3] >> 12)) * (1.0 / 4503599627370496.0);        // [0, 1)
    const double r = sqrt(-2.0 * log(u1));
    double s, c;
    sincospi(2.0 * u2, &s, &c);
    z0 = r * c;
    z1 = r * s;
}

// One workgroup per TRLWE row of the bootstrapping key: row id = i * R + r, r = comp * L + level.
// Output: bk[row][c][N], c = 0 the b polynomial, c = 1..K the mask polynomials.
// b = e + sum_k a_k (*) s_k (negacyclic, s binary), then s0_i * 2^(32 - bg (level + 1)) is added to
// coefficient 0 of polynomial `comp`.  Word j of mask polynomial k is word j % 16 of block j / 16 of
// (stream mask, row * K + k); noise coefficient j is normal j % 2 of the Box-Muller pair made from
// words 4 (j / 2 % 4) .. + 3 of block j / 8 of (stream noise, row).
__global__ void __launch_bounds__(256)
keygen_bk_kernel(uint32_t* __restrict__ bk, const uint8_t* __restrict__ s_lv0, const uint8_t* __restrict__ s_ring,
                 int N, int K, int L, int bg_bit, double alpha, const KeygenSeed seed) {
    extern __shared__ uint32_t s_mem[];  // a[N] | s bits as words [N]
    uint32_t* s_a = s_mem;
    uint32_t* s_s = s_mem + N;
    const int K1 = K + 1, R = K1 * L;
    const uint64_t row = blockIdx.x;
    const int i = (int)(row / R), r = (int)(row % R), comp = r / L, level = r % L;
    uint32_t* out = bk + row * (size_t)K1 * N;
    const int t = threadIdx.x, T = blockDim.x;

    // noise into b (eight normals per block)
    for (int j = 8 * t; j < N; j += 8 * T) {
        uint32_t w[16];
        chacha20_block(w, seed, kStreamBkNoise, row, (uint32_t)(j >> 3));
#pragma unroll
        for (int e = 0; e < 4; e++) {
            double z0, z1;
            normal2(w + 4 * e, z0, z1);
            if (j + 2 * e < N) out[j + 2 * e] = torus_from_double(z0 * alpha);
            if (j + 2 * e + 1 < N) out[j + 2 * e + 1] = torus_from_double(z1 * alpha);
        }
    }
    for (int k = 0; k < K; k++) {
        __syncthreads();
        for (int j = 16 * t; j < N; j += 16 * T) {
            uint32_t w[16];
            chacha20_block(w, seed, kStreamBkMask, row * (uint64_t)K + k, (uint32_t)(j >> 4));
#pragma unroll
            for (int e = 0; e < 16; e++)
                if (j + e < N) {
                    s_a[j + e] = w[e];
                    out[(size_t)(1 + k) * N + j + e] = w[e];
                }
        }
        for (int j = t; j < N; j += T) s_s[j] = s_ring[(size_t)k * N + j];
        __syncthreads();
        // b[j] += sum_m s[m] * (+-) a[(j - m) mod N]
        for (int j = t; j < N; j += T) {
            uint32_t acc = 0;
            for (int m = 0; m < N; m++) {
                const int idx = j - m;
                const uint32_t v = s_a[idx & (N - 1)];
                const uint32_t term = idx < 0 ? 0u - v : v;
                acc += s_s[m] ? term : 0u;
            }
            out[j] += acc;
        }
    }
    __syncthreads();
    if (t == 0 && s_lv0[i]) out[(size_t)comp * N] += 1u << (32 - bg_bit * (level + 1));
}

// One wave per row (i, m, o) with o >= 1 of the key-switching key; rows with o = 0 stay zero (the
// reference never writes or reads them).  row = [b, a_0 .. a_{n-1}],
// b = e + sum_j a_j s0_j + o * s1_i * 2^(32 - bb (m + 1)).  Mask word j is word j % 16 of block j / 16
// of (stream mask, row); the noise is the first normal of block 0 of (stream noise, row).
__global__ void __launch_bounds__(256)
keygen_ks_kernel(uint32_t* __restrict__ ks, const uint8_t* __restrict__ s_lv0, const uint8_t* __restrict__ s_lv1,
                 size_t n_rows, int n, int T, int bb, double alpha, const KeygenSeed seed) {
    const int lane = threadIdx.x & 63;
    const size_t row = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n_rows) return;
    const int o = (int)(row & ((1u << bb) - 1));
    uint32_t* out = ks + row * (size_t)(n + 1);
    if (o == 0) {
        for (int j = lane; j <= n; j += 64) out[j] = 0u;
        return;
    }
    const size_t im = row >> bb;
    const int m = (int)(im % T);
    const size_t i = im / T;
    uint32_t acc = 0;
    for (int j0 = 0; j0 < n; j0 += 1024) {
        const int j = j0 + 16 * lane;
        if (j < n) {
            uint32_t w[16];
            chacha20_block(w, seed, kStreamKsMask, row, (uint32_t)(j >> 4));
#pragma unroll
            for (int e = 0; e < 16; e++)
                if (j + e < n) {
                    out[1 + j + e] = w[e];
                    if (s_lv0[j + e]) acc += w[e];
                }
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if (lane == 0) {
        uint32_t w[16];
        chacha20_block(w, seed, kStreamKsNoise, row, 0u);
        double z0, z1;
        normal2(w, z0, z1);
        const uint32_t msg = s_lv1[i] ? (uint32_t)o << (32 - (m + 1) * bb) : 0u;
        out[0] = acc + torus_from_double(z0 * alpha) + msg;
    }
}

hipError_t launch_keygen_bk(uint32_t* bk, const uint8_t* s_lv0, const uint8_t* s_ring, int n, int N, int K, int L,
                            int bg_bit, double alpha, const KeygenSeed& seed, hipStream_t stream) {
    const size_t rows = (size_t)n * (K + 1) * L;
    if (rows == 0) return hipSuccess;
    const int threads = N >= 256 ? 256 : 64;
    keygen_bk_kernel<<<(unsigned)rows, threads, 2 * (size_t)N * sizeof(uint32_t), stream>>>(bk, s_lv0, s_ring, N, K, L,
                                                                                          bg_bit, alpha, seed);
    return hipGetLastError();
}

hipError_t launch_keygen_ks(uint32_t* ks, const uint8_t* s_lv0, const uint8_t* s_lv1, int n, int nlv1, int T, int bb,
                            double alpha, const KeygenSeed& seed, hipStream_t stream) {
    const size_t rows = (size_t)nlv1 * T << bb;
    if (rows == 0) return hipSuccess;
    keygen_ks_kernel<<<(unsigned)((rows + 3) / 4), 256, 0, stream>>>(ks, s_lv0, s_lv1, rows, n, T, bb, alpha, seed);
    return hipGetLastError();
}

}  // namespace tfhe_amd
