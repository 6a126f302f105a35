// Evaluation-key generation on the GPU: the bootstrapping key (reference TRGSW::make_bk,
// src/trgsw.hpp:61-82,119-129, over TRLWE::encrypt, src/trlwe.hpp:75-82) and the key-switching key
// (reference TLWEKeySwitching::make_ks, src/tlwe_key_switching.hpp:26-45, over TLWE::encrypt,
// src/tlwe.hpp:55-61) in the flat layouts of include/tfhe_amd.h.
//
// Randomness: the reference draws masks and noise from std::random_device (src/random.hpp:10) and
// cannot be reproduced; here every word comes from Philox-4x32-10 keyed by the caller's seed and
// counted by (stream, row, index), so a key set is a pure function of (secret key, seed) whatever
// the launch geometry.  Distributions follow the reference: uniform 32-bit masks
// (src/random.hpp:39-41), Gaussian noise of standard deviation alpha converted to the torus with
// the truncating double -> torus rule (src/types.hpp:28-33).
#include "kernels.hpp"

namespace tfhe_amd {

struct Philox {
    uint32_t k0, k1;
    __device__ __forceinline__ void round(uint32_t (&c)[4], uint32_t a, uint32_t b) const {
        const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
        const uint32_t hi0 = __umulhi(M0, c[0]), lo0 = M0 * c[0];
        const uint32_t hi1 = __umulhi(M1, c[2]), lo1 = M1 * c[2];
        const uint32_t n0 = hi1 ^ c[1] ^ a, n1 = lo1, n2 = hi0 ^ c[3] ^ b, n3 = lo0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
    }
    // four words for counter (c0, c1, c2, c3)
    __device__ __forceinline__ void block(uint32_t (&c)[4]) const {
        uint32_t a = k0, b = k1;
#pragma unroll
        for (int r = 0; r < 10; r++) {
            round(c, a, b);
            a += 0x9E3779B9u;
            b += 0xBB67AE85u;
        }
    }
};

// counter layout: (index, row low, row high, stream)
enum : uint32_t { kStreamBkMask = 1, kStreamBkNoise = 2, kStreamKsMask = 3, kStreamKsNoise = 4 };

__device__ __forceinline__ void philox4(uint32_t (&out)[4], uint64_t seed, uint32_t stream, uint64_t row, uint32_t index) {
    Philox ph{(uint32_t)seed, (uint32_t)(seed >> 32)};
    out[0] = index; out[1] = (uint32_t)row; out[2] = (uint32_t)(row >> 32); out[3] = stream;
    ph.block(out);
}

// reference src/types.hpp:28-33: fmod, +1 if negative, times 2^32, truncate
__device__ __forceinline__ uint32_t torus_from_double(double d) {
    d = fmod(d, 1.0);
    if (d < 0.0) d += 1.0;
    return (uint32_t)(unsigned long long)(d * 4294967296.0);
}

// two standard normals from four words (Box-Muller on 53- and 52-bit uniforms)
__device__ __forceinline__ void normal2(const uint32_t (&w)[4], double& z0, double& z1) {
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
// coefficient 0 of polynomial `comp`.
__global__ void __launch_bounds__(256)
keygen_bk_kernel(uint32_t* __restrict__ bk, const uint8_t* __restrict__ s_lv0, const uint8_t* __restrict__ s_ring,
                 int N, int K, int L, int bg_bit, double alpha, uint64_t seed) {
    extern __shared__ uint32_t s_mem[];  // a[N] | s bits as words [N]
    uint32_t* s_a = s_mem;
    uint32_t* s_s = s_mem + N;
    const int K1 = K + 1, R = K1 * L;
    const uint64_t row = blockIdx.x;
    const int i = (int)(row / R), r = (int)(row % R), comp = r / L, level = r % L;
    uint32_t* out = bk + row * (size_t)K1 * N;
    const int t = threadIdx.x, T = blockDim.x;

    // noise into b (two normals per Philox block)
    for (int j = 2 * t; j < N; j += 2 * T) {
        uint32_t w[4];
        philox4(w, seed, kStreamBkNoise, row, (uint32_t)(j >> 1));
        double z0, z1;
        normal2(w, z0, z1);
        out[j] = torus_from_double(z0 * alpha);
        if (j + 1 < N) out[j + 1] = torus_from_double(z1 * alpha);
    }
    for (int k = 0; k < K; k++) {
        __syncthreads();
        for (int j = 4 * t; j < N; j += 4 * T) {
            uint32_t w[4];
            philox4(w, seed, kStreamBkMask, row * (uint64_t)K + k, (uint32_t)(j >> 2));
#pragma unroll
            for (int e = 0; e < 4; e++)
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
// b = e + sum_j a_j s0_j + o * s1_i * 2^(32 - bb (m + 1)).
__global__ void __launch_bounds__(256)
keygen_ks_kernel(uint32_t* __restrict__ ks, const uint8_t* __restrict__ s_lv0, const uint8_t* __restrict__ s_lv1,
                 size_t n_rows, int n, int T, int bb, double alpha, uint64_t seed) {
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
    for (int j0 = 0; j0 < n; j0 += 256) {
        const int j = j0 + 4 * lane;
        uint32_t w[4];
        philox4(w, seed, kStreamKsMask, row, (uint32_t)(j >> 2));
#pragma unroll
        for (int e = 0; e < 4; e++)
            if (j + e < n) {
                out[1 + j + e] = w[e];
                if (s_lv0[j + e]) acc += w[e];
            }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if (lane == 0) {
        uint32_t w[4];
        philox4(w, seed, kStreamKsNoise, row, 0u);
        double z0, z1;
        normal2(w, z0, z1);
        const uint32_t msg = s_lv1[i] ? (uint32_t)o << (32 - (m + 1) * bb) : 0u;
        out[0] = acc + torus_from_double(z0 * alpha) + msg;
    }
}

hipError_t launch_keygen_bk(uint32_t* bk, const uint8_t* s_lv0, const uint8_t* s_ring, int n, int N, int K, int L,
                            int bg_bit, double alpha, uint64_t seed, hipStream_t stream) {
    const size_t rows = (size_t)n * (K + 1) * L;
    if (rows == 0) return hipSuccess;
    const int threads = N >= 256 ? 256 : 64;
    keygen_bk_kernel<<<(unsigned)rows, threads, 2 * (size_t)N * sizeof(uint32_t), stream>>>(bk, s_lv0, s_ring, N, K, L,
                                                                                          bg_bit, alpha, seed);
    return hipGetLastError();
}

hipError_t launch_keygen_ks(uint32_t* ks, const uint8_t* s_lv0, const uint8_t* s_lv1, int n, int nlv1, int T, int bb,
                            double alpha, uint64_t seed, hipStream_t stream) {
    const size_t rows = (size_t)nlv1 * T << bb;
    if (rows == 0) return hipSuccess;
    keygen_ks_kernel<<<(unsigned)((rows + 3) / 4), 256, 0, stream>>>(ks, s_lv0, s_lv1, rows, n, T, bb, alpha, seed);
    return hipGetLastError();
}

}  // namespace tfhe_amd
