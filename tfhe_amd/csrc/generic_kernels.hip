// Parameter-generic kernels: any ring degree N (power of two), any K, L, gadget base.  Two forms: per-gate
// state in LDS (N <= 2048 and (2 (K+1) + 1 + (K+1) L) N words within 160 KB), or in a global workspace.
// They evaluate the negacyclic products directly in wrap-around 32-bit arithmetic, which is
// the definition itself (what the reference executes for N < 32, src/types.hpp:225-236), so
// they are exact for every parameter set at O(N^2) cost.  They back the lower-level entry
// points (external product, CMUX, blind rotation with an arbitrary test vector) and every
// parameter set the specialised NTT kernels do not cover.
//
//   external product   reference src/trgsw.hpp:100-110
//   CMUX               reference src/trgsw.hpp:112-117
//   blind rotation     reference src/trgsw.hpp:131-170
#include "kernels.hpp"

namespace tfhe_amd {

namespace {

constexpr int kGenThreads = 256;
constexpr int kGenMaxOut = 8;  // outputs per thread: N <= 2048

struct GenShape {
    int N, K1, L, R, bg;
};

// offset that turns the reference's carry chain into plain binary carries (see kernels.hip)
__device__ __forceinline__ uint32_t decomp_offset(int L, int bg) {
    uint32_t off = 1u << (32 - L * bg - 1);
    for (int l = 0; l < L; l++) off += 1u << (32 - bg * (l + 1) + bg - 1);
    return off;
}

// dig[(q*L + l)*N + j] = digit l of src_q[j], for all q, l, j (block-cooperative)
__device__ void decompose_block(int32_t* dig, const uint32_t* src /*[K1][N] in LDS*/, const GenShape& s) {
    const uint32_t off = decomp_offset(s.L, s.bg), mask = (1u << s.bg) - 1;
    const int half = 1 << (s.bg - 1);
    for (int e = threadIdx.x; e < s.K1 * s.N; e += kGenThreads) {
        const int q = e / s.N, j = e - q * s.N;
        const uint32_t t = src[e] + off;
        for (int l = 0; l < s.L; l++)
            dig[(q * s.L + l) * s.N + j] = (int)((t >> (32 - s.bg * (l + 1))) & mask) - half;
    }
}

// out_col[k] (+)= sum_r dig_r (*) C[r][col], for every col; C is [R][K1][N] in global memory.
// stage: LDS buffer of N words.  res: LDS [K1][N], overwritten when !accumulate.
__device__ void extprod_block(uint32_t* res, const int32_t* dig, const uint32_t* __restrict__ C, uint32_t* stage,
                              const GenShape& s, bool accumulate) {
    const int N = s.N;
    for (int col = 0; col < s.K1; col++) {
        uint32_t sum[kGenMaxOut];
#pragma unroll
        for (int t = 0; t < kGenMaxOut; t++) sum[t] = 0u;
        for (int r = 0; r < s.R; r++) {
            __syncthreads();
            const uint32_t* poly = C + ((size_t)r * s.K1 + col) * N;
            for (int j = threadIdx.x; j < N; j += kGenThreads) stage[j] = poly[j];
            __syncthreads();
            const int32_t* d = dig + r * N;
            for (int j = 0; j < N; j++) {
                const uint32_t dj = (uint32_t)d[j];
#pragma unroll
                for (int t = 0; t < kGenMaxOut; t++) {
                    const int k = threadIdx.x + kGenThreads * t;
                    if (k < N) {
                        const int idx = k - j;  // dig[j] * poly[k - j], negated when it wraps
                        const uint32_t v = stage[idx & (N - 1)] * dj;
                        sum[t] += idx < 0 ? 0u - v : v;
                    }
                }
            }
        }
#pragma unroll
        for (int t = 0; t < kGenMaxOut; t++) {
            const int k = threadIdx.x + kGenThreads * t;
            if (k < N) res[col * N + k] = accumulate ? res[col * N + k] + sum[t] : sum[t];
        }
    }
    __syncthreads();
}

}  // namespace

// one workgroup per item; mode 0: out = C (x) x;  mode 1 (CMUX): out = C (x) (d1 - d0) + d0
__global__ void __launch_bounds__(kGenThreads)
external_product_generic_kernel(const uint32_t* __restrict__ trgsw, const uint32_t* __restrict__ x0,
                                const uint32_t* __restrict__ x1, uint32_t* __restrict__ out, GenShape s, int mode) {
    extern __shared__ uint32_t s_mem[];
    const int W = s.K1 * s.N;
    uint32_t* in = s_mem;                                       // [K1][N]
    uint32_t* res = in + W;                                     // [K1][N]
    uint32_t* stage = res + W;                                  // [N]
    int32_t* dig = reinterpret_cast<int32_t*>(stage + s.N);     // [R][N]
    const size_t item = blockIdx.x;
    for (int e = threadIdx.x; e < W; e += kGenThreads) {
        const uint32_t a = x0[item * W + e];
        in[e] = mode ? x1[item * W + e] - a : a;
        res[e] = mode ? a : 0u;
    }
    __syncthreads();
    decompose_block(dig, in, s);
    extprod_block(res, dig, trgsw + item * (size_t)s.R * W, stage, s, true);
    for (int e = threadIdx.x; e < W; e += kGenThreads) out[item * W + e] = res[e];
}

// one workgroup per gate.  tv: test vector [K1][N] shared by all gates, or nullptr for the
// gate-bootstrapping one (1/8 in every coefficient of the b polynomial).  extract != 0: write
// sample_extract_index(acc, 0) as [N*K + 1] words per gate, else the accumulator [K1][N].
__global__ void __launch_bounds__(kGenThreads)
blind_rotate_generic_kernel(const uint32_t* __restrict__ bk, const uint32_t* __restrict__ c,
                            const uint32_t* __restrict__ tv, uint32_t* __restrict__ out, GenShape s, int n,
                            int extract) {
    extern __shared__ uint32_t s_mem[];
    const int N = s.N, W = s.K1 * N;
    uint32_t* acc = s_mem;                                      // [K1][N]
    uint32_t* diff = acc + W;                                   // [K1][N]
    uint32_t* stage = diff + W;                                 // [N]
    int32_t* dig = reinterpret_cast<int32_t*>(stage + N);       // [R][N]
    const size_t gate = blockIdx.x;
    const uint32_t* cg = c + gate * (size_t)(n + 1);
    int log2n = 0;
    while ((1 << log2n) < N) log2n++;
    const int sh = 32 - (log2n + 1);

    // acc = tv * X^(-floor(b * 2N))  (reference src/trgsw.hpp:155-158, src/types.hpp:299-312)
    const int bbar = (int)(cg[0] >> sh);
    for (int e = threadIdx.x; e < W; e += kGenThreads) {
        const int q = e / N, j = e - q * N;
        const int src = (j + bbar) & (2 * N - 1);
        const uint32_t v = tv ? tv[q * N + (src & (N - 1))] : (q == 0 ? (1u << 29) : 0u);
        acc[e] = (src & N) ? 0u - v : v;
    }
    __syncthreads();
    for (int i = 0; i < n; i++) {
        const int abar = (int)((cg[1 + i] + (1u << (sh - 1))) >> sh);
        for (int e = threadIdx.x; e < W; e += kGenThreads) {
            const int q = e / N, j = e - q * N;
            const int src = (j - abar) & (2 * N - 1);
            const uint32_t v = acc[q * N + (src & (N - 1))];
            diff[e] = ((src & N) ? 0u - v : v) - acc[e];
        }
        __syncthreads();
        decompose_block(dig, diff, s);
        extprod_block(acc, dig, bk + (size_t)i * s.R * W, stage, s, true);
    }
    if (extract) {
        uint32_t* o = out + gate * (size_t)((s.K1 - 1) * N + 1);
        for (int e = threadIdx.x; e < (s.K1 - 1) * N; e += kGenThreads) {
            const int q = e / N, j = e - q * N;
            const uint32_t v = acc[(1 + q) * N + j];
            if (j == 0) o[1 + q * N] = v;
            else        o[1 + q * N + (N - j)] = 0u - v;
        }
        if (threadIdx.x == 0) o[0] = acc[0];
    } else {
        for (int e = threadIdx.x; e < W; e += kGenThreads) out[gate * (size_t)W + e] = acc[e];
    }
}

// ------------------------------------------------------------------------------------------
// The same algorithm for rings whose per-gate state does not fit LDS (N > 2048, or many mask
// polynomials / levels): accumulator, difference and digits live in a global workspace of
// generic_big_workspace_words() words per item; one workgroup of 1024 threads per item.  A product
// stages one digit polynomial and one key polynomial in LDS while both fit (N <= 16384) and reads
// them from global memory beyond that.
// ------------------------------------------------------------------------------------------
constexpr int kBigThreads = 1024;
constexpr int kBigLdsN = 16384;  // 2 x 64 KB

__device__ void decompose_big(int32_t* dig, const uint32_t* src, const GenShape& s) {
    const uint32_t off = decomp_offset(s.L, s.bg), mask = (1u << s.bg) - 1;
    const int half = 1 << (s.bg - 1);
    for (int e = threadIdx.x; e < s.K1 * s.N; e += kBigThreads) {
        const int q = e / s.N, j = e - q * s.N;
        const uint32_t t = src[e] + off;
        for (int l = 0; l < s.L; l++)
            dig[(size_t)(q * s.L + l) * s.N + j] = (int)((t >> (32 - s.bg * (l + 1))) & mask) - half;
    }
    __syncthreads();
}

// res_col += sum_r dig_r (*) C[r][col]; dig [R][N] and res [K1][N] in global memory, lds: 2 N words or null
__device__ void extprod_big(uint32_t* res, const int32_t* dig, const uint32_t* __restrict__ C, uint32_t* lds,
                            const GenShape& s) {
    const int N = s.N;
    for (int col = 0; col < s.K1; col++)
        for (int r = 0; r < s.R; r++) {
            const uint32_t* poly = C + ((size_t)r * s.K1 + col) * N;
            const int32_t* d = dig + (size_t)r * N;
            if (lds) {
                __syncthreads();
                for (int j = threadIdx.x; j < N; j += kBigThreads) {
                    lds[j] = poly[j];
                    lds[N + j] = (uint32_t)d[j];
                }
                __syncthreads();
                poly = lds;
                d = reinterpret_cast<const int32_t*>(lds + N);
            }
            for (int k = threadIdx.x; k < N; k += kBigThreads) {
                uint32_t sum = 0u;
                for (int j = 0; j <= k; j++) sum += poly[k - j] * (uint32_t)d[j];
                for (int j = k + 1; j < N; j++) sum -= poly[k - j + N] * (uint32_t)d[j];
                res[(size_t)col * N + k] += sum;
            }
        }
    __syncthreads();
}

__global__ void __launch_bounds__(kBigThreads)
external_product_generic_big_kernel(const uint32_t* __restrict__ trgsw, const uint32_t* __restrict__ x0,
                                    const uint32_t* __restrict__ x1, uint32_t* __restrict__ out, uint32_t* __restrict__ ws,
                                    GenShape s, int mode, int use_lds) {
    extern __shared__ uint32_t s_mem[];
    const int W = s.K1 * s.N;
    const size_t item = blockIdx.x;
    uint32_t* in = ws + item * (size_t)(2 * W + s.R * s.N);
    uint32_t* res = out + item * (size_t)W;
    int32_t* dig = reinterpret_cast<int32_t*>(in + 2 * W);
    for (int e = threadIdx.x; e < W; e += kBigThreads) {
        const uint32_t a = x0[item * W + e];
        in[e] = mode ? x1[item * W + e] - a : a;
        res[e] = mode ? a : 0u;
    }
    __syncthreads();
    decompose_big(dig, in, s);
    extprod_big(res, dig, trgsw + item * (size_t)s.R * W, use_lds ? s_mem : nullptr, s);
}

__global__ void __launch_bounds__(kBigThreads)
blind_rotate_generic_big_kernel(const uint32_t* __restrict__ bk, const uint32_t* __restrict__ c,
                                const uint32_t* __restrict__ tv, uint32_t* __restrict__ out, uint32_t* __restrict__ ws,
                                GenShape s, int n, int extract, int use_lds) {
    extern __shared__ uint32_t s_mem[];
    const int N = s.N, W = s.K1 * N;
    const size_t gate = blockIdx.x;
    uint32_t* acc = ws + gate * (size_t)(2 * W + s.R * N);
    uint32_t* diff = acc + W;
    int32_t* dig = reinterpret_cast<int32_t*>(diff + W);
    const uint32_t* cg = c + gate * (size_t)(n + 1);
    int log2n = 0;
    while ((1 << log2n) < N) log2n++;
    const int sh = 32 - (log2n + 1);
    const int bbar = (int)(cg[0] >> sh);
    for (int e = threadIdx.x; e < W; e += kBigThreads) {
        const int q = e / N, j = e - q * N;
        const int src = (j + bbar) & (2 * N - 1);
        const uint32_t v = tv ? tv[q * N + (src & (N - 1))] : (q == 0 ? (1u << 29) : 0u);
        acc[e] = (src & N) ? 0u - v : v;
    }
    __syncthreads();
    for (int i = 0; i < n; i++) {
        const int abar = (int)((cg[1 + i] + (1u << (sh - 1))) >> sh);
        for (int e = threadIdx.x; e < W; e += kBigThreads) {
            const int q = e / N, j = e - q * N;
            const int src = (j - abar) & (2 * N - 1);
            const uint32_t v = acc[q * N + (src & (N - 1))];
            diff[e] = ((src & N) ? 0u - v : v) - acc[e];
        }
        __syncthreads();
        decompose_big(dig, diff, s);
        extprod_big(acc, dig, bk + (size_t)i * s.R * W, use_lds ? s_mem : nullptr, s);
    }
    if (extract) {
        uint32_t* o = out + gate * (size_t)((s.K1 - 1) * N + 1);
        for (int e = threadIdx.x; e < (s.K1 - 1) * N; e += kBigThreads) {
            const int q = e / N, j = e - q * N;
            const uint32_t v = acc[(1 + q) * N + j];
            if (j == 0) o[1 + q * N] = v;
            else        o[1 + q * N + (N - j)] = 0u - v;
        }
        if (threadIdx.x == 0) o[0] = acc[0];
    } else {
        for (int e = threadIdx.x; e < W; e += kBigThreads) out[gate * (size_t)W + e] = acc[e];
    }
}

static size_t generic_lds_bytes(int N, int K1, int R) { return sizeof(uint32_t) * ((size_t)2 * K1 * N + N + (size_t)R * N); }

static bool generic_fits_lds(int N, int K, int L) {
    return N <= kGenThreads * kGenMaxOut && generic_lds_bytes(N, K + 1, (K + 1) * L) <= 160 * 1024;
}

// every ring the reference accepts (a power of two, src/trgsw.hpp:142-143) with any K and L; rings
// above 2^26 would overflow the 32-bit indices of these kernels
bool generic_has_kernel(int N, int K, int L) {
    return N >= 2 && N <= (1 << 26) && !(N & (N - 1)) && K >= 1 && L >= 1;
}

size_t generic_big_workspace_words(size_t items, int N, int K, int L) {
    if (generic_fits_lds(N, K, L)) return 0;
    return items * ((size_t)2 * (K + 1) * N + (size_t)(K + 1) * L * N);
}

hipError_t launch_external_product_generic(const uint32_t* trgsw, const uint32_t* x0, const uint32_t* x1,
                                           uint32_t* out, size_t batch, int N, int K, int L, int bg, int mode,
                                           uint32_t* ws, hipStream_t stream) {
    if (batch == 0) return hipSuccess;
    if (!generic_has_kernel(N, K, L)) return hipErrorInvalidValue;
    GenShape s{N, K + 1, L, (K + 1) * L, bg};
    if (!generic_fits_lds(N, K, L)) {
        if (!ws) return hipErrorInvalidValue;
        const int use_lds = N <= kBigLdsN;
        const size_t big_lds = use_lds ? sizeof(uint32_t) * 2 * (size_t)N : 0;
        hipError_t e = hipFuncSetAttribute((const void*)external_product_generic_big_kernel,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)big_lds);
        if (e != hipSuccess) return e;
        external_product_generic_big_kernel<<<(unsigned)batch, kBigThreads, big_lds, stream>>>(trgsw, x0, x1, out, ws, s,
                                                                                             mode, use_lds);
        return hipGetLastError();
    }
    const size_t lds = generic_lds_bytes(N, s.K1, s.R);
    hipError_t e = hipFuncSetAttribute((const void*)external_product_generic_kernel,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    external_product_generic_kernel<<<(unsigned)batch, kGenThreads, lds, stream>>>(trgsw, x0, x1, out, s, mode);
    return hipGetLastError();
}

hipError_t launch_blind_rotate_generic(const uint32_t* bk, const uint32_t* c, const uint32_t* tv, uint32_t* out,
                                       size_t n_gates, int n, int N, int K, int L, int bg, int extract,
                                       uint32_t* ws, hipStream_t stream) {
    if (n_gates == 0) return hipSuccess;
    if (!generic_has_kernel(N, K, L)) return hipErrorInvalidValue;
    GenShape s{N, K + 1, L, (K + 1) * L, bg};
    if (!generic_fits_lds(N, K, L)) {
        if (!ws) return hipErrorInvalidValue;
        const int use_lds = N <= kBigLdsN;
        const size_t big_lds = use_lds ? sizeof(uint32_t) * 2 * (size_t)N : 0;
        hipError_t e = hipFuncSetAttribute((const void*)blind_rotate_generic_big_kernel,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)big_lds);
        if (e != hipSuccess) return e;
        blind_rotate_generic_big_kernel<<<(unsigned)n_gates, kBigThreads, big_lds, stream>>>(bk, c, tv, out, ws, s, n,
                                                                                            extract, use_lds);
        return hipGetLastError();
    }
    const size_t lds = generic_lds_bytes(N, s.K1, s.R);
    hipError_t e = hipFuncSetAttribute((const void*)blind_rotate_generic_kernel,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    blind_rotate_generic_kernel<<<(unsigned)n_gates, kGenThreads, lds, stream>>>(bk, c, tv, out, s, n, extract);
    return hipGetLastError();
}

}  // namespace tfhe_amd
