// Device check of the verification mode's double -> torus conversion (fft256_wave.hpp) against the
// reference's formula (reference src/types.hpp:28-33) evaluated on the host:
//     d = fmod(d, 1.0); if (d < 0.0) d += 1.0; d *= 2^32; (uint32_t)(unsigned long long)d
// Inputs: random values over the range the kernel sees (|d| < 2^17), and the neighbourhoods where a
// shortcut could differ: negative values just below a multiple of 2^-32 (the rounded 1 + d reaches the
// next multiple), -0, +-1/2, values around integers.  Prints "convert_check ok <n>" or the first mismatches.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <random>
#include <vector>

#include <hip/hip_runtime.h>

#include "../../tfhe_amd/csrc/fft256_wave.hpp"

using namespace tfhe_amd;

__global__ void convert_kernel(const double* in, uint32_t* out, int n16) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;  // one group of 16 values per thread
    if (t >= n16) return;
    double a[8], b[8];
    for (int i = 0; i < 8; i++) {
        a[i] = in[(size_t)t * 16 + i];
        b[i] = in[(size_t)t * 16 + 8 + i];
    }
    uint32_t oa[8], ob[8];
    torus_from_double_16(oa, ob, a, b);
    for (int i = 0; i < 8; i++) {
        out[(size_t)t * 16 + i] = oa[i];
        out[(size_t)t * 16 + 8 + i] = ob[i];
    }
}

static uint32_t reference_convert(double d) {
    d = std::fmod(d, 1.0);
    if (d < 0.0) d += 1.0;
    d *= 4294967296.0;
    return (uint32_t)(unsigned long long)d;
}

int main() {
    std::vector<double> v;
    std::mt19937_64 rng(12345);
    std::uniform_real_distribution<double> wide(-131072.0, 131072.0), unit(-1.0, 1.0);
    for (int i = 0; i < 1 << 20; i++) v.push_back(wide(rng));
    for (int i = 0; i < 1 << 18; i++) v.push_back(unit(rng));
    for (int i = 0; i < 1 << 16; i++) v.push_back(std::ldexp(unit(rng), -(int)(rng() % 80)));  // tiny magnitudes
    const double specials[] = {0.0, -0.0, 0.5, -0.5, 1.0, -1.0, 0.25, -0.25, 65536.0, -65536.0, 65535.5, -65535.5};
    for (double s : specials) v.push_back(s);
    // negative values within a few units of 2^-54 .. 2^-70 of a multiple of 2^-32, on both sides
    for (int k = 0; k < 4096; k++) {
        const double m = std::ldexp((double)(rng() % 0xFFFFFFFFull), -32);  // in [0, 1)
        for (int e = 52; e <= 75; e++)
            for (int sgn = -1; sgn <= 1; sgn += 2) {
                v.push_back(-m + sgn * std::ldexp(1.0, -e));
                v.push_back(-m + sgn * std::ldexp(1.5, -e));
                v.push_back(-(double)(rng() % 1000) - m + sgn * std::ldexp(1.0, -e));
            }
    }
    for (int e = 30; e <= 1074; e += 7) { v.push_back(-std::ldexp(1.0, -e)); v.push_back(std::ldexp(1.0, -e)); }
    while (v.size() % (16 * 64)) v.push_back(0.0);
    // values outside (-1/2, 0) sprinkled between the others, so that waves take both paths
    std::shuffle(v.begin(), v.begin() + (1 << 20) + (1 << 18), rng);

    const size_t n = v.size();
    double* d_in;
    uint32_t* d_out;
    if (hipMalloc(&d_in, n * sizeof(double)) != hipSuccess || hipMalloc(&d_out, n * sizeof(uint32_t)) != hipSuccess) return 2;
    hipMemcpy(d_in, v.data(), n * sizeof(double), hipMemcpyHostToDevice);
    const int n16 = (int)(n / 16);
    convert_kernel<<<(n16 + 63) / 64, 64>>>(d_in, d_out, n16);
    std::vector<uint32_t> got(n);
    if (hipMemcpy(got.data(), d_out, n * sizeof(uint32_t), hipMemcpyDeviceToHost) != hipSuccess) return 2;
    int bad = 0;
    for (size_t i = 0; i < n; i++) {
        const uint32_t want = reference_convert(v[i]);
        if (got[i] != want && bad++ < 10) std::printf("mismatch at %zu: d = %a, got %08x, want %08x\n", i, v[i], got[i], want);
    }
    if (bad) { std::printf("convert_check FAILED: %d of %zu\n", bad, n); return 1; }
    std::printf("convert_check ok %zu\n", n);
    return 0;
}
