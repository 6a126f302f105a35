// Dependent-issue latency of FP64 vector instructions on gfx950: cycles per instruction of one
// wave running C independent dependency chains (C = 1: every instruction waits for the one
// before it).  Stand-alone diagnostic: not linked into the library.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
    fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1);} } while (0)

constexpr int ITER = 256;

template <int C, int KIND>
__global__ void __launch_bounds__(64) k(unsigned long long* out, double seed) {
    double d[8];
    for (int i = 0; i < 8; i++) d[i] = seed + i + threadIdx.x;
    double c0 = seed * 0.5, c1 = seed * 0.25;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int rep = 0; rep < 24 / C; rep++) {
#pragma unroll
            for (int i = 0; i < C; i++) {
                if (KIND == 0) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[i]) : "v"(c0), "v"(c1));
                if (KIND == 1) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[i]) : "v"(c0));
                if (KIND == 2) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(c0));
                if (KIND == 3) asm volatile("v_rndne_f64 %0, %0" : "+v"(d[i]));
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    double s = 0;
    for (int i = 0; i < 8; i++) s += d[i];
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = (unsigned long long)s; }
}

template <int C, int KIND>
void run(const char* name, unsigned long long* dout) {
    unsigned long long h[2];
    for (int r = 0; r < 2; r++) {
        k<C, KIND><<<1, 64>>>(dout, 1.0);
        CHECK(hipDeviceSynchronize());
    }
    CHECK(hipMemcpy(h, dout, sizeof(h), hipMemcpyDeviceToHost));
    const double n = (double)ITER * (24 / C) * C;
    printf("%-12s chains=%d  %.2f memtime-ticks/instr\n", name, C, h[0] / n);
}

int main() {
    unsigned long long* dout;
    CHECK(hipMalloc(&dout, 64));
#define ALL(KIND, NAME) run<1, KIND>(NAME, dout); run<2, KIND>(NAME, dout); run<3, KIND>(NAME, dout); \
    run<4, KIND>(NAME, dout); run<6, KIND>(NAME, dout); run<8, KIND>(NAME, dout);
    ALL(0, "v_fma_f64") ALL(1, "v_mul_f64") ALL(2, "v_add_f64") ALL(3, "v_rndne_f64")
    return 0;
}
