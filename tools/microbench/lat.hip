// latency probe: dependent chains on one wavefront (scratch tool)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4_t __attribute__((ext_vector_type(4)));
__global__ void k(double* out, long long* cyc, double seed) {
    double x = seed + threadIdx.x * 1e-9; double4_t c = {x, x, x, x};
    long long t0, t1;
    // 1. dependent f64 fma chain
    t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int i = 0; i < 256; ++i) x = fma(x, 0.999999, 1e-12);
    asm volatile("" : "+v"(x));
    t1 = __builtin_amdgcn_s_memtime(); if (threadIdx.x == 0) cyc[0] = t1 - t0;
    // 2. independent f64 fma (8 chains)
    double y[8]; for (int j = 0; j < 8; ++j) y[j] = x + j;
    t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int i = 0; i < 32; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) y[j] = fma(y[j], 0.999999, 1e-12);
    for (int j = 0; j < 8; ++j) asm volatile("" : "+v"(y[j]));
    t1 = __builtin_amdgcn_s_memtime(); if (threadIdx.x == 0) cyc[1] = t1 - t0;
    for (int j = 0; j < 8; ++j) x += y[j];
    // 3. dependent mfma chain
    t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int i = 0; i < 64; ++i) c = __builtin_amdgcn_mfma_f64_16x16x4f64(x, 1e-9, c, 0, 0, 0);
    asm volatile("" : "+v"(c));
    t1 = __builtin_amdgcn_s_memtime(); if (threadIdx.x == 0) cyc[2] = t1 - t0;
    // 4. independent mfma (4 accumulators)
    double4_t d[4] = {c, c, c, c};
    t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int i = 0; i < 16; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) d[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(x, 1e-9, d[j], 0, 0, 0);
    for (int j = 0; j < 4; ++j) asm volatile("" : "+v"(d[j]));
    t1 = __builtin_amdgcn_s_memtime(); if (threadIdx.x == 0) cyc[3] = t1 - t0;
    // 5. mfma -> readlane -> fma -> mfma chain
    t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int i = 0; i < 32; ++i) {
        c = __builtin_amdgcn_mfma_f64_16x16x4f64(x, 1e-9, c, 0, 0, 0);
        double v = c[0]; double s = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 3), __builtin_amdgcn_readlane(__double2loint(v), 3));
        x = fma(s, 1e-9, x);
    }
    asm volatile("" : "+v"(c));
    t1 = __builtin_amdgcn_s_memtime(); if (threadIdx.x == 0) cyc[4] = t1 - t0;
    // 6. dependent rcp chain
    t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int i = 0; i < 64; ++i) x = __builtin_amdgcn_rcp(x + 1.5);
    asm volatile("" : "+v"(x));
    t1 = __builtin_amdgcn_s_memtime(); if (threadIdx.x == 0) cyc[5] = t1 - t0;
    // 7. readlane -> fma chain (VALU -> SGPR -> VALU)
    t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int i = 0; i < 64; ++i) { double s = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), 5), __builtin_amdgcn_readlane(__double2loint(x), 5)); x = fma(s, 1e-9, x); }
    asm volatile("" : "+v"(x));
    t1 = __builtin_amdgcn_s_memtime(); if (threadIdx.x == 0) cyc[6] = t1 - t0;
    // 8. 4x4x4 mfma dependent chain
    double e = x;
    t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int i = 0; i < 64; ++i) e = __builtin_amdgcn_mfma_f64_4x4x4f64(x, 1e-9, e, 0, 0, 0);
    asm volatile("" : "+v"(e));
    t1 = __builtin_amdgcn_s_memtime(); if (threadIdx.x == 0) cyc[7] = t1 - t0;
    out[threadIdx.x] = x + c[0] + c[1] + d[0][0] + d[1][1] + d[2][2] + d[3][3] + e;
}
int main() {
    double* out; long long* cyc; hipMalloc(&out, 64 * 8); hipMalloc(&cyc, 64);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, out, cyc, 1.0);
    long long h[8]; hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
    printf("dep fma f64        : %.1f cyc/op\n", h[0] / 256.0);
    printf("indep fma f64      : %.1f cyc/op\n", h[1] / 256.0);
    printf("dep mfma 16x16x4   : %.1f cyc/op\n", h[2] / 64.0);
    printf("indep mfma 16x16x4 : %.1f cyc/op\n", h[3] / 64.0);
    printf("mfma->readlane->fma: %.1f cyc/iter\n", h[4] / 32.0);
    printf("dep rcp(+add)      : %.1f cyc/iter\n", h[5] / 64.0);
    printf("readlane->fma      : %.1f cyc/iter\n", h[6] / 64.0);
    printf("dep mfma 4x4x4     : %.1f cyc/op\n", h[7] / 64.0);
    return 0;
}
