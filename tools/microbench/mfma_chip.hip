// sustained v_mfma_f64_16x16x4_f64 / v_fma_f64 rate of the WHOLE chip in wall-clock time (the clock under load, not cycles): grid x 256
// threads, 16 independent accumulators per wave, nothing else in the loop; wave 0 of workgroup 0 also reads the shader clock counter
// (s_memtime) against the constant 100 MHz one (s_memrealtime)  (scratch tool)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4_t __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(256) void k(double* out, long long* clk, int iters, double seed) {
    double x = seed + threadIdx.x * 1e-9;
    double4_t d[16];
    for (int j = 0; j < 16; ++j) d[j] = double4_t{x, x, x, x};
    const long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
        if constexpr (MODE == 0) {
#pragma unroll
            for (int j = 0; j < 16; ++j) d[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(x, 1e-9, d[j], 0, 0, 0);
        } else if constexpr (MODE == 2) {           // the same with the accumulators forced into VGPRs (the compiler picks AGPRs above)
            const double y = 1e-9;
#pragma unroll
            for (int j = 0; j < 16; ++j) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(d[j]) : "v"(x), "v"(y));
        } else {
#pragma unroll
            for (int j = 0; j < 16; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) d[j][r] = __builtin_fma(d[j][r], x, 1e-9);
        }
    }
    double s = 0; for (int j = 0; j < 16; ++j) s += d[j][0] + d[j][3] + d[j][1] + d[j][2];
    asm volatile("v_add_f64 %0, %0, %0\n\ts_nop 4" : "+v"(s));
    const long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = c1 - c0; clk[1] = r1 - r0; }
    if (s == 12345.678) out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
    double* out; long long* clk; (void)hipMalloc(&out, 8 * 256 * 4096); (void)hipMalloc(&clk, 16);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int mode = 0; mode < 3; ++mode) for (int grid : {1, 256, 512, 2048}) for (int iters : {20000}) {
        auto launch = [&]() { if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(256), 0, 0, out, clk, iters, 1.0); else if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(grid), dim3(256), 0, 0, out, clk, iters, 1.0); else hipLaunchKernelGGL(k<1>, dim3(grid), dim3(256), 0, 0, out, clk, iters, 1.0); };
        launch();
        (void)hipEventRecord(e0); launch(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        long long h[2]; (void)hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
        const double flops = mode != 1 ? (double)grid * 4 * iters * 16 * 2048.0 : (double)grid * 256 * iters * 64 * 2.0;
        printf("%s grid %5d x 4 waves, %6d x 16 per wave: %8.3f ms  %6.2f TFLOP/s   s_memtime %lld / s_memrealtime %lld -> %.0f MHz, %.1f s_memtime ticks per %s\n", mode == 0 ? "mfma_f64_16x16x4 (AGPR acc)" : mode == 2 ? "mfma_f64_16x16x4 (VGPR acc)" : "v_fma_f64                  ", grid, iters, ms, flops / ms * 1e-9,
               h[0], h[1], (double)h[0] / (double)h[1] * 100.0, (double)h[0] / ((double)iters * 16 * (mode == 1 ? 4 : 1)), mode == 1 ? "fma" : "mfma");
    }
    return 0;
}
