// throughput of v_mfma_f64_16x16x4_f64 on one SIMD: independent accumulators, one or two waves on the SIMD (scratch tool)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4_t __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ void k(double* out, long long* cyc, double seed, int active_waves_mask) {
    const int wave = threadIdx.x >> 6;
    if (!((active_waves_mask >> wave) & 1)) return;
    double x = seed + threadIdx.x * 1e-9;
    double4_t d[NACC];
    for (int j = 0; j < NACC; ++j) d[j] = double4_t{x, x, x, x};
    long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int rep = 0; rep < 16; ++rep) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < NACC; ++j) d[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(x, 1e-9, d[j], 0, 0, 0);
    }
    double s = 0; for (int j = 0; j < NACC; ++j) s += d[j][0] + d[j][3];
    // the result is consumed by a vector instruction before the clock is read again
    asm volatile("v_add_f64 %0, %0, %0\n\ts_nop 4" : "+v"(s));
    long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) cyc[wave] = t1 - t0;
    out[threadIdx.x] = s;
}
int main() {
    double* out; long long* cyc; hipMalloc(&out, 1024 * 8); hipMalloc(&cyc, 128);
    long long h[16];
    auto run = [&](auto kern, int nacc, int mask, const char* what) {
        for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(kern, dim3(1), dim3(1024), 0, 0, out, cyc, 1.0, mask);
        hipMemcpy(h, cyc, 128, hipMemcpyDeviceToHost);
        int w0 = 0; while (!((mask >> w0) & 1)) ++w0;
        long long mx = 0; int nw = 0; for (int w = 0; w < 16; ++w) if ((mask >> w) & 1) { mx = h[w] > mx ? h[w] : mx; ++nw; }
        printf("%-50s %7.1f cycles per MFMA (wave %d), slowest wave %7.1f, %d waves -> %.1f MFMA-cycles per CU-cycle\n", what, (double)h[w0] / (16.0 * 8 * nacc), w0, (double)mx / (16.0 * 8 * nacc), nw, nw * 64.0 / ((double)mx / (16.0 * 8 * nacc)));
    };
    run(k<1>, 1, 1, "1 accumulator (dependent), 1 wave");
    run(k<2>, 2, 1, "2 accumulators, 1 wave");
    run(k<4>, 4, 1, "4 accumulators, 1 wave");
    run(k<8>, 8, 1, "8 accumulators, 1 wave");
    run(k<4>, 4, 0x11, "4 accumulators, waves 0 and 4 (same SIMD?)");
    run(k<4>, 4, 0x03, "4 accumulators, waves 0 and 1 (different SIMDs?)");
    run(k<4>, 4, 0xff, "4 accumulators, all 8 waves");
    run(k<1>, 1, 0x11, "1 accumulator, waves 0 and 4");
    run(k<4>, 4, 0x111, "4 accumulators, waves 0, 4, 8");
    run(k<4>, 4, 0x1111, "4 accumulators, waves 0, 4, 8, 12");
    run(k<4>, 4, 0xffff, "4 accumulators, all 16 waves");
    run(k<1>, 1, 0xffff, "1 accumulator, all 16 waves");
    return 0;
}
