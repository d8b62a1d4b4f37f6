// throughput of v_fma_f64 on one SIMD: 1, 2, 4 waves (scratch tool)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(double* out, long long* cyc, const double* in, int mask) {
    const int wave = threadIdx.x >> 6;
    if (!((mask >> wave) & 1)) return;
    double x = in[threadIdx.x & 63], m = in[64 + (threadIdx.x & 63)];
    double y[16];
    for (int j = 0; j < 16; ++j) y[j] = x + j;
    long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int rep = 0; rep < 64; ++rep) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 16; ++j) y[j] = fma(y[j], m, x);
    }
    double s = 0; for (int j = 0; j < 16; ++j) s += y[j];
    asm volatile("v_add_f64 %0, %0, %0\n\ts_nop 4" : "+v"(s));
    long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) cyc[wave] = t1 - t0;
    out[threadIdx.x] = s;
}
int main() {
    double* out; long long* cyc; double* in; hipMalloc(&out, 1024 * 8); hipMalloc(&cyc, 128); hipMalloc(&in, 1024);
    double hin[128]; for (int i = 0; i < 128; ++i) hin[i] = i < 64 ? 1.0 + i * 1e-3 : 0.999; hipMemcpy(in, hin, 1024, hipMemcpyHostToDevice);
    long long h[16];
    auto run = [&](int mask, const char* what) {
        for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k, dim3(1), dim3(1024), 0, 0, out, cyc, in, mask);
        hipMemcpy(h, cyc, 128, hipMemcpyDeviceToHost);
        long long mx = 0; int nw = 0; for (int w = 0; w < 16; ++w) if ((mask >> w) & 1) { mx = h[w] > mx ? h[w] : mx; ++nw; }
        printf("%-40s slowest wave %6.2f cycles per FMA instruction; %d waves -> %.2f cycles of SIMD time per wave-FMA\n", what, (double)mx / (64.0 * 64), nw, (double)mx / (64.0 * 64) / ((nw + 3) / 4));
    };
    run(0x1, "1 wave"); run(0x11, "waves 0, 4 (one SIMD)"); run(0x1111, "waves 0, 4, 8, 12 (one SIMD)"); run(0xf, "waves 0-3 (four SIMDs)"); run(0xffff, "16 waves");
    return 0;
}
