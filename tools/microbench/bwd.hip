// standalone timing of one backward block step (scratch tool)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4_t __attribute__((ext_vector_type(4)));
constexpr int NBW = 5, SLOT = (NBW + 1) * 256 + 128;
template <int VAR>
__global__ void k(const double* src, double* out, long long* cyc) {
    __shared__ double ring[2 * SLOT];
    const int lane = threadIdx.x & 63, li = lane & 15, lk = lane >> 4;
    for (int i = lane; i < 2 * SLOT; i += 64) ring[i] = src[i % 997] * 1e-3;
    __syncthreads();
    double4_t xk[5];
    for (int K = 0; K < 5; ++K) xk[K] = double4_t{1.0 + K, 2.0, 3.0, 4.0};
    double pre[NBW][4];
    for (int K = 0; K < NBW; ++K) for (int m = 0; m < 4; ++m) pre[K][m] = ring[(K + 1) * 256 + (4 * m + lk) * 16 + li];
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int J = 63; J >= 0; --J) {
        const double* B0 = ring + (J & 1) * SLOT;
        double4_t acc[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) acc[m] = double4_t{0, 0, 0, 0};
        if (VAR != 2) {
#pragma unroll
        for (int K = NBW; K >= 1; --K) {
            const double* T = B0 + K * 256;
#pragma unroll
            for (int m = 0; m < 4; ++m) acc[m] = __builtin_amdgcn_mfma_f64_16x16x4f64(VAR == 3 ? pre[K - 1][m] : T[(4 * m + lk) * 16 + li], xk[K - 1][m], acc[m], 0, 0, 0);
        }
        }
        double4_t t;
#pragma unroll
        for (int r = 0; r < 4; ++r) t[r] = B0[(NBW + 1) * 256 + lk + 4 * r] - ((acc[0][r] + acc[1][r]) + (acc[2][r] + acc[3][r]));
        double4_t xn;
        if (VAR != 1 && VAR != 3) {
        double4_t xp[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) xp[m] = __builtin_amdgcn_mfma_f64_16x16x4f64(B0[(4 * m + lk) * 16 + li], t[m], double4_t{0, 0, 0, 0}, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) xn[r] = (xp[0][r] + xp[1][r]) + (xp[2][r] + xp[3][r]);
        } else xn = t;
        if (li == 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) out[16 * J + lk + 4 * r] = xn[r];
        }
#pragma unroll
        for (int K = 4; K >= 1; --K) xk[K] = xk[K - 1];
        xk[0] = xn;
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) cyc[0] = (t1 - t0) / 64;
}
int main() {
    double* src; double* out; long long* cyc; hipMalloc(&src, 8192); hipMalloc(&out, 64 * 16 * 8); hipMalloc(&cyc, 64);
    double h[1024]; for (int i = 0; i < 1024; ++i) h[i] = (i % 17) * 0.01; hipMemcpy(src, h, 8192, hipMemcpyHostToDevice);
    long long c;
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k<0>, dim3(1), dim3(64), 0, 0, src, out, cyc);
    hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost); printf("full block step      : %lld cycles\n", c);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k<1>, dim3(1), dim3(64), 0, 0, src, out, cyc);
    hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost); printf("tile products only   : %lld cycles\n", c);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k<2>, dim3(1), dim3(64), 0, 0, src, out, cyc);
    hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost); printf("inv(L)' product only : %lld cycles\n", c);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k<3>, dim3(1), dim3(64), 0, 0, src, out, cyc);
    hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost); printf("tile products, A in registers: %lld cycles\n", c);
    return 0;
}
