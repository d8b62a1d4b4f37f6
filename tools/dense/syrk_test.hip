// Stand-alone timing of the dense trailing update (syrk_update128_kernel of nlls_solve.hip) and of variants of it: one pass over a
// 6000-dof trailing matrix, checked against the library's kernel.  (scratch tool)
// build: hipcc -O3 -std=c++20 --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form -o tools/dense/syrk_test.out tools/dense/syrk_test.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef double double4_t __attribute__((ext_vector_type(4)));
constexpr int NB = 64;
constexpr int S128_KC = 16, S128_LD = 144;

// VAR 0: the library's kernel.  1: no read-modify-write (plain store).  2: no operand loads in the loop (first chunk reused).  3: both.
__device__ long long g_stamps[64];
#define STAMP(i) do { if (threadIdx.x == 0 && (blockIdx.x == 0 || blockIdx.x == gridDim.x - 1)) g_stamps[(blockIdx.x ? 8 : 0) + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
template <int VAR>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void syrk128(double* __restrict__ S, const double* __restrict__ W0, const double* __restrict__ W1, int npad, int k0, int jb0) {
    __shared__ double As[2][S128_KC * S128_LD], Bs[2][S128_KC * S128_LD];
    STAMP(0);
    int ti, tj;
    { const int tix = blockIdx.x; ti = (int)((sqrt(8.0 * tix + 1.0) - 1.0) * 0.5); while (ti * (ti + 1) / 2 > tix) --ti; while ((ti + 1) * (ti + 2) / 2 <= tix) ++ti; tj = tix - ti * (ti + 1) / 2; }
    const int I0 = jb0 * NB + 128 * ti, J0 = jb0 * NB + 128 * tj;
    const int t = threadIdx.x, w = __builtin_amdgcn_readfirstlane(t >> 6), lane = t & 63, li = lane & 15, lk = lane >> 4;
    const int r0w = (w & 1) * 64, c0w = (w >> 1) * 64;
    const bool active = I0 + r0w < npad && J0 + c0w < npad && !(ti == tj && c0w > r0w);
    double4_t acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b2 = 0; b2 < 4; ++b2) acc[a][b2] = double4_t{0, 0, 0, 0};
    const int cr = t & 127, kq = t >> 7;
    const int arow = I0 + cr < npad ? I0 + cr : npad - 1, brow = J0 + cr < npad ? J0 + cr : npad - 1;
    constexpr int NCP = S128_KC / 2, CPP = NB / S128_KC;
    double ra[NCP], rb[NCP];
    auto gload = [&](int chunk) {
        const int q = chunk / CPP, col0 = (chunk % CPP) * S128_KC;
        const double* Wq = q == 0 ? W0 : W1;
        const double* Ga = Wq + (size_t)arow + (size_t)npad * col0;
        const double* Gb = S + (size_t)brow + (size_t)npad * ((size_t)(k0 + q) * NB + col0);
#pragma unroll
        for (int i = 0; i < NCP; ++i) { ra[i] = Ga[(size_t)npad * (kq + 2 * i)]; rb[i] = Gb[(size_t)npad * (kq + 2 * i)]; }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NCP; ++i) { As[buf][(kq + 2 * i) * S128_LD + cr] = ra[i]; Bs[buf][(kq + 2 * i) * S128_LD + cr] = rb[i]; }
    };
    constexpr int NCH = 2 * NB / S128_KC;
    gload(0); lstore(0);
    __syncthreads();
    STAMP(1);
#pragma unroll 1
    for (int ch = 0; ch < NCH; ++ch) {
        const int buf = (VAR & 2) ? 0 : (ch & 1);
        if (!(VAR & 2) && ch + 1 < NCH) gload(ch + 1);
        if (active) {
#pragma unroll
            for (int kk = 0; kk < S128_KC; kk += 4) {
                double av[4], bv[4];
#pragma unroll
                for (int a = 0; a < 4; ++a) av[a] = As[buf][(kk + lk) * S128_LD + r0w + 16 * a + li];
#pragma unroll
                for (int b2 = 0; b2 < 4; ++b2) bv[b2] = Bs[buf][(kk + lk) * S128_LD + c0w + 16 * b2 + li];
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int b2 = 0; b2 < 4; ++b2) acc[a][b2] = __builtin_amdgcn_mfma_f64_16x16x4f64(bv[b2], av[a], acc[a][b2], 0, 0, 0);
            }
        }
        if (!(VAR & 2) && ch + 1 < NCH) lstore(buf ^ 1);
        __syncthreads();
    }
    STAMP(2);
    if (!active) return;
    double* Cg = S + (size_t)(I0 + r0w) + (size_t)npad * (J0 + c0w);
#pragma unroll
    for (int b2 = 0; b2 < 4; ++b2) {
        double cold[4][4];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int r = 0; r < 4; ++r) cold[a][r] = (VAR & 1) ? 1.0 : Cg[(size_t)(16 * a + li) + (size_t)npad * (16 * b2 + lk + 4 * r)];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int r = 0; r < 4; ++r) Cg[(size_t)(16 * a + li) + (size_t)npad * (16 * b2 + lk + 4 * r)] = cold[a][r] - acc[a][b2][r];
    }
    __builtin_amdgcn_s_waitcnt(0);      // (for the stamp: the stores have left)
    STAMP(3);
}
#include "syrk_variants.hpp"

int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 6000, npad = ((n + 1 + 63) / 64) * 64, nblk = npad / 64;
    const int k0 = 0, jb0 = 2, T = nblk - 2, T128 = (T + 1) / 2, ntiles = T128 * (T128 + 1) / 2;
    std::vector<double> hS((size_t)npad * npad), hW((size_t)npad * 128);
    srand(1); for (auto& v : hS) v = (rand() % 2001 - 1000) * 1e-3; for (auto& v : hW) v = (rand() % 2001 - 1000) * 1e-3;
    double *S, *S0, *Sref, *W; (void)hipMalloc(&S, hS.size() * 8); (void)hipMalloc(&S0, hS.size() * 8); (void)hipMalloc(&Sref, hS.size() * 8); (void)hipMalloc(&W, hW.size() * 8);
    (void)hipMemcpy(S0, hS.data(), hS.size() * 8, hipMemcpyHostToDevice); (void)hipMemcpy(W, hW.data(), hW.size() * 8, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const double flops = 0; (void)flops;
    double useful = 0; for (int ti = 0; ti < T128; ++ti) for (int tj = 0; tj <= ti; ++tj) useful += (ti == tj ? 0.75 : 1.0) * 2.0 * 128 * 128 * 128;
    auto run = [&](const char* what, auto launch, bool check) {
        (void)hipMemcpy(S, S0, hS.size() * 8, hipMemcpyDeviceToDevice);
        launch();
        double err = -1;
        if (check) {
            std::vector<double> a(hS.size()), b(hS.size()); (void)hipMemcpy(a.data(), S, hS.size() * 8, hipMemcpyDeviceToHost); (void)hipMemcpy(b.data(), Sref, hS.size() * 8, hipMemcpyDeviceToHost);
            err = 0; for (size_t i = 0; i < a.size(); ++i) { const size_t r = i % npad, c = i / npad; if (r >= c && r < (size_t)n) err = fmax(err, fabs(a[i] - b[i])); }   // lower triangle, real rows
        }
        for (int i = 0; i < 3; ++i) launch();
        (void)hipEventRecord(e0); for (int i = 0; i < 20; ++i) launch(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        long long h[16]; (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_stamps), 128);
        printf("%-60s %7.1f us per pass  %5.1f TFLOP/s useful  max |diff| %.2e   first WG: prologue %lld loop %lld epilogue %lld cycles; last WG: %lld %lld %lld, starts %lld after the first\n", what, 1e3 * ms / 20, useful / (ms / 20) * 1e-9, err,
               h[1] - h[0], h[2] - h[1], h[3] - h[2], h[9] - h[8], h[10] - h[9], h[11] - h[10], h[8] - h[0]);
    };
    printf("n %d npad %d: %d tiles of 128 x 128, K = 128\n", n, npad, ntiles);
    (void)hipMemcpy(Sref, S0, hS.size() * 8, hipMemcpyDeviceToDevice);
    hipLaunchKernelGGL(syrk128<0>, dim3(ntiles), dim3(256), 0, 0, Sref, W, W + (size_t)npad * 64, npad, k0, jb0);
    run("0 library kernel", [&]() { hipLaunchKernelGGL(syrk128<0>, dim3(ntiles), dim3(256), 0, 0, S, W, W + (size_t)npad * 64, npad, k0, jb0); }, true);
    run("1 plain store instead of the read-modify-write", [&]() { hipLaunchKernelGGL(syrk128<1>, dim3(ntiles), dim3(256), 0, 0, S, W, W + (size_t)npad * 64, npad, k0, jb0); }, false);
    run("2 no operand loads in the loop", [&]() { hipLaunchKernelGGL(syrk128<2>, dim3(ntiles), dim3(256), 0, 0, S, W, W + (size_t)npad * 64, npad, k0, jb0); }, false);
    run("3 neither", [&]() { hipLaunchKernelGGL(syrk128<3>, dim3(ntiles), dim3(256), 0, 0, S, W, W + (size_t)npad * 64, npad, k0, jb0); }, false);
    run_variants(run, S, W, npad, k0, jb0, T128, ntiles);
    return 0;
}
