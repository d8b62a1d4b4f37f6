// standalone timing of the 16x16 diagonal-tile LDL' variants (scratch tool)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef double double4_t __attribute__((ext_vector_type(4)));
#define DEV __device__ __forceinline__
DEV double readlane_d(double x, int k) { return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), k), __builtin_amdgcn_readlane(__double2loint(x), k)); }
constexpr int P = 17;

template <int VARIANT>
__global__ void k(const double* tile_in, double* Wout, double* Liout, double* dout, long long* cyc) {
    __shared__ double tile[16 * P], Wb[16 * P], Li[16 * P], db[32];
    const int lane = threadIdx.x & 63, li = lane & 15, lk = lane >> 4;
    for (int i = lane; i < 256; i += 64) tile[(i >> 4) * P + (i & 15)] = tile_in[i];
    __syncthreads();
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int rep = 0; rep < 8; ++rep) {
    double4_t A, Bt;
#pragma unroll
    for (int r = 0; r < 4; ++r) { A[r] = tile[(lk + 4 * r) * P + li]; Bt[r] = (lk + 4 * r == li) ? 1.0 : 0.0; }
    if (VARIANT == 0) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int q = k & 3, r = k >> 2;
            const double w = A[r], bt = Bt[r];
            const double dk = readlane_d(w, 16 * q + k);
            double rdk = __builtin_amdgcn_rcp(dk);
            const bool rowq = lk == q;
            const double am = (rowq && li > k) ? w : 0.0;
            const double bm = rowq ? bt : 0.0;
            rdk = fma(fma(-dk, rdk, 1.0), rdk, rdk); rdk = fma(fma(-dk, rdk, 1.0), rdk, rdk);
            db[k] = dk; db[16 + k] = rdk;
            if (k < 15) {
                A = __builtin_amdgcn_mfma_f64_16x16x4f64(am, am * -rdk, A, 0, 0, 0);
                Bt = __builtin_amdgcn_mfma_f64_16x16x4f64(am, bm * -rdk, Bt, 0, 0, 0);
            }
        }
    } else if (VARIANT == 1) {   // A only (no inverse)
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int q = k & 3, r = k >> 2;
            const double w = A[r];
            const double dk = readlane_d(w, 16 * q + k);
            double rdk = __builtin_amdgcn_rcp(dk);
            const bool rowq = lk == q;
            const double am = (rowq && li > k) ? w : 0.0;
            rdk = fma(fma(-dk, rdk, 1.0), rdk, rdk); rdk = fma(fma(-dk, rdk, 1.0), rdk, rdk);
            db[k] = dk; db[16 + k] = rdk;
            if (k < 15) A = __builtin_amdgcn_mfma_f64_16x16x4f64(am, am * -rdk, A, 0, 0, 0);
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) { Wb[(lk + 4 * r) * P + li] = A[r]; Li[li * P + (lk + 4 * r)] = Bt[r]; }
    asm volatile("s_waitcnt lgkmcnt(0)");
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) cyc[0] = (t1 - t0) / 8;
    __syncthreads();
    for (int i = lane; i < 256; i += 64) { Wout[i] = Wb[(i >> 4) * P + (i & 15)]; Liout[i] = Li[(i >> 4) * P + (i & 15)]; }
    if (lane < 32) dout[lane] = db[lane];
}
int main() {
    double h[256]; srand(1);
    double M[16][16]; for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) M[i][j] = (rand() % 1000) / 1000.0 - 0.5;
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { double s = 0; for (int k = 0; k < 16; ++k) s += M[i][k] * M[j][k]; h[i * 16 + j] = s + (i == j ? 1.0 : 0.0); }
    double *tin, *W, *Li, *d; long long* cyc; hipMalloc(&tin, 2048); hipMalloc(&W, 2048); hipMalloc(&Li, 2048); hipMalloc(&d, 256); hipMalloc(&cyc, 64);
    hipMemcpy(tin, h, 2048, hipMemcpyHostToDevice);
    long long c;
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k<0>, dim3(1), dim3(64), 0, 0, tin, W, Li, d, cyc);
    hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost); printf("variant 0 (A + Bt mfma): %lld cycles per tile\n", c);
    double hd[32]; hipMemcpy(hd, d, 256, hipMemcpyDeviceToHost);
    // reference LDL'
    double Aref[16][16]; for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) Aref[i][j] = h[i * 16 + j];
    double dref[16]; for (int k = 0; k < 16; ++k) { dref[k] = Aref[k][k]; for (int i = k + 1; i < 16; ++i) { double l = Aref[i][k] / dref[k]; for (int j = k + 1; j < 16; ++j) Aref[i][j] -= l * Aref[k][j]; } }
    double err = 0; for (int k = 0; k < 16; ++k) err = fmax(err, fabs(hd[k] - dref[k]) / fabs(dref[k])); printf("max rel err of D: %.2e\n", err);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k<1>, dim3(1), dim3(64), 0, 0, tin, W, Li, d, cyc);
    hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost); printf("variant 1 (A only)     : %lld cycles per tile\n", c);
    return 0;
}
