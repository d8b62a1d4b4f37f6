#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void probe(const double* A, const double* B, double* D) {
    const int l = threadIdx.x;
    double d = 0.0;
    d = __builtin_amdgcn_mfma_f64_4x4x4f64(A[l], B[l], d, 0, 0, 0);
    D[l] = d;
}
int main() {
    double *A, *B, *D; hipMallocManaged(&A, 512); hipMallocManaged(&B, 512); hipMallocManaged(&D, 512);
    // A one-hot at lane a, B all ones -> which D lanes are nonzero
    for (int a = 0; a < 64; ++a) {
        for (int i = 0; i < 64; ++i) { A[i] = (i == a); B[i] = 1.0; }
        hipLaunchKernelGGL(probe, 1, 64, 0, 0, A, B, D); hipDeviceSynchronize();
        printf("A1hot %2d ->", a); for (int i = 0; i < 64; ++i) if (D[i] != 0) printf(" %d", i); printf("\n");
    }
    for (int b = 0; b < 64; ++b) {
        for (int i = 0; i < 64; ++i) { B[i] = (i == b); A[i] = 1.0; }
        hipLaunchKernelGGL(probe, 1, 64, 0, 0, A, B, D); hipDeviceSynchronize();
        printf("B1hot %2d ->", b); for (int i = 0; i < 64; ++i) if (D[i] != 0) printf(" %d", i); printf("\n");
    }
    // k pairing: A one-hot at lane a (a in 0..15), B one-hot at lane b: nonzero?
    for (int a = 0; a < 16; ++a) { printf("pair A%2d:", a);
        for (int b = 0; b < 16; ++b) {
            for (int i = 0; i < 64; ++i) { A[i] = (i == a); B[i] = (i == b); }
            hipLaunchKernelGGL(probe, 1, 64, 0, 0, A, B, D); hipDeviceSynchronize();
            int hit = -1; for (int i = 0; i < 64; ++i) if (D[i] != 0) hit = i;
            if (hit >= 0) printf(" (B%d->D%d)", b, hit);
        }
        printf("\n"); }
    return 0;
}
