// write-only / copy HBM bandwidth probe (scratch tool)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void fill16(double2* p, size_t n) { size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; size_t st = (size_t)gridDim.x * blockDim.x; for (; i < n; i += st) p[i] = double2{1.0, 2.0}; }
__global__ void fill8(double* p, size_t n) { size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; size_t st = (size_t)gridDim.x * blockDim.x; for (; i < n; i += st) p[i] = 1.0; }
__global__ void fillblk(double2* p, size_t n_per_blk) { double2* q = p + blockIdx.x * n_per_blk; for (size_t i = threadIdx.x; i < n_per_blk; i += blockDim.x) q[i] = double2{1.0, 2.0}; }
__global__ void copy16(const double2* a, double2* p, size_t n) { size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; size_t st = (size_t)gridDim.x * blockDim.x; for (; i < n; i += st) p[i] = a[i]; }
__global__ void read16(const double2* a, double* out, size_t n) { size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; size_t st = (size_t)gridDim.x * blockDim.x; double s = 0; for (; i < n; i += st) { double2 v = a[i]; s += v.x + v.y; } if (s == 1.2345) out[0] = s; }
int main() {
    size_t bytes = 154ull << 20; double2 *p, *a; hipMalloc(&p, bytes); hipMalloc(&a, bytes); hipMemset(a, 0, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto T = [&](const char* name, auto f, double mult) { for (int i = 0; i < 3; ++i) f(); hipEventRecord(e0); for (int i = 0; i < 20; ++i) f(); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); printf("%-28s %.1f us  %.0f GB/s\n", name, ms / 20 * 1e3, mult * bytes / (ms / 20 * 1e-3) / 1e9); };
    size_t n16 = bytes / 16;
    for (int g : {1024, 2048, 4096, 16384}) { char nm[64]; sprintf(nm, "fill16 grid %d", g); T(nm, [&] { hipLaunchKernelGGL(fill16, dim3(g), dim3(256), 0, 0, p, n16); }, 1); }
    T("fill8 grid 4096", [&] { hipLaunchKernelGGL(fill8, dim3(4096), dim3(256), 0, 0, (double*)p, bytes / 8); }, 1);
    T("fillblk 40KB/WG", [&] { hipLaunchKernelGGL(fillblk, dim3((unsigned)(bytes / 40960)), dim3(256), 0, 0, p, (size_t)2560); }, 1);
    T("copy16 grid 4096 (r+w)", [&] { hipLaunchKernelGGL(copy16, dim3(4096), dim3(256), 0, 0, a, p, n16); }, 2);
    T("read16 grid 4096", [&] { hipLaunchKernelGGL(read16, dim3(4096), dim3(256), 0, 0, a, (double*)p, n16); }, 1);
    T("hipMemsetAsync", [&] { hipMemsetAsync(p, 0, bytes, 0); }, 1);
    return 0;
}
