// What does a producer -> consumer hand-over INSIDE one launch cost against a launch boundary?  (round 6: would the block cyclic reduction's update jobs be better off as
// workgroups of the panel launch that wait for the panels' exports?)
//   two launches : P workgroups write T tiles of 2 KB each (plain stores), a second launch of C workgroups reads tiles of two producers each, multiplies (4 MFMAs per tile pair), stores
//   one launch   : the same producers, then __threadfence() + one relaxed agent-scope atomic add per producer; the consumers (higher block indices) poll the two counters they need,
//                  __threadfence(), then read -- everything else identical
//   chain        : K such steps one after the other (producer of step k + 1 = consumer of step k), as 2 K launches and as K fused launches and as ONE persistent launch
//   hipcc -O3 --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form -o handoff tools/microbench/handoff.hip && ./handoff
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));
constexpr int TILES = 16;                      // tiles per producer (a panel workgroup exports ~2 x 9 x 4 tiles)
template <bool BY = false>
__device__ __forceinline__ void produce(double* buf, int p, int step, double seed) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int t = wave; t < TILES; t += blockDim.x / 64) { d4 v = {seed + p, seed + t, seed + lane, seed + step}; double* q = buf + ((size_t)p * TILES + t) * 256 + 4 * lane;
        if constexpr (BY) { for (int i = 0; i < 4; ++i) __hip_atomic_store(q + i, v[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }     // past the (per-XCD, mutually incoherent) L2s
        else *reinterpret_cast<d4*>(q) = v; }
}
template <bool BY = false>
__device__ __forceinline__ double consume(const double* buf, int p0, int p1) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6; double acc0 = 0;
    for (int t = wave; t < TILES; t += blockDim.x / 64) {
        d4 a, b; const double* qa = buf + ((size_t)p0 * TILES + t) * 256 + 4 * lane; const double* qb = buf + ((size_t)p1 * TILES + t) * 256 + 4 * lane;
        if constexpr (BY) { for (int i = 0; i < 4; ++i) { a[i] = __hip_atomic_load(qa + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); b[i] = __hip_atomic_load(qb + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } }
        else { a = *reinterpret_cast<const d4*>(qa); b = *reinterpret_cast<const d4*>(qb); }
        d4 acc = {0, 0, 0, 0};
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[0], b[0], acc, 0, 0, 0); acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[1], b[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[2], b[2], acc, 0, 0, 0); acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[3], b[3], acc, 0, 0, 0);
        acc0 += acc[0] + acc[1] + acc[2] + acc[3];
    }
    return acc0;
}
__global__ __launch_bounds__(256) void k_produce(double* buf, int step, double seed) { produce(buf, blockIdx.x, step, seed); }
__global__ __launch_bounds__(256) void k_consume(const double* buf, double* out, int P) {
    const int c = blockIdx.x; const double v = consume(buf, c % P, (c + 1) % P); out[(size_t)c * 256 + threadIdx.x] = v;
}
// one launch: blocks [0, P) produce, blocks [P, P + C) consume behind the hand-over
__global__ __launch_bounds__(256) void k_fused(double* buf, double* out, unsigned* cnt, int P, int step, double seed, unsigned target) {
    if ((int)blockIdx.x < P) {
        produce(buf, blockIdx.x, step, seed);
        __threadfence(); __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_fetch_add(cnt + blockIdx.x, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    const int c = blockIdx.x - P, p0 = c % P, p1 = (c + 1) % P;
    if (threadIdx.x < 2) { const unsigned* q = cnt + (threadIdx.x ? p1 : p0); while (__hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1); }
    __syncthreads(); __threadfence();
    const double v = consume(buf, p0, p1); out[(size_t)c * 256 + threadIdx.x] = v;
}
// ONE persistent launch for K steps: P workgroups; in step k workgroup p consumes the tiles of (p, p + 1) of step k - 1 (behind the hand-over) and produces its own of step k
__global__ __launch_bounds__(256) void k_persist(double* buf0, double* buf1, double* out, unsigned* cnt, int P, int K, double seed) {
    const int p = blockIdx.x; double v = 0;
    for (int k = 0; k < K; ++k) {
        double* wb = (k & 1) ? buf1 : buf0; const double* rb = (k & 1) ? buf0 : buf1;
        if (k > 0) {
            if (threadIdx.x < 2) { const unsigned* q = cnt + (size_t)(k - 1) * P + (threadIdx.x ? (p + 1) % P : p); while (__hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < 1u) __builtin_amdgcn_s_sleep(1); }
            __syncthreads(); __threadfence();
            v += consume(rb, p, (p + 1) % P);
        }
        produce(wb, p, k, seed + v * 1e-300);
        __threadfence(); __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_fetch_add(cnt + (size_t)k * P + p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    out[(size_t)p * 256 + threadIdx.x] = v;
}
// the same persistent chain with the tiles themselves travelling past the L2s (relaxed agent-scope stores / loads, no fence; the counter behind s_waitcnt vmcnt(0))
__global__ __launch_bounds__(256) void k_persist_by(double* buf0, double* buf1, double* out, unsigned* cnt, int P, int K, double seed) {
    const int p = blockIdx.x; double v = 0;
    for (int k = 0; k < K; ++k) {
        double* wb = (k & 1) ? buf1 : buf0; const double* rb = (k & 1) ? buf0 : buf1;
        if (k > 0) {
            if (threadIdx.x < 2) { const unsigned* q = cnt + (size_t)(k - 1) * P + (threadIdx.x ? (p + 1) % P : p); while (__hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < 1u) __builtin_amdgcn_s_sleep(1); }
            __syncthreads();
            v += consume<true>(rb, p, (p + 1) % P);
        }
        produce<true>(wb, p, k, seed + v * 1e-300);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_fetch_add(cnt + (size_t)k * P + p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    out[(size_t)p * 256 + threadIdx.x] = v;
}
int main() {
    const int P = 48, C = 96, K = 7, REP = 200;
    double *buf, *buf1, *out; unsigned* cnt;
    (void)hipMalloc(&buf, sizeof(double) * P * TILES * 256); (void)hipMalloc(&buf1, sizeof(double) * P * TILES * 256); (void)hipMalloc(&out, sizeof(double) * 256 * (C + P)); (void)hipMalloc(&cnt, 4 * P * (K + 1));
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1); float ms;
    auto timeit = [&](const char* name, auto fn, int per) { for (int w = 0; w < 5; ++w) fn(w); (void)hipDeviceSynchronize(); (void)hipEventRecord(e0);
        for (int r = 0; r < REP; ++r) fn(5 + r); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); (void)hipEventElapsedTime(&ms, e0, e1);
        printf("%-64s %8.2f us per step\n", name, 1e3 * ms / REP / per); };
    unsigned target = 0;
    timeit("two launches (produce | consume)", [&](int r) { hipLaunchKernelGGL(k_produce, dim3(P), dim3(256), 0, 0, buf, r, 1.0 + r); hipLaunchKernelGGL(k_consume, dim3(C), dim3(256), 0, 0, buf, out, P); }, 1);
    (void)hipMemset(cnt, 0, 4 * P * (K + 1));
    timeit("one launch (producers, consumers behind fence + counter)", [&](int r) { ++target; hipLaunchKernelGGL(k_fused, dim3(P + C), dim3(256), 0, 0, buf, out, cnt, P, r, 1.0 + r, target); }, 1);
    timeit("chain of 7 steps as 14 launches", [&](int r) { for (int k = 0; k < K; ++k) { hipLaunchKernelGGL(k_produce, dim3(P), dim3(256), 0, 0, buf, k, 1.0 + r); hipLaunchKernelGGL(k_consume, dim3(C), dim3(256), 0, 0, buf, out, P); } }, K);
    timeit("chain of 7 steps as 7 fused launches", [&](int r) { for (int k = 0; k < K; ++k) { ++target; hipLaunchKernelGGL(k_fused, dim3(P + C), dim3(256), 0, 0, buf, out, cnt, P, k, 1.0 + r, target); } }, K);
    timeit("chain of 7 steps as ONE persistent launch (hand-over per step)", [&](int r) { (void)hipMemsetAsync(cnt, 0, 4 * P * (K + 1), 0); hipLaunchKernelGGL(k_persist, dim3(P), dim3(256), 0, 0, buf, buf1, out, cnt, P, K, 1.0 + r); }, K);
    timeit("... the tiles past the L2s (relaxed agent-scope accesses, no fence)", [&](int r) { (void)hipMemsetAsync(cnt, 0, 4 * P * (K + 1), 0); hipLaunchKernelGGL(k_persist_by, dim3(P), dim3(256), 0, 0, buf, buf1, out, cnt, P, K, 1.0 + r); }, K);
    timeit("one empty-ish launch (produce only)", [&](int r) { hipLaunchKernelGGL(k_produce, dim3(P), dim3(256), 0, 0, buf, r, 1.0 + r); }, 1);
    return 0;
}
