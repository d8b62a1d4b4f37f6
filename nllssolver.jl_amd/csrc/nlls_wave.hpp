// nlls_wave.hpp -- wavefront / workgroup reductions and block-element accessors shared by the sweep and cost translation units.
#pragma once
#include "nlls_internal.hpp"

namespace nlls {

constexpr int TPB = 256;
// one thread of a launch leaves the constant clock in the pinned host mirror (nlls_ctx::stamp_ptr: the device-timed NLLSResult buckets); nullptr: nothing
NLLS_DEV void time_stamp(double* __restrict__ stamps, int k) { if (stamps) stamps[k] = (double)wall_clock64(); }

template <int N, class F>
NLLS_DEV void static_for(F&& f) {
    [&]<int... I>(std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }(std::make_integer_sequence<int, N>{});
}

// this file is compiled with -fno-honor-nans (see the Makefile): NaN tests must look at the bits
// the bits of v behind an empty asm: without it the compiler recognises the integer tests below as floating-point comparisons and, in a
// file built with -fno-honor-nans, emits the ORDERED ones (seen: "(bits & ~sign) != 0" became v_cmp_lg_f64, false for NaN)
NLLS_DEV long long opaque_bits(double v) { long long b = __double_as_longlong(v); asm volatile("" : "+v"(b)); return b; }
NLLS_DEV bool is_nan_bits(double v) { return (opaque_bits(v) & 0x7fffffffffffffffLL) > 0x7ff0000000000000LL; }
// v != 0.0 by the bits (true for NaN)
NLLS_DEV bool nonzero_bits(double v) { return (opaque_bits(v) & 0x7fffffffffffffffLL) != 0; }
NLLS_DEV double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}
// wavefront sum on the VALU (DPP row shifts + row broadcasts, gfx9): the total lands in lane 63.  18 VALU
// instructions per value; the ds_bpermute form (__shfl_down) costs an LDS round trip per step.
template <int CTRL, int ROWMASK>
NLLS_DEV double dpp_add(double v) {
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const int slo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROWMASK, 0xf, true);
    const int shi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROWMASK, 0xf, true);
    return v + __hiloint2double(shi, slo);
}
NLLS_DEV double wave_sum_dpp63(double v) {
    v = dpp_add<0x111, 0xf>(v);   // row_shr:1
    v = dpp_add<0x112, 0xf>(v);   // row_shr:2
    v = dpp_add<0x114, 0xf>(v);   // row_shr:4
    v = dpp_add<0x118, 0xf>(v);   // row_shr:8   -> lane 15 of every row holds its row's sum
    v = dpp_add<0x142, 0xa>(v);   // row_bcast:15 into rows 1 and 3
    v = dpp_add<0x143, 0xc>(v);   // row_bcast:31 into rows 2 and 3 -> lane 63 holds the wavefront's sum
    return v;
}
NLLS_DEV double wave_max(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_down(v, o, 64));
    return v;
}
// deterministic workgroup sum (fixed tree); result valid in thread 0
NLLS_DEV double block_sum(double v, double* red /* >= TPB/64 doubles of LDS */) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[w] = v;
    __syncthreads();
    double t = 0;
    if (threadIdx.x == 0) for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += red[i];
    return t;
}
NLLS_DEV double block_max(double v, double* red) {
    v = wave_max(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[w] = v;
    __syncthreads();
    double t = 0;
    if (threadIdx.x == 0) for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t = fmax(t, red[i]);
    return t;
}

// Runs of equal keys over the lanes of a wavefront, cut at the 16-lane rows the DPP shifts work in (entries of an entry list are sorted by block row: the
// lanes of one row are neighbours).  rank = position of the lane in its (cut) run, tail = last lane of it.  `key` must not be 0.
struct WaveRuns { int rank; bool tail; int maxstep; };
NLLS_DEV WaveRuns wave_runs(uint32_t key) {
    const int lane = (int)(threadIdx.x & 63);
    const uint32_t prev = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)key, 0x111, 0xf, 0xf, true);   // row_shr:1 -- the first lane of a 16-lane row reads 0
    const unsigned long long m = __ballot(prev != key);                     // run starts
    const int start = 63 - __clzll((long long)(m & (~0ull >> (63 - lane))));
    WaveRuns r; r.rank = lane - start; r.tail = lane == 63 || ((m >> (lane + 1)) & 1ull);
    int mx = 0;                                                              // the longest run of the wavefront decides how many doubling steps a sum takes (uniform)
#pragma unroll
    for (int s = 1; s < 16; s <<= 1) if (__ballot(r.rank >= s)) mx = s;
    r.maxstep = mx;
    return r;
}
template <int CTRL>
NLLS_DEV double dpp_get(double v) {
    const int lo = __double2loint(v), hi = __double2hiint(v);
    return __hiloint2double(__builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true), __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true));
}
// Sum of v over the lanes of this lane's run, valid in the run's TAIL lane: a segmented inclusive scan in four DPP row shifts on the vector ALU -- where the lanes
// of a run would otherwise all add to ONE address of an LDS accumulator: same-address LDS atomics serialise at the full latency of a read-modify-write, lane by lane
// (measured at BASELINE config 5: ten lanes per point row = 17 of a 59 us launch; a crossbar-shuffle version of this sum was SLOWER than the atomics: it queues in the
// same LDS pipe)
NLLS_DEV double wave_run_sum(double v, const WaveRuns& r) {
    if (r.maxstep >= 1) { const double t = dpp_get<0x111>(v); v += r.rank >= 1 ? t : 0.0; }
    if (r.maxstep >= 2) { const double t = dpp_get<0x112>(v); v += r.rank >= 2 ? t : 0.0; }
    if (r.maxstep >= 4) { const double t = dpp_get<0x114>(v); v += r.rank >= 4 ? t : 0.0; }
    if (r.maxstep >= 8) { const double t = dpp_get<0x118>(v); v += r.rank >= 8 ? t : 0.0; }
    return v;
}

// element (i of slot SA, j of slot SB) of the block's local Hessian / gradient (residual.jl:91-107)
template <int KIND, int SA, int SB>
NLLS_DEV double h_elem(const BlockGH<KIND>& B, int i, int j) {
    using R = Res<KIND>; using I = ResInfo<KIND>;
    constexpr bool KA = R::ADAPT && SA == 0, KB = R::ADAPT && SB == 0;
    if constexpr (KA && KB) return B.Hkk(i, j);
    else if constexpr (KA) return B.Hkv(i, I::joff(SB) + j);
    else if constexpr (KB) return B.Hkv(j, I::joff(SA) + i);
    else return B.H(I::joff(SA) + i, I::joff(SB) + j);
}
template <int KIND, int SA>
NLLS_DEV double g_elem(const BlockGH<KIND>& B, int i) {
    using R = Res<KIND>; using I = ResInfo<KIND>;
    if constexpr (R::ADAPT && SA == 0) return B.Gk(i); else return B.G(I::joff(SA) + i);
}

// the small dense system (nlls_ctx::tiny_dense): the sum of the sweep workgroups' images of [A | b] (nlls_sweep.hip), the lower triangle of A mirrored.
// Element e of [A | b] (e < n^2 + n), summed by one lane.
NLLS_DEV void dense_tiny_gather_elem(const double* __restrict__ slab, int nimg, int n, double* __restrict__ A, double* __restrict__ b, int e) {
    const int n2 = n * n, imglen = n2 + n;
    const int r = e < n2 ? e % n : 0, cc = e < n2 ? e / n : 0;
    if (e < n2 && r < cc) return;                                 // (the upper triangle is written by its mirror's lane)
    double a0 = 0, a1 = 0, a2 = 0, a3 = 0; int k = 0;             // (four partial sums, fixed association)
    for (; k + 4 <= nimg; k += 4) { a0 += slab[(size_t)k * imglen + e]; a1 += slab[(size_t)(k + 1) * imglen + e]; a2 += slab[(size_t)(k + 2) * imglen + e]; a3 += slab[(size_t)(k + 3) * imglen + e]; }
    for (; k < nimg; ++k) a0 += slab[(size_t)k * imglen + e];
    const double v = (a0 + a1) + (a2 + a3);
    if (e >= n2) b[e - n2] = v; else { A[e] = v; if (r > cc) A[cc + (size_t)n * r] = v; }
}
// the end of that system's LM trial: the cost partials' sum and the trial's scalars published to the pinned host mirror (what trial_finish_kernel does for sparse systems).
// Rides as workgroup 0 of the look-ahead sweep's accumulate launch when there is one (nlls_lm_trial), else tiny_trial_finish_kernel.
// (struct DenseFin: nlls_ctx.hpp)
// final deterministic reduction of per-workgroup partials by ONE workgroup of TPB threads (the order every cost total is summed in)
NLLS_DEV void reduce_partials_body(const double* __restrict__ partials, int64_t n, double* __restrict__ out, double* red) {
    double acc = 0;
    for (int64_t i = threadIdx.x; i < n; i += TPB) acc += partials[i];
    double t = block_sum(acc, red);
    if (threadIdx.x == 0) out[0] = t;
}
NLLS_DEV void dense_fin_body(const DenseFin& f, double* red) {
    reduce_partials_body(f.cpart, f.ncp, f.out, red);
    if (f.host_out && threadIdx.x == 0) {
        double* h = f.host_out; const double* o = f.out;
        h[0] = o[0]; h[1] = o[1]; h[2] = o[2]; h[4] = o[4]; h[5] = o[5]; h[8] = o[8]; h[9] = o[9]; h[10] = o[10];
        __threadfence_system();
        reinterpret_cast<volatile double*>(h)[32] = f.seq; reinterpret_cast<volatile double*>(h)[33] = f.seq;
    }
}
}  // namespace nlls
