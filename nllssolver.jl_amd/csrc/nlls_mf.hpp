// nlls_mf.hpp -- constants and small device helpers shared by the two translation units of the matrix-free LM trial (nlls_mf.hip: elimination; nlls_mfb.hip: back-substitution)
#pragma once
#include "nlls_wave.hpp"

namespace nlls {

typedef double double4_t __attribute__((ext_vector_type(4)));
constexpr int MF_NW = 4;            // wavefronts per workgroup of the back-substitution
constexpr int MF_ENW = 2;           // wavefronts per supernode of the elimination, each taking every other batch of members (four: two workgroups per CU, and a workgroup's atomic flush -- its slot
                                    // held until the memory side has taken 1891 atomics -- left the CU half idle: 132 us; two: four workgroups per CU, every large supernode of BASELINE config 4 resident at once)
constexpr int MF_BMAX = 8;          // members per batch at most (one lane per cost block: 64 / blocks per member, capped)
constexpr int MF_TRMAX = 5;         // tile rows of [E | b]: nd + 1 <= 80
constexpr int MF_SLOTS = 40;        // members one wavefront handles at most (128 members per supernode)
// one supernode of the matrix-free trial, in launch order (nlls_ctx::d_mf_desc): everything a workgroup needs to start on it comes with one uniform load
// (struct MfDesc: nlls_ctx.hpp -- v0, nmem, nd, rc_off, eb0, obs0, B = members per batch)

// a wavefront's own LDS traffic: writes of some lanes, then reads by others.  The LDS pipe serves one wavefront's instructions in order; the compiler must not
// move them across this point, and the counter wait covers the returned data
NLLS_DEV void wave_lds_sync() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier(); }
NLLS_DEV double mf_rcp(double d) { const double r = __builtin_amdgcn_rcp(d); const double e = fma(-d, r, 1.0); return fma(r, fma(e, e, e), r); }   // v_rcp_f64 (2^-24) + one cubic step: 1.1e-16



// ---- the end of a matrix-free trial (nlls_mfb.hip) ---------------------------------------------------------------------------------------------------
constexpr int MF_PW = 8;            // doubles per row of partials: [x'Hx share, cost, max |x|, NaN flag, |x|^2, g'x, -, -]
NLLS_DEV double mfb_wave_max(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
    return v;
}
NLLS_DEV double mfb_wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}
// ONE workgroup of 256 threads sums the rows of partials (fixed order: the totals are bit-reproducible) into the trial's scalars -- out[0] cost, [1] max|x| (NaN if any entry
// is), [2] x'x, [4] x'(H + lambda I)x, [5] g'x, [8] x'Hx, [9] x'x, [10] factorisation status: what trial_finish_kernel leaves -- and publishes them to the pinned host mirror
// with the trial's sequence number (the host spins on it)
struct MfFin { const double* part; int nrows; double lambda; double* out; const int* status; double* host_out; double seq; double* stamps; };
NLLS_DEV void mf_finish_body(const MfFin& f, double (*red)[4]) {
    if (threadIdx.x == 0) time_stamp(f.stamps, 2);
    double q = 0, cost = 0, mx = 0, nan = 0, ss = 0, bx = 0;
    for (int i = threadIdx.x; i < f.nrows; i += 256) { const double* r = f.part + (size_t)i * MF_PW; q += r[0]; cost += r[1]; mx = fmax(mx, r[2]); nan = fmax(nan, r[3]); ss += r[4]; bx += r[5]; }
    q = mfb_wave_sum(q); cost = mfb_wave_sum(cost); ss = mfb_wave_sum(ss); bx = mfb_wave_sum(bx); mx = mfb_wave_max(mx); nan = mfb_wave_max(nan);
    if ((threadIdx.x & 63) == 0) { const int w = threadIdx.x >> 6; red[0][w] = q; red[1][w] = cost; red[2][w] = mx; red[3][w] = nan; red[4][w] = ss; red[5][w] = bx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        q = red[0][0] + red[0][1] + red[0][2] + red[0][3]; cost = red[1][0] + red[1][1] + red[1][2] + red[1][3];
        ss = red[4][0] + red[4][1] + red[4][2] + red[4][3]; bx = red[5][0] + red[5][1] + red[5][2] + red[5][3];
        mx = fmax(fmax(red[2][0], red[2][1]), fmax(red[2][2], red[2][3])); nan = fmax(fmax(red[3][0], red[3][1]), fmax(red[3][2], red[3][3]));
        double* out = f.out;
        out[0] = cost; out[1] = nan > 0 ? __longlong_as_double(0x7ff8000000000000LL) : mx; out[2] = ss;
        out[4] = q + f.lambda * ss; out[5] = bx; out[8] = q; out[9] = ss; out[10] = (double)f.status[0];
        time_stamp(f.stamps, 3);
        if (f.host_out) {
            double* h = f.host_out;
            h[0] = out[0]; h[1] = out[1]; h[2] = out[2]; h[4] = out[4]; h[5] = out[5]; h[8] = out[8]; h[9] = out[9]; h[10] = out[10];
            __threadfence_system();
            reinterpret_cast<volatile double*>(h)[32] = f.seq; reinterpret_cast<volatile double*>(h)[33] = f.seq;
        }
    }
}

}  // namespace nlls
