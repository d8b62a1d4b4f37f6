// nlls_post.hpp -- what the iterators ask about the step of the last solve (fast_bAb(H, x), dot(g, x), maximum(abs, x), |x|^2:
// src/iterators.jl:163, src/utils.jl:71-106, src/optimize.jl:149): device bodies shared by the solve and the cost translation units -- in an LM
// trial they run as extra workgroups of the cost sweep's launch (nlls_cost.hip), elsewhere in a launch of their own (post_solve_kernel, nlls_solve.hip).
#pragma once
#include "nlls_wave.hpp"

namespace nlls {

NLLS_DEV double post_wsum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}
constexpr int QF_COLS = 8;      // threads per block of the quadratic form (one per column; blocks with more rows / columns than this take a loop)
// (bodies take a virtual workgroup index / count, so that post_solve_kernel can run several of them in one launch)
// One thread per (block, column): the column's entries and the entries of v it multiplies are requested at once (a loop over a run-time number of
// rows waits for every load before it issues the next: 36 dependent round trips for a 6 x 6 block, the longest chain of the whole launch).
__device__ __forceinline__ void quadform_blocks_body(const double* __restrict__ A, const SchurCopy* __restrict__ blk, int64_t nblk,
                                                     const double* __restrict__ v, const uint8_t* __restrict__ mask, double* __restrict__ partials, int bid, int nb, double* __restrict__ out_one = nullptr) {
    __shared__ double red[4];
    constexpr int MB = QF_COLS;
    double acc = 0;
    for (int64_t q = (int64_t)bid * 256 + threadIdx.x; q < nblk * MB; q += (int64_t)nb * 256) {
        const int64_t k = q / MB; const int j = (int)(q - k * MB);
        if (mask && !mask[k]) continue;
        const SchurCopy bk = blk[k];
        if (j >= bk.cols) continue;
        if (bk.rows <= MB) {
            double a[MB], w[MB];
#pragma unroll
            for (int i = 0; i < MB; ++i) { const bool in = i < bk.rows; a[i] = in ? A[bk.off + i + bk.rows * j] : 0.0; w[i] = in ? v[bk.r + i] : 0.0; }
            double c2 = 0;
#pragma unroll
            for (int i = 0; i < MB; ++i) c2 += a[i] * w[i];
            const double t = c2 * v[bk.c + j];
            acc += (bk.r == bk.c) ? t : 2.0 * t;
        } else {      // (dynamic-size blocks)
            double c2 = 0; for (int i = 0; i < bk.rows; ++i) c2 += A[bk.off + i + bk.rows * j] * v[bk.r + i];
            for (int jj = j; jj < bk.cols; jj += MB) { double c3 = c2; if (jj != j) { c3 = 0; for (int i = 0; i < bk.rows; ++i) c3 += A[bk.off + i + bk.rows * jj] * v[bk.r + i]; }
                const double t = c3 * v[bk.c + jj]; acc += (bk.r == bk.c) ? t : 2.0 * t; }
        }
    }
    acc = post_wsum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) { const double t = red[0] + red[1] + red[2] + red[3]; if (out_one) *out_one = t; else partials[bid] = t; }
}
// rows of fast-path members, for the step x of the last solve: x' A x restricted to row v is
//   2 x_v' (E_v x_R) + x_v' C_v x_v,   E_v x_R = -E_v s  -- and E_v s is what schur_backsub_fast_kernel left in tE
template <int DV>
__device__ __forceinline__ void quadform_points_body(const double* __restrict__ A, const int64_t* __restrict__ ediag, const uint32_t* __restrict__ eboff,
                                                     const uint32_t* __restrict__ members, int64_t nm, const double* __restrict__ tE,
                                                     const double* __restrict__ x, double* __restrict__ partials, int bid, int nb) {
    __shared__ double red[4];
    double acc = 0;
    for (int64_t i = (int64_t)bid * 256 + threadIdx.x; i < nm; i += (int64_t)nb * 256) {
        const uint32_t v = members ? members[i] : (uint32_t)i;      // (nullptr: every eliminated block is a fast-path member -- one load less in the chain)
        double xv[DV], t = 0;
#pragma unroll
        for (int a2 = 0; a2 < DV; ++a2) { xv[a2] = x[eboff[v] + a2]; t -= 2.0 * xv[a2] * tE[(int64_t)v * DV + a2]; }
#pragma unroll
        for (int j = 0; j < DV; ++j) { double c2 = 0;
#pragma unroll
            for (int i2 = 0; i2 < DV; ++i2) c2 += A[ediag[v] + i2 + DV * j] * xv[i2];
            t += c2 * xv[j]; }
        acc += t;
    }
    acc = post_wsum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partials[bid] = red[0] + red[1] + red[2] + red[3];
}
// Everything the iterators ask about the step x of the last solve -- maximum(abs, x), |x|^2 (src/optimize.jl:149,
// src/callbacks.jl:47), fast_bAb(H, x) and dot(g, x) (src/iterators.jl:163) -- in ONE launch plus one finishing workgroup
// (six launches before; a launch boundary costs ~5 us here).  Workgroups [0, np): blocks of H outside the fast-path rows;
// [np, np + np3): the fast-path rows from E_v s; [np + np3, np + np3 + np2): one pass over x and b.
struct PostSolveArgs { const double* A; const SchurCopy* blk; int64_t nblk; const uint8_t* blkmask; const int64_t* ediag; const uint32_t* eboff;
                       const uint32_t* members; int64_t nm; const double* tE; const double* x; const double* b; const double* dofmask; const double* dofmask_b; int64_t ndof;
                       double* partials; double* part2; int np, np3, np2;
                       double* stamps;                         /* nlls_ctx::stamp_ptr: the launch that carries these roles stamps the start of the trial's cost sweep */
                       int dv;                                 /* block size of the fast-path members (the run-time switch of post_roles_any) */
                       int nretract; const int32_t* vkind; const int32_t* vdim; const uint32_t* voff; const uint32_t* vboff; int64_t nvar; const double* vfrom; double* vto; };
// the roles of one virtual workgroup bid (256 threads) of [0, np + np3 + np2 (+ nretract)): blocks of H outside the fast-path rows, the fast-path rows
// from E_v s, one pass over x and b, and -- where the caller asks for it -- the retraction of the LM trial
template <int DV>
NLLS_DEV void post_roles_body(const PostSolveArgs& a, int bid) {
    if (bid < a.np) { quadform_blocks_body(a.A, a.blk, a.nblk, a.x, a.blkmask, a.partials, bid, a.np); return; }
    if (bid < a.np + a.np3) { quadform_points_body<DV>(a.A, a.ediag, a.eboff, a.members, a.nm, a.tE, a.x, a.partials + a.np, bid - a.np, a.np3); return; }
    if (bid >= a.np + a.np3 + a.np2) {                          // the retraction of the LM trial (update!, src/iterators.jl:155): one thread per variable
        const int64_t i = (int64_t)(bid - a.np - a.np3 - a.np2) * 256 + threadIdx.x;
        if (i < a.nvar) retract_one(a.vkind, a.vdim, a.voff, a.vboff, i, a.vfrom, a.x, a.vto);
        return;
    }
    __shared__ double red[5][4];
    const int b2 = bid - a.np - a.np3;
    double m = 0, ss = 0, vv = 0, bv = 0, nan = 0;
    for (int64_t i = (int64_t)b2 * 256 + threadIdx.x; i < a.ndof; i += (int64_t)a.np2 * 256) {
        const double x = a.x[i], w = a.dofmask ? a.dofmask[i] : 1.0, wb = a.dofmask_b ? a.dofmask_b[i] : w;
        if (x != x) nan = 1.0;
        m = fmax(m, w * fabs(x)); ss += x * x; vv += w * x * x; bv += wb * a.b[i] * x;    // (w: this rank's share under sharding, 1 otherwise; wb: its share of g -- the reduced part too while the reduced rows are not summed over ranks)
    }
    ss = post_wsum(ss); vv = post_wsum(vv); bv = post_wsum(bv);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { m = fmax(m, __shfl_xor(m, o)); nan = fmax(nan, __shfl_xor(nan, o)); }
    if ((threadIdx.x & 63) == 0) { const int w = threadIdx.x >> 6; red[0][w] = m; red[1][w] = nan; red[2][w] = ss; red[3][w] = vv; red[4][w] = bv; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double* o = a.part2 + 5 * b2;
        o[0] = fmax(fmax(red[0][0], red[0][1]), fmax(red[0][2], red[0][3])); o[1] = fmax(fmax(red[1][0], red[1][1]), fmax(red[1][2], red[1][3]));
        o[2] = red[2][0] + red[2][1] + red[2][2] + red[2][3]; o[3] = red[3][0] + red[3][1] + red[3][2] + red[3][3]; o[4] = red[4][0] + red[4][1] + red[4][2] + red[4][3];
    }
}

NLLS_DEV void post_roles_any(const PostSolveArgs& a, int bid) { if (a.dv == 3) post_roles_body<3>(a, bid); else if (a.dv == 2) post_roles_body<2>(a, bid); else post_roles_body<1>(a, bid); }

}  // namespace nlls
