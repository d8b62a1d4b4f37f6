// nlls_solve.hip -- damped normal-equation solve on the device (gfx950).
//
// Replaces  negate!(solve!(linsystem, options))  src/iterators.jl:152 -> src/linearsolver.jl:28-32 together
// with the damping uniformscaling!(hessian, k) (src/iterators.jl:149) and fast_bAb / dot
// (src/iterators.jl:163, src/utils.jl:95-106).
//
// The reference factors the FULL sparse system with LDLFactorizations; it has no Schur complement
// (SURVEY F1).  This path is new: an independent set of blocks (bundle adjustment: the points) is
// eliminated block-wise (C_v^-1 by a small Cholesky in LDS), the reduced system
//   S = B + lambda*I - sum_v E_v' C_v^-1 E_v ,   s = b_R - sum_v E_v' C_v^-1 b_v
// is assembled densely and factored by a blocked right-looking LDL' whose trailing update runs on
// the fp64 matrix cores (v_mfma_f64_16x16x4_f64); the eliminated blocks are recovered by
// back-substitution.  Parity contract: x solves (H + lambda*I) x = -b, unique for SPD systems.
#include <cstdlib>
#include <utility>

#include "nlls_wave.hpp"
#include "nlls_post.hpp"
#include "nlls_slayout.hpp"

namespace nlls {

constexpr int LDT = 80;         // LDS leading dimension of a 64-row operand tile (80 = 16 mod 32: conflict-free ds_read_b64)

typedef double double4_t __attribute__((ext_vector_type(4)));

NLLS_DEV double wsum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

// identity on the padding of the dense layout; the rhs as row n: factoring the bordered matrix
// [[S, s], [s', c]] leaves z = D^-1 L^-1 s in row n of the factor, so no separate forward substitution is needed.
template <class LAY = SLayout>
__global__ void schur_init_kernel(LAY L, double* __restrict__ s, const double* __restrict__ b, const uint32_t* __restrict__ red_boff) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < L.n) { *L.rhs(s, i) = b[red_boff[i]]; return; }
    if (!LAY_IS_TSP(L) && L.mode != SOLVE_BAND && i < L.npad) { s[i] = 0.0; L.S[(size_t)i + (size_t)L.npad * i] = (i == L.n) ? 1e300 : 1.0; }
}
// schur_init + schur_copy + the status reset in one launch (sparse systems): the first ninit workgroups initialise s (and
// the padding of the dense layout), the rest copy one reduced-reduced block each
template <class LAY = SLayout>
__global__ __launch_bounds__(256) void schur_prepare_kernel(LAY L, double* __restrict__ s, const double* __restrict__ b, const uint32_t* __restrict__ red_boff,
                                                            const double* __restrict__ A, const SchurCopy* __restrict__ copies, double lambda, int ninit, int* __restrict__ status) {
    if (blockIdx.x == 0 && threadIdx.x < 5) status[threadIdx.x] = 0;
    if ((int)blockIdx.x < ninit) {
        const int i = blockIdx.x * 256 + threadIdx.x;
        if (i < L.n) { *L.rhs(s, i) = b[red_boff[i]]; return; }
        if (!LAY_IS_TSP(L) && L.mode != SOLVE_BAND && i < L.npad) { s[i] = 0.0; L.S[(size_t)i + (size_t)L.npad * i] = (i == L.n) ? 1e300 : 1.0; }
        return;
    }
    const SchurCopy cp = copies[blockIdx.x - ninit];
    for (int e = threadIdx.x; e < cp.rows * cp.cols; e += 256) {
        const int i = e % cp.rows, j = e / cp.rows;
        double v = A[cp.off + e];
        if (cp.r == cp.c) { if (i < j) continue; if (i == j) v += lambda; *L.at(cp.r + i, cp.c + j) = v; }
        else if (cp.r > cp.c) *L.at(cp.r + i, cp.c + j) = v;
        else *L.at(cp.c + j, cp.r + i) = v;          // the border reordering flipped this block: store its transpose
    }
}
template <class LAY = SLayout>
__global__ void schur_copy_kernel(LAY L, const double* __restrict__ A, const SchurCopy* __restrict__ copies, double lambda) {
    const SchurCopy cp = copies[blockIdx.x];
    for (int e = threadIdx.x; e < cp.rows * cp.cols; e += blockDim.x) {
        const int i = e % cp.rows, j = e / cp.rows;
        double v = A[cp.off + e];
        if (cp.r == cp.c) { if (i < j) continue; if (i == j) v += lambda; *L.at(cp.r + i, cp.c + j) = v; }
        else if (cp.r > cp.c) *L.at(cp.r + i, cp.c + j) = v;
        else *L.at(cp.c + j, cp.r + i) = v;          // the border reordering flipped this block: store its transpose
    }
}
__global__ void dense_to_S_kernel(double* __restrict__ S, const double* __restrict__ A, double lambda, int n, int npad) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (int64_t)n * n) return;
    const int i = (int)(e % n), j = (int)(e / n);
    S[(size_t)i + (size_t)npad * j] = A[e] + (i == j ? lambda : 0.0);
}

// One wavefront per SUPERNODE = run of eliminated blocks with identical neighbour columns (bundle adjustment:
// consecutive points seen by the same cameras).  Per block v:  Y = C_v^-1 [E_v | b_v]  (LDL' of C_v in LDS);
// the products E_v' Y_E (lower triangle) and E_v' y_b are summed over the run in LDS accumulators owned
// lane-wise, then flushed once with HBM atomics -- 50-100x fewer atomics than one flush per block.
// LDS: C (dv x dv), E (dv x nd), Y (dv x (nd+1)), acc (nd(nd+1)/2 + nd), column map.
template <class LAY = SLayout>
__global__ __launch_bounds__(64) void schur_elim_kernel(const double* __restrict__ A, const double* __restrict__ b,
                                                        const int64_t* __restrict__ eptr, const SchurNbr* __restrict__ enbr,
                                                        const int64_t* __restrict__ ediag, const uint32_t* __restrict__ eboff,
                                                        const uint16_t* __restrict__ edim, const uint32_t* __restrict__ egroup,
                                                        const uint32_t* __restrict__ glist, double lambda, int maxdv, int maxnd, int use_acc,
                                                        LAY L, double* __restrict__ s, int* __restrict__ status) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int lane = threadIdx.x;
    double* C = sm;                                   // dv*dv
    double* E = C + maxdv * maxdv;                    // dv*nd  (col-major, column = reduced dof)
    double* Y = E + (size_t)maxdv * maxnd;            // dv*(nd+1)
    int64_t* csrc = reinterpret_cast<int64_t*>(Y + (size_t)maxdv * (maxnd + 1));   // nd: A.data offset of E(0, column)
    int64_t* nbo = csrc + maxnd;                                                    // neighbour block offsets (<= nd)
    uint32_t* rc = reinterpret_cast<uint32_t*>(nbo + maxnd);                        // nd: reduced column
    uint32_t* cstr = rc + maxnd;                                                    // nd: stride between rows of E in A.data
    uint32_t* nbi = cstr + maxnd;                                                   // neighbour rcol | dim << 24 | trans << 31
    double* acc = reinterpret_cast<double*>(nbi + maxnd + (maxnd & 1));             // pairs + nd
    const uint32_t g = glist[blockIdx.x];
    const uint32_t v0 = egroup[g], v1 = egroup[g + 1];
    int nd = 0, npairs = 0;
    for (uint32_t v = v0; v < v1; ++v) {
        const int dv = edim[v];
        const int64_t p0 = eptr[v], p1 = eptr[v + 1];
        __syncthreads();
        // gather: neighbour descriptors first (one coalesced load), then every element of [C | E | b] in parallel
        // through per-column (offset, stride) descriptors -- two memory latencies per block instead of one per neighbour
        const int nnb = (int)(p1 - p0);
        for (int p = lane; p < nnb; p += 64) { const SchurNbr nb = enbr[p0 + p]; nbo[p] = nb.off; nbi[p] = nb.rcol | ((uint32_t)nb.dim << 24) | ((uint32_t)nb.trans << 31); }
        for (int e = lane; e < dv * dv; e += 64) { const int i = e % dv, j = e / dv; C[e] = A[ediag[v] + e] + (i == j ? lambda : 0.0); }
        __syncthreads();
        int ndv = 0;
        for (int p = 0; p < nnb; ++p) {
            const uint32_t info = nbi[p]; const int du = (info >> 24) & 127, tr = info >> 31;
            for (int c2 = lane; c2 < du; c2 += 64) { csrc[ndv + c2] = nbo[p] + (tr ? c2 : (int64_t)dv * c2); cstr[ndv + c2] = tr ? du : 1; if (v == v0) rc[ndv + c2] = (info & 0xFFFFFF) + c2; }
            ndv += du;
        }
        __syncthreads();
        for (int e0 = lane; e0 < dv * ndv; e0 += 4 * 64) {      // (four loads in flight per lane: a copy loop waits for each load before the next)
            double g4[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { const int e = e0 + 64 * u; g4[u] = 0.0; if (e < dv * ndv) { const int a2 = e % dv, c2 = e / dv; g4[u] = A[csrc[c2] + (int64_t)a2 * cstr[c2]]; } }
#pragma unroll
            for (int u = 0; u < 4; ++u) { const int e = e0 + 64 * u; if (e < dv * ndv) E[e] = g4[u]; }
        }
        for (int a = lane; a < dv; a += 64) Y[a + dv * ndv] = b[eboff[v] + a];
        if (v == v0) { nd = ndv; npairs = nd * (nd + 1) / 2; if (use_acc) for (int t = lane; t < npairs + nd; t += 64) acc[t] = 0.0; }
        __syncthreads();
        // LDL' of C in place (unit lower L below the diagonal, D on it), lane-serial: dv <= 32.  No positivity is
        // required (the reference's LDLFactorizations has none either); only an exactly zero pivot fails.
        if (lane == 0) {
            for (int j = 0; j < dv; ++j) {
                double d = C[j + dv * j];
                for (int k = 0; k < j; ++k) d -= C[j + dv * k] * C[j + dv * k] * C[k + dv * k];
                if (d == 0.0 || d != d) { atomicCAS(status, 0, 1); d = 1.0; }
                C[j + dv * j] = d;
                for (int i = j + 1; i < dv; ++i) { double t = C[i + dv * j]; for (int k = 0; k < j; ++k) t -= C[i + dv * k] * C[j + dv * k] * C[k + dv * k]; C[i + dv * j] = t / d; }
            }
        }
        __syncthreads();
        // Y(:, c) = C^-1 [E | b](:, c): one column per lane, solved in place in LDS
        for (int c2 = lane; c2 <= nd; c2 += 64) {
            double* y = Y + dv * c2;
            for (int i = 0; i < dv; ++i) { double t = (c2 < nd) ? E[i + dv * c2] : y[i]; for (int k = 0; k < i; ++k) t -= C[i + dv * k] * y[k]; y[i] = t; }
            for (int i = 0; i < dv; ++i) y[i] /= C[i + dv * i];
            for (int i = dv - 1; i >= 0; --i) { double t = y[i]; for (int k = i + 1; k < dv; ++k) t -= C[k + dv * i] * y[k]; y[i] = t; }
        }
        __syncthreads();
        // pair (p >= q): E(:,p)' Y(:,q); walk (p, q) incrementally from the lane's first pair
        {
            int t = lane;
            int p = (int)((sqrt(8.0 * t + 1.0) - 1.0) * 0.5);
            while (p * (p + 1) / 2 > t) --p;
            while ((p + 1) * (p + 2) / 2 <= t) ++p;
            int q = t - p * (p + 1) / 2;
            for (; t < npairs; t += 64) {
                double a2 = 0; for (int a = 0; a < dv; ++a) a2 += E[a + dv * p] * Y[a + dv * q];
                if (use_acc) acc[t] += a2; else atomicAdd(L.at(rc[p] > rc[q] ? rc[p] : rc[q], rc[p] > rc[q] ? rc[q] : rc[p]), -a2);
                q += 64; while (q > p) { q -= p + 1; ++p; }
            }
        }
        for (int p = lane; p < nd; p += 64) { double a2 = 0; for (int a = 0; a < dv; ++a) a2 += E[a + dv * p] * Y[a + dv * nd];
            if (use_acc) acc[npairs + p] += a2; else atomicAdd(L.rhs(s, rc[p]), -a2); }
    }
    if (use_acc) {
        int t = lane;
        int p = (int)((sqrt(8.0 * t + 1.0) - 1.0) * 0.5);
        while (p * (p + 1) / 2 > t) --p;
        while ((p + 1) * (p + 2) / 2 <= t) ++p;
        int q = t - p * (p + 1) / 2;
        for (; t < npairs; t += 64) { atomicAdd(L.at(rc[p] > rc[q] ? rc[p] : rc[q], rc[p] > rc[q] ? rc[q] : rc[p]), -acc[t]); q += 64; while (q > p) { q -= p + 1; ++p; } }
        for (int p2 = lane; p2 < nd; p2 += 64) atomicAdd(L.rhs(s, rc[p2]), -acc[npairs + p2]);
    }
}

// (C_v + lambda I)^-1 of every eliminated block of the compile-time size DV, once per solve: the elimination and the
// back-substitution of the fast path then need no factorisation, no division and no dependent chain per member.
template <int DV>
__global__ __launch_bounds__(256) void schur_cinv_kernel(const double* __restrict__ A, const int64_t* __restrict__ ediag, const uint16_t* __restrict__ edim,
                                                         int64_t nel, double lambda, double* __restrict__ Cinv, int* __restrict__ status) {
    const int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (v >= nel || edim[v] != DV) return;
    double C[DV * DV];
#pragma unroll
    for (int j = 0; j < DV; ++j)
#pragma unroll
        for (int i = j; i < DV; ++i) C[i + DV * j] = A[ediag[v] + i + DV * j];
#pragma unroll
    for (int j = 0; j < DV; ++j) {                            // LDL'
        double d = C[j + DV * j] + lambda;
#pragma unroll
        for (int k = 0; k < j; ++k) d -= C[j + DV * k] * C[j + DV * k] * C[k + DV * k];
        if (d == 0.0 || d != d) { atomicCAS(status, 0, 1); d = 1.0; }
        C[j + DV * j] = d;
#pragma unroll
        for (int i = j + 1; i < DV; ++i) { double t = C[i + DV * j];
#pragma unroll
            for (int k = 0; k < j; ++k) t -= C[i + DV * k] * C[j + DV * k] * C[k + DV * k];
            C[i + DV * j] = t / d; }
    }
#pragma unroll
    for (int c2 = 0; c2 < DV; ++c2) {                          // column c2 of the inverse
        double y[DV];
#pragma unroll
        for (int i = 0; i < DV; ++i) { double t = (i == c2) ? 1.0 : 0.0;
#pragma unroll
            for (int k = 0; k < i; ++k) t -= C[i + DV * k] * y[k]; y[i] = t; }
#pragma unroll
        for (int i = 0; i < DV; ++i) y[i] /= C[i + DV * i];
#pragma unroll
        for (int i = DV - 1; i >= 0; --i) { double t = y[i];
#pragma unroll
            for (int k = i + 1; k < DV; ++k) t -= C[k + DV * i] * y[k]; y[i] = t; }
#pragma unroll
        for (int i = 0; i < DV; ++i) Cinv[v * (DV * DV) + i + DV * c2] = y[i];
    }
}

// x_v = (C_v + lambda I)^-1 (b_v - E_v x_R), stored negated, for the members of fast-path supernodes: one wavefront per
// supernode, four members at a time (16 lanes each).  A lane owns columns l, l + 16, ... of E_v -- the same reduced
// columns for every member of the supernode, so their x_R entries stay in registers -- and the 16 partial sums per
// component meet in lane 15 of the row through DPP row shifts.
template <int CTRL>
NLLS_DEV double row_shr_add(double v) {
    const int lo = __double2loint(v), hi = __double2hiint(v);
    return v + __hiloint2double(__builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true), __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true));
}
// The retraction of an LM trial (update!(to, from, x), src/iterators.jl:155) rides in this launch (on != 0): the supernode's wavefront retracts its own
// members from the step it has just formed (Euclidean variables of DV entries: checked at upload), workgroups behind the others retract every
// other variable from the reduced solution itself (x_R = -xr: the scatter of this very launch is not visible to them) -- the cost sweep is then the
// next launch, and the step statistics ride in ITS launch (nlls_post.hpp): no launch of their own for either.
template <int DV>
__global__ __launch_bounds__(64) void schur_backsub_fast_kernel(const double* __restrict__ A, const double* __restrict__ b,
                                                                const ElimDesc* __restrict__ desc, const uint32_t* __restrict__ rcflat,
                                                                const double* __restrict__ Cinv, const double* __restrict__ xr, double* __restrict__ x, double* __restrict__ tE,
                                                                uint32_t ngroups, const uint32_t* __restrict__ red_boff, int nred, int write_red,
                                                                double* __restrict__ Szero, int64_t nzero, uint32_t nextra, BsfRetract rt) {
    constexpr int MAXC = (72 + 15) / 16;                      // columns per lane (nd <= 72)
    __shared__ uint32_t rc[80];
    const int lane = threadIdx.x, l = lane & 15, gsub = lane >> 4;
    if (blockIdx.x == 0 && lane == 0) time_stamp(rt.stamps, 1);
    if (blockIdx.x >= ngroups) { backsub_rest_roles(blockIdx.x - ngroups, nextra, lane, xr, x, red_boff, nred, write_red, Szero, nzero, rt); return; }
    const ElimDesc d = desc[blockIdx.x];                       // uniform: one scalar load
    const uint32_t v0 = d.v0, v1 = d.v0 + d.nmem; const int nd = (int)d.nd;
    for (int c2 = lane; c2 < nd; c2 += 64) rc[c2] = rcflat[d.rc_off + c2];
    __syncthreads();
    double xw[MAXC];
#pragma unroll
    for (int k = 0; k < MAXC; ++k) { const int col = l + 16 * k; xw[k] = col < nd ? xr[rc[col]] : 0.0; }
    // (rt.on) the members' variables, two per lane, requested NOW: at the end of the loop only an add and a store are left
    uint32_t ro[2] = {0, 0}; double rv[2][DV];
    if (rt.on) {
#pragma unroll
        for (int h = 0; h < 2; ++h) { const uint32_t m = lane + 64 * h; ro[h] = rt.fast_voff[v0 + (m < v1 - v0 ? m : 0)];
#pragma unroll
            for (int a2 = 0; a2 < DV; ++a2) rv[h][a2] = rt.vfrom[ro[h] + a2]; }
    }
    // The members of a supernode are consecutive block rows (constant stride in A.data, b and x: nlls_structure.cpp), so
    // nothing is looked up per member.  Four members per step, one per 16 lanes; the loads of the next step are issued
    // before this one is reduced (two register sets), and the results wait in LDS until the loop is over -- a store
    // between the loads would make every wait a full vmcnt(0) and serialise the steps again.
    __shared__ double xs[128 * DV], ts[128 * DV];
    const int64_t dg0 = d.dg0, dstride = (int64_t)DV * nd + DV * DV; const uint32_t eb0 = d.eb0;
    struct Step { double a[MAXC][DV], bv[DV], ci[DV * DV]; };
    auto load = [&](uint32_t vb, Step& S) {
        const uint32_t v = vb + gsub; const uint32_t m = v < v1 ? v - v0 : 0u;
        const int64_t seg = dg0 + (int64_t)m * dstride - (int64_t)DV * nd;
#pragma unroll
        for (int k = 0; k < MAXC; ++k) { const int col = l + 16 * k;
#pragma unroll
            for (int a2 = 0; a2 < DV; ++a2) S.a[k][a2] = col < nd ? A[seg + (int64_t)DV * col + a2] : 0.0; }
        if (l == 15) {
#pragma unroll
            for (int a2 = 0; a2 < DV; ++a2) S.bv[a2] = b[eb0 + m * DV + a2];
#pragma unroll
            for (int q = 0; q < DV * DV; ++q) S.ci[q] = Cinv[(int64_t)(v0 + m) * (DV * DV) + q];
        }
    };
    auto reduce = [&](uint32_t vb, const Step& S) {
        const uint32_t v = vb + gsub; const bool live = v < v1;
        double acc[DV];
#pragma unroll
        for (int a2 = 0; a2 < DV; ++a2) acc[a2] = 0.0;
#pragma unroll
        for (int k = 0; k < MAXC; ++k)
#pragma unroll
            for (int a2 = 0; a2 < DV; ++a2) acc[a2] = fma(S.a[k][a2], xw[k], acc[a2]);
#pragma unroll
        for (int a2 = 0; a2 < DV; ++a2) { double t = acc[a2]; t = row_shr_add<0x111>(t); t = row_shr_add<0x112>(t); t = row_shr_add<0x114>(t); t = row_shr_add<0x118>(t); acc[a2] = t; }
        if (live && l == 15) {
            double r[DV];
#pragma unroll
            for (int a2 = 0; a2 < DV; ++a2) { r[a2] = S.bv[a2] - acc[a2]; ts[(v - v0) * DV + a2] = acc[a2]; }
#pragma unroll
            for (int i = 0; i < DV; ++i) { double t = 0;
#pragma unroll
                for (int j = 0; j < DV; ++j) t = fma(S.ci[i + DV * j], r[j], t);
                xs[(v - v0) * DV + i] = -t; }
        }
    };
    Step S0, S1;
    load(v0, S0);
#pragma unroll 1
    for (uint32_t vb = v0; vb < v1; vb += 8) {
        if (vb + 4 < v1) load(vb + 4, S1);
        reduce(vb, S0);
        if (vb + 4 < v1) { if (vb + 8 < v1) load(vb + 8, S0); reduce(vb + 4, S1); }
    }
    __syncthreads();
    const uint32_t nout = (v1 - v0) * DV;                     // x and tE of the supernode's members are contiguous
    for (uint32_t i = lane; i < nout; i += 64) { x[eb0 + i] = xs[i]; tE[(int64_t)v0 * DV + i] = ts[i]; }
    if (rt.on) {
#pragma unroll
        for (int h = 0; h < 2; ++h) { const uint32_t m = lane + 64 * h; if (m >= v1 - v0) continue;
#pragma unroll
            for (int a2 = 0; a2 < DV; ++a2) rt.vto[ro[h] + a2] = rv[h][a2] + xs[m * DV + a2]; }
    }
}

// Fast path of the elimination for supernodes whose members (a) have the compile-time block size DV and (b) store
// their off-diagonal blocks back to back in their own block row, in reduced-column order, followed by the diagonal
// block -- the layout every bundle-adjustment point row has (src/BlockSparseMatrix.jl:37-44).
// One 256-thread workgroup per supernode.  Per member v: thread c <= nd owns column c of [E | b] -- it loads its DV
// entries straight into registers (a few members ahead), multiplies by (C_v + lambda I)^-1 from schur_cinv_kernel,
// y_c = (C_v + lambda I)^-1 e_c -- and publishes e_c, y_c in LDS,
// component-major, two buffers so that one barrier per member suffices.  The rank-DV update S -= E' Y is register
// tiled: thread t owns a 4x4 tile of pairs (p, q), reads four columns of E and four of Y (2 x DV 32-byte LDS reads)
// and does 16 DV-term dot products; a few more threads take the rhs column.  The tiles are flushed once per supernode
// with HBM atomics (supernodes of different points overlap in S).
constexpr int ELIM_PF = 4;                                    // members in flight per solver thread
constexpr int ELIM_NDP = 76;                                  // columns of [E | b] rounded up to a multiple of 4 (nd + 1 <= 72)
// NC: columns of [E | b] per lane of the solver wave (1: nd + 1 <= 64; 2: up to ELIM_NDP - 4).  TW: tile waves -- two hold
// the 4x4 tiles of nd <= 60 (120 tiles on 128 lanes: the bundle-adjustment point seen by ten cameras), three the rest.
// (EXT: the workgroup's LDS is handed in -- ext, 16-byte aligned, schur_elim_tiled_lds<DV, NC>() doubles -- so that a kernel that runs either this body or
//  another one in a workgroup pays for the larger of the two, not for their sum)
template <int DV, int NC> constexpr int schur_elim_tiled_lds() { constexpr int NDM = NC == 1 ? 63 : ELIM_NDP - 5; return 4 * DV * ELIM_NDP + ELIM_NDP + NDM * (NDM + 1) / 2 + NDM + 2; }
template <int DV, int NC, int TW, bool EXT = false, class LAY = SLayout>
__device__ __forceinline__ void schur_elim_tiled_body(const double* __restrict__ A, const double* __restrict__ b,
                                                      const ElimDesc* __restrict__ desc, const uint32_t* __restrict__ rcflat,
                                                      const double* __restrict__ Cinv, const LAY& L, double* __restrict__ s, uint32_t bidx, double* ext = nullptr) {
    double (*Es)[DV][ELIM_NDP]; double (*Ys)[DV][ELIM_NDP]; uint32_t* rc; uint32_t* rs; double* img;      // rc: reduced column of list column p (MEMORY order); rs: the list columns by ascending reduced column
    if constexpr (EXT) {
        Es = reinterpret_cast<double (*)[DV][ELIM_NDP]>(ext); Ys = reinterpret_cast<double (*)[DV][ELIM_NDP]>(ext + 2 * DV * ELIM_NDP);
        rc = reinterpret_cast<uint32_t*>(ext + 4 * DV * ELIM_NDP); rs = rc + ELIM_NDP; img = ext + 4 * DV * ELIM_NDP + ELIM_NDP;
    } else {
        __shared__ __attribute__((aligned(16))) double Es_[2][DV][ELIM_NDP], Ys_[2][DV][ELIM_NDP];
        __shared__ uint32_t rc_[ELIM_NDP], rs_[ELIM_NDP];
        __shared__ double img_[(NC == 1 ? 63 : ELIM_NDP - 5) * ((NC == 1 ? 63 : ELIM_NDP - 5) + 1) / 2 + (NC == 1 ? 63 : ELIM_NDP - 5)];
        Es = Es_; Ys = Ys_; rc = rc_; rs = rs_; img = img_;
    }
    const int tid = threadIdx.x; constexpr int NT = 64 * (1 + TW);
    const ElimDesc d = desc[bidx];                       // uniform: one scalar load (the run's structure is identical for all members)
    const uint32_t v0 = d.v0, v1 = d.v0 + d.nmem; const int nd = (int)d.nd;
    for (int c2 = tid; c2 < nd; c2 += NT) { rc[c2] = rcflat[d.rc_off + c2]; rs[c2] = rcflat[d.rc_off + nd + c2]; }
    for (int i = tid; i < 2 * DV * ELIM_NDP; i += NT) { (&Es[0][0][0])[i] = 0.0; (&Ys[0][0][0])[i] = 0.0; }
    __syncthreads();
    // this thread's tile: t < ntile -> (tp, tq), tq <= tp, pairs (4 tp + i, 4 tq + j).  (The rhs column E' y_b is summed by the
    // solver wave, which has every column of E in registers: the tile waves then carry no half-empty tiles.)
    const int T = (nd + 3) >> 2, ntile = T * (T + 1) / 2;
    // (tiles live on waves 1-3: wave 0 is the solver and runs one member ahead of them)
    const int tt = tid - 64;
    int tp = 0, tq = 0; const bool has_tile = tt >= 0 && tt < ntile;
    if (has_tile) { tp = (int)((sqrt(8.0 * tt + 1.0) - 1.0) * 0.5); while (tp * (tp + 1) / 2 > tt) --tp; while ((tp + 1) * (tp + 2) / 2 <= tt) ++tp; tq = tt - tp * (tp + 1) / 2; }
    double acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.0;
    // (the members of a supernode are consecutive block rows: constant stride in A.data and in b, nlls_structure.cpp)
    const int64_t dg0 = d.dg0, dstride = (int64_t)DV * nd + DV * DV; const uint32_t eb0 = d.eb0;
    auto member_barrier = [] { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };   // LDS only: loads stay in flight
    constexpr int NDMAX = NC == 1 ? 63 : ELIM_NDP - 5;
    double* const irhs = img + NDMAX * (NDMAX + 1) / 2; double* const rhs_out = irhs;   // (the flush image: used after the member loop)
    if (tid < 64) {
        // ---- solver wave.  Software pipeline: registers hold the column and the inverse diagonal block (schur_cinv_kernel)
        // of the next ELIM_PF members (HBM latency is a multiple of a member's processing time).  What keeps the pipeline
        // alive in the compiled code: (a) every load is unconditional -- lanes beyond the last column and steps beyond the
        // last member re-load a valid address -- so that a load writes the register it is consumed from and its wait sits at
        // the use, one round later (a conditional load becomes a copy plus vmcnt(0) at the end of the round); (b) the
        // inverse, although the same for every lane, does NOT come through scalar loads: they share lgkmcnt with the LDS
        // traffic and return out of order, so the LDS wait of every member would also wait for the scalar load issued a
        // moment ago for the member four ahead (`vz` hides the uniformity from the compiler).
        uint32_t vz = 0; asm volatile("" : "+v"(vz));
        double en[ELIM_PF][NC][DV], cn[ELIM_PF][DV * DV];
        double racc[NC];                                          // entry tid (+ 64) of the rhs column E' y_b
#pragma unroll
        for (int k = 0; k < NC; ++k) racc[k] = 0.0;
        const int kb = nd >> 6, lb = nd & 63;                     // where the rhs column sits: lane lb, slot kb
        auto issue = [&](uint32_t v, int slot) {
            const uint32_t m = (v < v1 ? v : v1 - 1) - v0;
#pragma unroll
            for (int k = 0; k < NC; ++k) {
                const int col = tid + 64 * k < nd ? tid + 64 * k : nd;
                const double* src = col < nd ? A + (dg0 + (int64_t)m * dstride - (int64_t)DV * nd + (int64_t)DV * col) : b + (eb0 + m * DV);
#pragma unroll
                for (int a2 = 0; a2 < DV; ++a2) en[slot][k][a2] = src[a2];
            }
#pragma unroll
            for (int j = 0; j < DV; ++j)
#pragma unroll
                for (int i = j; i < DV; ++i) cn[slot][i + DV * j] = Cinv[(int64_t)(v0 + m) * (DV * DV) + i + DV * j + vz];   // symmetric: lower triangle
        };
#pragma unroll
        for (int u = 0; u < ELIM_PF; ++u) issue(v0 + u, u);
        int buf = 0;
#pragma unroll 1
        for (uint32_t vb = v0; vb < v1; vb += ELIM_PF) {
#pragma unroll
            for (int u = 0; u < ELIM_PF; ++u) {
                const uint32_t v = vb + u;
                if (v >= v1) break;
                double e[NC][DV], C[DV * DV];
#pragma unroll
                for (int k = 0; k < NC; ++k)
#pragma unroll
                    for (int a2 = 0; a2 < DV; ++a2) e[k][a2] = en[u][k][a2];
#pragma unroll
                for (int j = 0; j < DV; ++j)
#pragma unroll
                    for (int i = j; i < DV; ++i) C[i + DV * j] = cn[u][i + DV * j];
                issue(v + ELIM_PF, u);
                double y[NC][DV];                              // y = (C_v + lambda I)^-1 e
#pragma unroll
                for (int k = 0; k < NC; ++k) {
#pragma unroll
                    for (int i = 0; i < DV; ++i) { double t = 0;
#pragma unroll
                        for (int j = 0; j < DV; ++j) t = fma(i >= j ? C[i + DV * j] : C[j + DV * i], e[k][j], t);
                        y[k][i] = t; }
                    if (tid + 64 * k <= nd) {
#pragma unroll
                        for (int a2 = 0; a2 < DV; ++a2) { Es[buf][a2][tid + 64 * k] = e[k][a2]; Ys[buf][a2][tid + 64 * k] = y[k][a2]; }
                    }
                }
#pragma unroll
                for (int a2 = 0; a2 < DV; ++a2) {              // the rhs column: y_b broadcast from its lane
                    const double ysel = (NC == 2 && kb == 1) ? +y[NC - 1][a2] : +y[0][a2];
                    const double yb = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(ysel), lb), __builtin_amdgcn_readlane(__double2loint(ysel), lb));
#pragma unroll
                    for (int k = 0; k < NC; ++k) racc[k] = fma(e[k][a2], yb, racc[k]);
                }
                member_barrier();                              // member v published; the other buffer is free for v + 1
                buf ^= 1;
            }
        }
#pragma unroll
        for (int k = 0; k < NC; ++k) if (tid + 64 * k < nd) rhs_out[tid + 64 * k] = racc[k];
    } else {
        // ---- tile waves: one barrier per member, then this thread's 4x4 tile of the rank-DV update
        int buf = 0;
#pragma unroll 1
        for (uint32_t v = v0; v < v1; ++v) {
            member_barrier();
            if (has_tile) {
                double ep[DV][4], yq[DV][4];
#pragma unroll
                for (int a2 = 0; a2 < DV; ++a2) {
                    const double4_t ev = *reinterpret_cast<const double4_t*>(&Es[buf][a2][4 * tp]);
                    ep[a2][0] = ev[0]; ep[a2][1] = ev[1]; ep[a2][2] = ev[2]; ep[a2][3] = ev[3];
                    const double4_t yv = *reinterpret_cast<const double4_t*>(&Ys[buf][a2][4 * tq]); yq[a2][0] = yv[0]; yq[a2][1] = yv[1]; yq[a2][2] = yv[2]; yq[a2][3] = yv[3];
                }
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        double t = acc[i][j];
#pragma unroll
                        for (int a2 = 0; a2 < DV; ++a2) t = fma(ep[a2][i], yq[a2][j], t);
                        acc[i][j] = t;
                    }
            }
            buf ^= 1;
        }
    }
    // Flush.  The register tiles go through a packed column-major LDS image of the supernode's lower triangle, so that
    // the lanes of one atomic instruction cover consecutive rows of one column of S -- consecutive addresses when the
    // supernode's columns are consecutive (a camera range) -- instead of one cache line per lane.
    auto colstart = [nd](int q) { return q * nd - q * (q - 1) / 2 - q; };   // + p addresses (p, q), p >= q
    if (has_tile) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int p = 4 * tp + i; if (p >= nd) continue;
#pragma unroll
            for (int j = 0; j < 4; ++j) { const int q = 4 * tq + j; if (q <= p) img[colstart(q) + p] = acc[i][j]; }
        }
    }
    __syncthreads();
    const int wv = tid >> 6, ln = tid & 63;
    // (in ascending REDUCED order -- a permutation of the memory order when the reduced system was re-ordered at upload -- so that the lanes of one
    //  instruction still walk down one column of S: addressed pair by pair as (max, min) in memory order, half of them would land in other columns)
    for (int qs = wv; qs < nd; qs += NT / 64) { const int q = (int)rs[qs];
        for (int ps = qs + ln; ps < nd; ps += 64) { const int p = (int)rs[ps]; atomicAdd(L.at(rc[p], rc[q]), -img[p > q ? colstart(q) + p : colstart(p) + q]); } }
    if (tid < nd) atomicAdd(L.rhs(s, rc[tid]), -irhs[tid]);
}

template <int DV, int NC, int TW, class LAY = SLayout>
__global__ __launch_bounds__(64 * (1 + TW)) void schur_elim_tiled_kernel(const double* __restrict__ A, const double* __restrict__ b,
                                                               const ElimDesc* __restrict__ desc, const uint32_t* __restrict__ rcflat,
                                                               const double* __restrict__ Cinv, LAY L, double* __restrict__ s) {
    schur_elim_tiled_body<DV, NC, TW>(A, b, desc, rcflat, Cinv, L, s, blockIdx.x);
}

// The same elimination for supernodes with nd + 1 <= 64 columns of [E | b], on the matrix cores and WITHOUT LDS traffic or barriers in
// the member loop.  S -= E' (C + lambda I)^-1 E is a rank-DV update per member: on v_mfma_f64_16x16x4_f64 the 16x16 tile (R, C) of it is
// ONE instruction whose A operand is lane (i, k) <- e_{16R+i}[k] and whose B operand is lane (j, k) <- y_{16C+j}[k], y = C^-1 e -- and
// those are exactly the values a lane can load itself: the DV entries of "its" column of the member's block row are contiguous in A.data.
// So every wave is independent: it takes every fourth member of the supernode, loads its operands straight from memory into the
// operand layout (one member ahead), and accumulates the lower tiles in registers; the four waves' tiles meet in the packed LDS image of
// the old kernel and leave with the same coalesced atomics.  The fp64 matrix rate equals the vector rate on this chip -- what this buys
// is the operand delivery: the register-tiled kernel above is bound by its LDS reads (2 x DV 32-byte reads per lane and member, bank
// conflicts included) and by one barrier per member.  The right-hand side rides along as column nd (row nd of the lower triangle).
constexpr int ELIM_MFMA_NW = 4;                              // waves per supernode, each taking every fourth member (3 and 6 measured slower: 72 and 93 us against 65)
template <int DV, class LAY = SLayout>
__device__ __forceinline__ void schur_elim_mfma_body(const double* __restrict__ A, const double* __restrict__ b,
                                                     const ElimDesc* __restrict__ desc, const uint32_t* __restrict__ rcflat,
                                                     const double* __restrict__ Cinv, const LAY& L, double* __restrict__ s, uint32_t bidx) {
    constexpr int NDMAX = 63, NW = ELIM_MFMA_NW, NTH = 64 * NW;
    __shared__ uint32_t rc[64], rs[64];                       // rc: reduced column of list column p (MEMORY order); rs: the list columns by ascending reduced column
    __shared__ double img[NDMAX * (NDMAX + 1) / 2 + NDMAX];
    double* const irhs = img + NDMAX * (NDMAX + 1) / 2;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), li = lane & 15, lk = lane >> 4;
    const ElimDesc d = desc[bidx];                       // uniform: one scalar load
    const uint32_t v0 = d.v0, nmem = d.nmem; const int nd = (int)d.nd;
    if (tid < nd) { rc[tid] = rcflat[d.rc_off + tid]; rs[tid] = rcflat[d.rc_off + nd + tid]; }               // (nd <= 63 < NTH)
    for (int i = tid; i < nd * (nd + 1) / 2; i += NTH) img[i] = 0.0;
    if (tid < NDMAX) irhs[tid] = 0.0;
    const int T16 = (nd + 1 + 15) >> 4;                       // tile rows of [E | b] in use (<= 4)
    const int64_t dg0 = d.dg0, dstride = (int64_t)DV * nd + DV * DV; const uint32_t eb0 = d.eb0;
    // this lane's column in every tile row: < nd a column of E, == nd the right-hand side, beyond: clamped to nd and masked out.
    // Lane (li, lk) loads ONE double per tile row and member -- e_{16r+li}[lk], the operand value itself -- and one entry of the inverse.
    constexpr int PF = 2;                                     // members in flight per wave (HBM latency is several members' worth of MFMA time)
    const bool kslot = lk < DV;                               // the fourth k-slot of the instruction stays zero for DV = 3
    const int kk = kslot ? lk : 0;
    const uint32_t nall = nmem;
    const double* ebase[4]; bool live[4], isb[4];               // member m's operand value sits at ebase[r] + (a column of E: offE(m); the right-hand side: offB(m))
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int col = 16 * r + li; live[r] = col <= nd && kslot; isb[r] = col >= nd;
        ebase[r] = isb[r] ? b + eb0 + kk : A + dg0 + (int64_t)DV * col - (int64_t)DV * nd + kk;
    }
    double4_t acc[10];
#pragma unroll
    for (int t = 0; t < 10; ++t) acc[t] = double4_t{0.0, 0.0, 0.0, 0.0};
    // (the member loop is compiled once per number of tile rows: with T16 a run-time value every instruction sat behind its own branch)
    auto members = [&](auto T16c) {
        constexpr int TR = decltype(T16c)::value;
        double en[PF][TR], cn[PF][DV];
        auto issue = [&](uint32_t m, int slot) {              // unconditional, clamped loads (a predicated load becomes copy + vmcnt(0))
            const uint32_t mm = m < nall ? m : nall - 1;
            const int64_t offE = (int64_t)mm * dstride, offB = (int64_t)mm * DV; const uint32_t vi = v0 + mm;      // (uniform) the member's row / right-hand side relative to the supernode's first member, its inverse block
#pragma unroll
            for (int r = 0; r < TR; ++r) en[slot][r] = ebase[r][isb[r] ? offB : offE];
#pragma unroll
            for (int j = 0; j < DV; ++j) cn[slot][j] = Cinv[(int64_t)vi * (DV * DV) + j + DV * kk];   // row lk of the (symmetric) inverse
        };
#pragma unroll
        for (int u = 0; u < PF; ++u) issue(wave + NW * u, u);
#pragma unroll 1
        for (uint32_t mb = wave; mb < nall; mb += NW * PF) {
#pragma unroll
            for (int u = 0; u < PF; ++u) {
                const uint32_t m = mb + NW * u;
                if (m >= nall) break;
                double aop[TR], bop[TR], c[DV];
#pragma unroll
                for (int r = 0; r < TR; ++r) aop[r] = live[r] ? en[u][r] : 0.0;
#pragma unroll
                for (int j = 0; j < DV; ++j) c[j] = kslot ? cn[u][j] : 0.0;
                issue(m + NW * PF, u);
                // y_i[k] = sum_j Cinv[k][j] e_i[j]: the three components of column i sit in the lanes (i, 0..2) -- fetched through the LDS
                // crossbar (ds_bpermute: no LDS memory, no bank conflicts), not recomputed on the matrix cores (four more instructions
                // per member on the pipe that bounds this loop)
#pragma unroll
                for (int r = 0; r < TR; ++r) {
                    double y = 0.0;
#pragma unroll
                    for (int j = 0; j < DV; ++j) {
                        const int src = 4 * (16 * j + li);
                        const double ej = __hiloint2double(__builtin_amdgcn_ds_bpermute(src, __double2hiint(aop[r])), __builtin_amdgcn_ds_bpermute(src, __double2loint(aop[r])));
                        y = fma(c[j], ej, y);
                    }
                    bop[r] = y;
                }
#pragma unroll
                for (int R = 0; R < TR; ++R)
#pragma unroll
                    for (int C = 0; C <= R; ++C) acc[R * (R + 1) / 2 + C] = __builtin_amdgcn_mfma_f64_16x16x4f64(aop[R], bop[C], acc[R * (R + 1) / 2 + C], 0, 0, 0);
            }
        }
    };
    if (T16 == 4) members(std::integral_constant<int, 4>{});
    else if (T16 == 3) members(std::integral_constant<int, 3>{});
    else if (T16 == 2) members(std::integral_constant<int, 2>{});
    else members(std::integral_constant<int, 1>{});
    __syncthreads();                                           // rc and the zeroed image are in place
    // the four waves' tiles meet in the packed column-major image of the lower triangle (register v of lane (li, lk) = entry
    // (row lk + 4 v, column li) of its tile); row nd of the triangle is the right-hand side
    auto colstart = [nd](int q) { return q * nd - q * (q - 1) / 2 - q; };   // + p addresses (p, q), p >= q
#pragma unroll
    for (int R = 0; R < 4; ++R)
#pragma unroll
        for (int C = 0; C <= R; ++C) {
            if (R >= T16) continue;
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int pp = 16 * R + lk + 4 * v, q = 16 * C + li; const double val = acc[R * (R + 1) / 2 + C][v];
                if (pp < nd && q <= pp) atomicAdd(&img[colstart(q) + pp], val);
                else if (pp == nd && q < nd) atomicAdd(&irhs[q], val);
            }
        }
    __syncthreads();
    // (in ascending REDUCED order, see schur_elim_tiled_body: one column of S per instruction whatever the memory order of the columns)
    for (int qs = wave; qs < nd; qs += NW) { const int q = (int)rs[qs];
        for (int ps = qs + lane; ps < nd; ps += 64) { const int pp = (int)rs[ps]; atomicAdd(L.at(rc[pp], rc[q]), -img[pp > q ? colstart(q) + pp : colstart(pp) + q]); } }
    if (tid < nd) atomicAdd(L.rhs(s, rc[tid]), -irhs[tid]);
}

template <int DV, class LAY = SLayout>
__global__ __launch_bounds__(64 * ELIM_MFMA_NW) __attribute__((amdgpu_waves_per_eu(3, 3))) void schur_elim_mfma_kernel(const double* __restrict__ A, const double* __restrict__ b,
                                                              const ElimDesc* __restrict__ desc, const uint32_t* __restrict__ rcflat,
                                                              const double* __restrict__ Cinv, LAY L, double* __restrict__ s) {
    schur_elim_mfma_body<DV>(A, b, desc, rcflat, Cinv, L, s, blockIdx.x);
}
// Both kinds of fast supernode in ONE launch: the narrow ones (matrix-core body) first, the wide ones (register-tiled body, NC = 2)
// behind them.  The narrow supernodes need 1.3 rounds of the chip's wave slots; in a launch of their own the second round leaves most
// CUs idle, and the wide supernodes -- 1-2 members each, all fixed cost -- then wait for it to end.  Here they fill those slots.
template <int DV, class LAY = SLayout>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void schur_elim_fused_kernel(const double* __restrict__ A, const double* __restrict__ b,
                                                              const ElimDesc* __restrict__ desc, const uint32_t* __restrict__ rcflat,
                                                              const double* __restrict__ Cinv, LAY L, double* __restrict__ s, uint32_t nnarrow) {
    static_assert(ELIM_MFMA_NW == 4, "both bodies run in 256-thread workgroups");
    if (blockIdx.x < nnarrow) schur_elim_mfma_body<DV>(A, b, desc, rcflat, Cinv, L, s, blockIdx.x);
    else schur_elim_tiled_body<DV, 2, 3>(A, b, desc, rcflat, Cinv, L, s, blockIdx.x);
}
// ... and with the two small launches that used to stand in front of it folded in (4 + 8 us of launch latency per solve):
//  * every supernode's workgroup inverts the diagonal blocks of ITS members first ((C_v + lambda I)^-1: one lane per member, what schur_cinv_kernel
//    does), stores them for the back-substitution and reads them back itself behind a workgroup barrier;
//  * the workgroups behind the supernodes ADD the reduced-reduced blocks (+ lambda on their diagonals) and the reduced right-hand side into
//    [S | s] -- the storage is zero when the launch starts (the previous solve's back-substitution leaves it so), and sums commute with the
//    supernodes' atomic adds, so the order inside the launch does not matter.
template <int DV, class LAY = SLayout>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void schur_elim_all_kernel(const double* __restrict__ A, const double* __restrict__ b,
                                                              const ElimDesc* __restrict__ desc, const uint32_t* __restrict__ rcflat,
                                                              double* __restrict__ Cinv, LAY L, double* __restrict__ s, uint32_t nnarrow, PrepArgs pa) {
    if (blockIdx.x == 0 && threadIdx.x == 0) time_stamp(pa.stamps, 0);
    if (blockIdx.x >= pa.nfast) { schur_prep_roles(A, b, L, s, pa, (int)(blockIdx.x - pa.nfast)); return; }
    {   // the members' inverse diagonal blocks (schur_cinv_kernel's arithmetic, same bits)
        const ElimDesc d = desc[blockIdx.x];
        const int64_t dstride = (int64_t)DV * d.nd + DV * DV;
        for (uint32_t m = threadIdx.x; m < d.nmem; m += 256) {
            const double* Cg = A + d.dg0 + (int64_t)m * dstride; double C[DV * DV];
            const int64_t vidx = (int64_t)(d.v0 + m);
#pragma unroll
            for (int j = 0; j < DV; ++j)
#pragma unroll
                for (int i = j; i < DV; ++i) C[i + DV * j] = Cg[i + DV * j];
#pragma unroll
            for (int j = 0; j < DV; ++j) {
                double dd = C[j + DV * j] + pa.lambda;
#pragma unroll
                for (int k = 0; k < j; ++k) dd -= C[j + DV * k] * C[j + DV * k] * C[k + DV * k];
                if (dd == 0.0 || dd != dd) { atomicCAS(pa.status, 0, 1); dd = 1.0; }
                C[j + DV * j] = dd;
#pragma unroll
                for (int i = j + 1; i < DV; ++i) { double t = C[i + DV * j];
#pragma unroll
                    for (int k = 0; k < j; ++k) t -= C[i + DV * k] * C[j + DV * k] * C[k + DV * k];
                    C[i + DV * j] = t / dd; }
            }
#pragma unroll
            for (int c2 = 0; c2 < DV; ++c2) {
                double y[DV];
#pragma unroll
                for (int i = 0; i < DV; ++i) { double t = (i == c2) ? 1.0 : 0.0;
#pragma unroll
                    for (int k = 0; k < i; ++k) t -= C[i + DV * k] * y[k]; y[i] = t; }
#pragma unroll
                for (int i = 0; i < DV; ++i) y[i] /= C[i + DV * i];
#pragma unroll
                for (int i = DV - 1; i >= 0; --i) { double t = y[i];
#pragma unroll
                    for (int k = i + 1; k < DV; ++k) t -= C[k + DV * i] * y[k]; y[i] = t; }
#pragma unroll
                for (int i = 0; i < DV; ++i) Cinv[vidx * (DV * DV) + i + DV * c2] = y[i];
            }
        }
        __threadfence_block();          // the workgroup's own stores, then its own loads of them (workgroup scope)
        __syncthreads();
    }
    if (blockIdx.x < nnarrow) schur_elim_mfma_body<DV>(A, b, desc, rcflat, Cinv, L, s, blockIdx.x);
    else schur_elim_tiled_body<DV, 2, 3>(A, b, desc, rcflat, Cinv, L, s, blockIdx.x);
}

template <int DV, int NC, int TW>
__global__ __launch_bounds__(64 * (1 + TW)) void schur_elim_slab_kernel(const double* __restrict__ A, const double* __restrict__ b,
                                                               const int64_t* __restrict__ eptr, const SchurNbr* __restrict__ enbr,
                                                               const int64_t* __restrict__ ediag, const uint32_t* __restrict__ eboff,
                                                               const uint32_t* __restrict__ egroup, const uint32_t* __restrict__ glist,
                                                               const double* __restrict__ Cinv, double* __restrict__ slab, const uint32_t* __restrict__ slab_off) {
    __shared__ __attribute__((aligned(16))) double Es[2][DV][ELIM_NDP], Ys[2][DV][ELIM_NDP];
    __shared__ uint32_t rc[ELIM_NDP];
    const int tid = threadIdx.x; constexpr int NT = 64 * (1 + TW);
    const uint32_t g = glist[blockIdx.x];
    const uint32_t v0 = egroup[g], v1 = egroup[g + 1];
    // structure of the run (identical for all members): reduced column of every E column
    const int64_t p0 = eptr[v0]; const int nnb = (int)(eptr[v0 + 1] - p0);
    int nd = 0;
    __shared__ uint8_t cblk[ELIM_NDP], coff[ELIM_NDP]; __shared__ uint16_t bdim[16], poff[16 * 17 / 2 + 1];
    for (int p = 0; p < nnb; ++p) { const SchurNbr nb = enbr[p0 + p]; for (int c2 = tid; c2 < nb.dim; c2 += NT) { cblk[nd + c2] = (uint8_t)p; coff[nd + c2] = (uint8_t)c2; } if (tid == 0) bdim[p] = nb.dim; nd += nb.dim; }
    if (tid == 0) { int acc0 = 0; for (int A2 = 0; A2 < nnb; ++A2) { const int dA = enbr[p0 + A2].dim; for (int B2 = 0; B2 <= A2; ++B2) { poff[A2 * (A2 + 1) / 2 + B2] = (uint16_t)acc0; acc0 += dA * enbr[p0 + B2].dim; } } poff[nnb * (nnb + 1) / 2] = (uint16_t)acc0; }
    for (int i = tid; i < 2 * DV * ELIM_NDP; i += NT) { (&Es[0][0][0])[i] = 0.0; (&Ys[0][0][0])[i] = 0.0; }
    __syncthreads();
    // this thread's tile: t < ntile -> (tp, tq), tq <= tp, pairs (4 tp + i, 4 tq + j).  (The rhs column E' y_b is summed by the
    // solver wave, which has every column of E in registers: the tile waves then carry no half-empty tiles.)
    const int T = (nd + 3) >> 2, ntile = T * (T + 1) / 2;
    // (tiles live on waves 1-3: wave 0 is the solver and runs one member ahead of them)
    const int tt = tid - 64;
    int tp = 0, tq = 0; const bool has_tile = tt >= 0 && tt < ntile;
    if (has_tile) { tp = (int)((sqrt(8.0 * tt + 1.0) - 1.0) * 0.5); while (tp * (tp + 1) / 2 > tt) --tp; while ((tp + 1) * (tp + 2) / 2 <= tt) ++tp; tq = tt - tp * (tp + 1) / 2; }
    double acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.0;
    // (the members of a supernode are consecutive block rows: constant stride in A.data and in b, nlls_structure.cpp)
    const int64_t dg0 = ediag[v0], dstride = (int64_t)DV * nd + DV * DV; const uint32_t eb0 = eboff[v0];
    auto member_barrier = [] { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };   // LDS only: loads stay in flight
    constexpr int NDMAX = NC == 1 ? 63 : ELIM_NDP - 5;
    __shared__ double img[NDMAX * (NDMAX + 1) / 2 + 16 * NDMAX + NDMAX];   // one column-major block per pair of neighbour blocks (diagonal pairs: lower triangle filled), then the rhs
    double* const rhs_out = img + poff[nnb * (nnb + 1) / 2];
    if (tid < 64) {
        // ---- solver wave.  Software pipeline: registers hold the column and the inverse diagonal block (schur_cinv_kernel)
        // of the next ELIM_PF members (HBM latency is a multiple of a member's processing time).  What keeps the pipeline
        // alive in the compiled code: (a) every load is unconditional -- lanes beyond the last column and steps beyond the
        // last member re-load a valid address -- so that a load writes the register it is consumed from and its wait sits at
        // the use, one round later (a conditional load becomes a copy plus vmcnt(0) at the end of the round); (b) the
        // inverse, although the same for every lane, does NOT come through scalar loads: they share lgkmcnt with the LDS
        // traffic and return out of order, so the LDS wait of every member would also wait for the scalar load issued a
        // moment ago for the member four ahead (`vz` hides the uniformity from the compiler).
        uint32_t vz = 0; asm volatile("" : "+v"(vz));
        double en[ELIM_PF][NC][DV], cn[ELIM_PF][DV * DV];
        double racc[NC];                                          // entry tid (+ 64) of the rhs column E' y_b
#pragma unroll
        for (int k = 0; k < NC; ++k) racc[k] = 0.0;
        const int kb = nd >> 6, lb = nd & 63;                     // where the rhs column sits: lane lb, slot kb
        auto issue = [&](uint32_t v, int slot) {
            const uint32_t m = (v < v1 ? v : v1 - 1) - v0;
#pragma unroll
            for (int k = 0; k < NC; ++k) {
                const int col = tid + 64 * k < nd ? tid + 64 * k : nd;
                const double* src = col < nd ? A + (dg0 + (int64_t)m * dstride - (int64_t)DV * nd + (int64_t)DV * col) : b + (eb0 + m * DV);
#pragma unroll
                for (int a2 = 0; a2 < DV; ++a2) en[slot][k][a2] = src[a2];
            }
#pragma unroll
            for (int j = 0; j < DV; ++j)
#pragma unroll
                for (int i = j; i < DV; ++i) cn[slot][i + DV * j] = Cinv[(int64_t)(v0 + m) * (DV * DV) + i + DV * j + vz];   // symmetric: lower triangle
        };
#pragma unroll
        for (int u = 0; u < ELIM_PF; ++u) issue(v0 + u, u);
        int buf = 0;
#pragma unroll 1
        for (uint32_t vb = v0; vb < v1; vb += ELIM_PF) {
#pragma unroll
            for (int u = 0; u < ELIM_PF; ++u) {
                const uint32_t v = vb + u;
                if (v >= v1) break;
                double e[NC][DV], C[DV * DV];
#pragma unroll
                for (int k = 0; k < NC; ++k)
#pragma unroll
                    for (int a2 = 0; a2 < DV; ++a2) e[k][a2] = en[u][k][a2];
#pragma unroll
                for (int j = 0; j < DV; ++j)
#pragma unroll
                    for (int i = j; i < DV; ++i) C[i + DV * j] = cn[u][i + DV * j];
                issue(v + ELIM_PF, u);
                double y[NC][DV];                              // y = (C_v + lambda I)^-1 e
#pragma unroll
                for (int k = 0; k < NC; ++k) {
#pragma unroll
                    for (int i = 0; i < DV; ++i) { double t = 0;
#pragma unroll
                        for (int j = 0; j < DV; ++j) t = fma(i >= j ? C[i + DV * j] : C[j + DV * i], e[k][j], t);
                        y[k][i] = t; }
                    if (tid + 64 * k <= nd) {
#pragma unroll
                        for (int a2 = 0; a2 < DV; ++a2) { Es[buf][a2][tid + 64 * k] = e[k][a2]; Ys[buf][a2][tid + 64 * k] = y[k][a2]; }
                    }
                }
#pragma unroll
                for (int a2 = 0; a2 < DV; ++a2) {              // the rhs column: y_b broadcast from its lane
                    const double ysel = (NC == 2 && kb == 1) ? +y[NC - 1][a2] : +y[0][a2];
                    const double yb = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(ysel), lb), __builtin_amdgcn_readlane(__double2loint(ysel), lb));
#pragma unroll
                    for (int k = 0; k < NC; ++k) racc[k] = fma(e[k][a2], yb, racc[k]);
                }
                member_barrier();                              // member v published; the other buffer is free for v + 1
                buf ^= 1;
            }
        }
#pragma unroll
        for (int k = 0; k < NC; ++k) if (tid + 64 * k < nd) rhs_out[tid + 64 * k] = racc[k];
    } else {
        // ---- tile waves: one barrier per member, then this thread's 4x4 tile of the rank-DV update
        int buf = 0;
#pragma unroll 1
        for (uint32_t v = v0; v < v1; ++v) {
            member_barrier();
            if (has_tile) {
                double ep[DV][4], yq[DV][4];
#pragma unroll
                for (int a2 = 0; a2 < DV; ++a2) {
                    const double4_t ev = *reinterpret_cast<const double4_t*>(&Es[buf][a2][4 * tp]);
                    ep[a2][0] = ev[0]; ep[a2][1] = ev[1]; ep[a2][2] = ev[2]; ep[a2][3] = ev[3];
                    const double4_t yv = *reinterpret_cast<const double4_t*>(&Ys[buf][a2][4 * tq]); yq[a2][0] = yv[0]; yq[a2][1] = yv[1]; yq[a2][2] = yv[2]; yq[a2][3] = yv[3];
                }
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        double t = acc[i][j];
#pragma unroll
                        for (int a2 = 0; a2 < DV; ++a2) t = fma(ep[a2][i], yq[a2][j], t);
                        acc[i][j] = t;
                    }
            }
            buf ^= 1;
        }
    }
    // Flush: the register tiles go into the block image in LDS, the image leaves for the supernode's own slab with plain
    // coalesced stores -- no atomics; schur_gather_kernel sums the shares of all supernodes in a fixed order.
    if (has_tile) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int p = 4 * tp + i; if (p >= nd) continue;
            const int A2 = cblk[p], oa = coff[p], dA = bdim[A2];
#pragma unroll
            for (int j = 0; j < 4; ++j) { const int q = 4 * tq + j; if (q <= p) { const int B2 = cblk[q]; img[poff[A2 * (A2 + 1) / 2 + B2] + oa + dA * coff[q]] = acc[i][j]; } }
        }
    }
    __syncthreads();
    double* const out = slab + slab_off[blockIdx.x];
    const int total = poff[nnb * (nnb + 1) / 2] + nd;
    for (int t = tid; t < total; t += NT) out[t] = img[t];
}

// Elimination without a barrier per member: FOUR INDEPENDENT wavefronts per supernode, each taking every fourth member and
// owning a full set of 4x4 register tiles of the supernode's share of S.  A wave loads its members' columns of [E | b] itself
// (a few members ahead, one column per lane -- every byte of the point rows is requested once, by one wave, and with four
// waves per supernode enough bytes are in flight to cover the HBM latency), multiplies by (C_v + lambda I)^-1, stages e_c, y_c
// in its own LDS buffer (LDS operations of one wave execute in order: no barrier between its stores and its loads) and applies
// the rank-DV update to its tiles.  The four partial shares are summed through the block image in LDS in wave order
// (deterministic) and leave for the supernode's slab with plain coalesced stores: one column-major block per pair of neighbour
// blocks, then the rhs.
constexpr int ELIM_NW = 2;
template <int DV, int NC, int TPL>
__global__ __launch_bounds__(64 * ELIM_NW) void schur_elim_wave_kernel(const double* __restrict__ A, const double* __restrict__ b,
                                                              const int64_t* __restrict__ eptr, const SchurNbr* __restrict__ enbr,
                                                              const int64_t* __restrict__ ediag, const uint32_t* __restrict__ eboff,
                                                              const uint32_t* __restrict__ egroup, const uint32_t* __restrict__ glist,
                                                              const double* __restrict__ Cinv, double* __restrict__ slab, const uint32_t* __restrict__ slab_off) {
    constexpr int NT2 = 64 * ELIM_NW;
    __shared__ __attribute__((aligned(16))) double Es[ELIM_NW][DV][ELIM_NDP], Ys[ELIM_NW][DV][ELIM_NDP];      // [wave]: private staging
    __shared__ uint8_t cblk[ELIM_NDP], coff[ELIM_NDP]; __shared__ uint16_t bdim[16], poff[16 * 17 / 2 + 1];
    constexpr int NDMAX = NC == 1 ? 63 : ELIM_NDP - 5;
    __shared__ double img[NDMAX * (NDMAX + 1) / 2 + 16 * NDMAX + NDMAX];
    const int tid = threadIdx.x, wv = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const uint32_t g = glist[blockIdx.x];
    const uint32_t v0 = egroup[g], v1 = egroup[g + 1];
    const int64_t p0 = eptr[v0]; const int nnb = (int)(eptr[v0 + 1] - p0);
    int nd = 0;
    for (int p = 0; p < nnb; ++p) { const SchurNbr nb = enbr[p0 + p]; for (int c2 = tid; c2 < nb.dim; c2 += NT2) { cblk[nd + c2] = (uint8_t)p; coff[nd + c2] = (uint8_t)c2; } if (tid == 0) bdim[p] = nb.dim; nd += nb.dim; }
    if (tid == 0) { int acc0 = 0; for (int A2 = 0; A2 < nnb; ++A2) { const int dA = enbr[p0 + A2].dim; for (int B2 = 0; B2 <= A2; ++B2) { poff[A2 * (A2 + 1) / 2 + B2] = (uint16_t)acc0; acc0 += dA * enbr[p0 + B2].dim; } } poff[nnb * (nnb + 1) / 2] = (uint16_t)acc0; }
    for (int i = tid; i < ELIM_NW * DV * ELIM_NDP; i += NT2) { (&Es[0][0][0])[i] = 0.0; (&Ys[0][0][0])[i] = 0.0; }
    __syncthreads();
    // this lane's tiles: t = lane + 64 k < ntile -> (tp, tq), tq <= tp, pairs (4 tp + i, 4 tq + j).  Consecutive lanes hold
    // consecutive tiles of a tile row, i.e. consecutive 32-byte groups of Y: a 16-byte LDS read per lane then touches every bank
    // twice (lanes l and l + 8 of a 16-lane group are 256 bytes apart) -- unless the lanes with an odd (tq / 8) read the two
    // halves of their group in the opposite order.  Their accumulator columns are then permuted (j ^ 2), which only the flush sees.
    const int T = (nd + 3) >> 2, ntile = T * (T + 1) / 2;
    int tp[TPL], tq[TPL], sw[TPL]; bool has[TPL];
#pragma unroll
    for (int k = 0; k < TPL; ++k) {
        const int tt = lane + 64 * k; has[k] = tt < ntile; tp[k] = 0; tq[k] = 0; sw[k] = 0;
        if (has[k]) { int a2 = (int)((sqrt(8.0 * tt + 1.0) - 1.0) * 0.5); while (a2 * (a2 + 1) / 2 > tt) --a2; while ((a2 + 1) * (a2 + 2) / 2 <= tt) ++a2; tp[k] = a2; tq[k] = tt - a2 * (a2 + 1) / 2; sw[k] = (tq[k] >> 3) & 1; }
    }
    double acc[TPL][4][4];
#pragma unroll
    for (int k = 0; k < TPL; ++k)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[k][i][j] = 0.0;
    const int64_t dg0 = ediag[v0], dstride = (int64_t)DV * nd + DV * DV; const uint32_t eb0 = eboff[v0];
    uint32_t vz = 0; asm volatile("" : "+v"(vz));                // (keeps the inverse's loads vector loads: see the kernel above)
    double en[ELIM_PF][NC][DV], cn[ELIM_PF][DV * DV];
    double racc[NC];
#pragma unroll
    for (int k = 0; k < NC; ++k) racc[k] = 0.0;
    const int kb = nd >> 6, lb = nd & 63;                         // where the rhs column sits: lane lb, slot kb
    const uint32_t nmem = v1 - v0;
    // member index m of this wave's step s2: wv + 4 s2 (clamped: steps behind the last member re-load a valid row)
    auto issue = [&](uint32_t s2, int slot) {
        uint32_t m = wv + ELIM_NW * s2; m = m < nmem ? m : nmem - 1;
#pragma unroll
        for (int k = 0; k < NC; ++k) {
            const int col = lane + 64 * k < nd ? lane + 64 * k : nd;
            const double* src = col < nd ? A + (dg0 + (int64_t)m * dstride - (int64_t)DV * nd + (int64_t)DV * col) : b + (eb0 + m * DV);
#pragma unroll
            for (int a2 = 0; a2 < DV; ++a2) en[slot][k][a2] = src[a2];
        }
#pragma unroll
        for (int j = 0; j < DV; ++j)
#pragma unroll
            for (int i = j; i < DV; ++i) cn[slot][i + DV * j] = Cinv[(int64_t)(v0 + m) * (DV * DV) + i + DV * j + vz];
    };
#pragma unroll
    for (int u = 0; u < ELIM_PF; ++u) issue(u, u);
    double (*const myE)[ELIM_NDP] = Es[wv]; double (*const myY)[ELIM_NDP] = Ys[wv];
    const uint32_t nstep = nmem > (uint32_t)wv ? (nmem - wv + ELIM_NW - 1) / ELIM_NW : 0;
#pragma unroll 1
    for (uint32_t sb = 0; sb < nstep; sb += ELIM_PF) {
#pragma unroll
        for (int u = 0; u < ELIM_PF; ++u) {
            const uint32_t s2 = sb + u;
            if (s2 >= nstep) break;
            double e[NC][DV], C[DV * DV];
#pragma unroll
            for (int k = 0; k < NC; ++k)
#pragma unroll
                for (int a2 = 0; a2 < DV; ++a2) e[k][a2] = en[u][k][a2];
#pragma unroll
            for (int j = 0; j < DV; ++j)
#pragma unroll
                for (int i = j; i < DV; ++i) C[i + DV * j] = cn[u][i + DV * j];
            issue(s2 + ELIM_PF, u);
            double y[NC][DV];
#pragma unroll
            for (int k = 0; k < NC; ++k) {
#pragma unroll
                for (int i = 0; i < DV; ++i) { double t = 0;
#pragma unroll
                    for (int j = 0; j < DV; ++j) t = fma(i >= j ? C[i + DV * j] : C[j + DV * i], e[k][j], t);
                    y[k][i] = t; }
                if (lane + 64 * k <= nd) {
#pragma unroll
                    for (int a2 = 0; a2 < DV; ++a2) { myE[a2][lane + 64 * k] = e[k][a2]; myY[a2][lane + 64 * k] = y[k][a2]; }
                }
            }
#pragma unroll
            for (int a2 = 0; a2 < DV; ++a2) {                  // the rhs column: y_b broadcast from its lane
                const double ysel = (NC == 2 && kb == 1) ? +y[NC - 1][a2] : +y[0][a2];
                const double yb = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(ysel), lb), __builtin_amdgcn_readlane(__double2loint(ysel), lb));
#pragma unroll
                for (int k = 0; k < NC; ++k) racc[k] = fma(e[k][a2], yb, racc[k]);
            }
            asm volatile("" ::: "memory");                          // the loads below stay behind the stores above (same wave, in-order LDS)
#pragma unroll
            for (int k = 0; k < TPL; ++k) {
                double ep[DV][4], yq[DV][4];
#pragma unroll
                for (int a2 = 0; a2 < DV; ++a2) {
                    const double4_t ev = *reinterpret_cast<const double4_t*>(&myE[a2][4 * tp[k]]);
                    ep[a2][0] = ev[0]; ep[a2][1] = ev[1]; ep[a2][2] = ev[2]; ep[a2][3] = ev[3];
                    typedef double double2_t __attribute__((ext_vector_type(2)));
                    const double2_t y0 = *reinterpret_cast<const double2_t*>(&myY[a2][4 * tq[k] + 2 * sw[k]]), y1 = *reinterpret_cast<const double2_t*>(&myY[a2][4 * tq[k] + 2 - 2 * sw[k]]);
                    yq[a2][0] = y0[0]; yq[a2][1] = y0[1]; yq[a2][2] = y1[0]; yq[a2][3] = y1[1];
                }
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        double t = acc[k][i][j];
#pragma unroll
                        for (int a2 = 0; a2 < DV; ++a2) t = fma(ep[a2][i], yq[a2][j], t);
                        acc[k][i][j] = t;
                    }
            }
            asm volatile("" ::: "memory");                          // ... and the next member's stores behind these loads
        }
    }
    // the four partial shares meet in the block image, in wave order
    const int roff = poff[nnb * (nnb + 1) / 2];
    for (int w = 0; w < ELIM_NW; ++w) {
        if (wv == w) {
#pragma unroll
            for (int k = 0; k < TPL; ++k) if (has[k]) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int p = 4 * tp[k] + i; if (p >= nd) continue;
                    const int A2 = cblk[p], oa = coff[p], dA = bdim[A2];
#pragma unroll
                    for (int j = 0; j < 4; ++j) { const int q = 4 * tq[k] + (j ^ (2 * sw[k])); if (q <= p) { const int B2 = cblk[q]; double* d = &img[poff[A2 * (A2 + 1) / 2 + B2] + oa + dA * coff[q]]; *d = w == 0 ? acc[k][i][j] : *d + acc[k][i][j]; } }
                }
            }
#pragma unroll
            for (int k = 0; k < NC; ++k) if (lane + 64 * k < nd) { double* d = &img[roff + lane + 64 * k]; *d = w == 0 ? racc[k] : *d + racc[k]; }
        }
        __syncthreads();
    }
    double* const out = slab + slab_off[blockIdx.x];
    const int total = roff + nd;
    for (int t = tid; t < total; t += NT2) out[t] = img[t];
}

// x_v = C_v^-1 (b_v - E_v x_R), stored negated (negate!, src/iterators.jl:3)
__global__ __launch_bounds__(64) void schur_backsub_kernel(const double* __restrict__ A, const double* __restrict__ b,
                                                           const int64_t* __restrict__ eptr, const SchurNbr* __restrict__ enbr,
                                                           const int64_t* __restrict__ ediag, const uint32_t* __restrict__ eboff,
                                                           const uint16_t* __restrict__ edim, const uint32_t* __restrict__ vlist, double lambda, int maxdv,
                                                           const double* __restrict__ xr, double* __restrict__ x) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int v = vlist ? (int)vlist[blockIdx.x] : (int)blockIdx.x, lane = threadIdx.x; const int dv = edim[v];
    double* C = sm; double* rhs = C + maxdv * maxdv;
    for (int e = lane; e < dv * dv; e += 64) { const int i = e % dv, j = e / dv; C[e] = A[ediag[v] + e] + (i == j ? lambda : 0.0); }
    // rhs[a] = b[a] - sum_p E[a,p] xr[p]: lanes over the neighbour blocks' elements, summed with LDS atomics
    for (int a = lane; a < dv; a += 64) rhs[a] = b[eboff[v] + a];
    __syncthreads();
    for (int64_t p = eptr[v]; p < eptr[v + 1]; ++p) {
        const SchurNbr nb = enbr[p]; const int du = nb.dim;
        for (int e = lane; e < dv * du; e += 64) {
            int a, c2; if (!nb.trans) { a = e % dv; c2 = e / dv; } else { c2 = e % du; a = e / du; }
            atomicAdd(&rhs[a], -A[nb.off + e] * xr[nb.rcol + c2]);
        }
    }
    __syncthreads();
    if (lane == 0) {
        for (int j = 0; j < dv; ++j) {
            double d = C[j + dv * j]; for (int k = 0; k < j; ++k) d -= C[j + dv * k] * C[j + dv * k] * C[k + dv * k];
            if (d == 0.0 || d != d) d = 1.0;
            C[j + dv * j] = d;
            for (int i = j + 1; i < dv; ++i) { double t = C[i + dv * j]; for (int k = 0; k < j; ++k) t -= C[i + dv * k] * C[j + dv * k] * C[k + dv * k]; C[i + dv * j] = t / d; }
        }
        for (int i = 0; i < dv; ++i) { double t = rhs[i]; for (int k = 0; k < i; ++k) t -= C[i + dv * k] * rhs[k]; rhs[i] = t; }
        for (int i = 0; i < dv; ++i) rhs[i] /= C[i + dv * i];
        for (int i = dv - 1; i >= 0; --i) { double t = rhs[i]; for (int k = i + 1; k < dv; ++k) t -= C[k + dv * i] * rhs[k]; rhs[i] = t; }
        for (int i = 0; i < dv; ++i) x[eboff[v] + i] = -rhs[i];
    }
}
__global__ void scatter_reduced_kernel(const double* __restrict__ xr, const uint32_t* __restrict__ red_boff, int n, double* __restrict__ x, int write) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) x[red_boff[i]] = write ? -xr[i] : 0.0;   // under sharding only rank 0 contributes the reduced part to the sum of x
}

// ---------------------------------------------------------------------------------------------------
// small systems (n < 64): Cholesky, else LU with partial pivoting (the reference falls back from
// cholesky to qr, src/linearsolver.jl:20-26; any exact solver of a nonsingular system is equivalent)
// ---------------------------------------------------------------------------------------------------
// One wavefront.  M (64 x 65 doubles of LDS) = the symmetric matrix read from the lower triangle of S (leading dimension ld) + lambda on the diagonal,
// rhs = bsrc[map ? map[i] : i]; on return (all lanes, behind a barrier) rhs holds the solution.
NLLS_DEV void small_solve_lds(const double* __restrict__ S, int ld, double lambda, const double* __restrict__ bsrc, const uint32_t* __restrict__ map, int n,
                              double* M, double* rhs, int* piv, int* ok, int* __restrict__ status) {
    const int t = threadIdx.x;
    auto load = [&]() {
        for (int e = t; e < n * n; e += 64) { const int i = e % n, j = e / n; M[i + 65 * j] = ((i >= j) ? S[(size_t)i + (size_t)ld * j] : S[(size_t)j + (size_t)ld * i]) + (i == j ? lambda : 0.0); }
        if (t < n) rhs[t] = bsrc[map ? map[t] : (uint32_t)t];
    };
    load();
    if (t == 0) *ok = 1;
    __syncthreads();
    // Cholesky (right-looking), thread = row
    for (int j = 0; j < n; ++j) {
        const double d = M[j + 65 * j];
        if (!(d > 0)) { if (t == 0) *ok = 0; }
        __syncthreads();
        if (!*ok) break;
        const double sd = sqrt(d);
        double lij = 0;
        if (t > j && t < n) { lij = M[t + 65 * j] / sd; }
        __syncthreads();
        if (t == j) M[j + 65 * j] = sd;
        if (t > j && t < n) M[t + 65 * j] = lij;
        __syncthreads();
        if (t > j && t < n) for (int c2 = j + 1; c2 <= t; ++c2) M[t + 65 * c2] -= lij * M[c2 + 65 * j];
        __syncthreads();
    }
    if (*ok) {
        if (t == 0) {
            for (int i = 0; i < n; ++i) { double v = rhs[i]; for (int k = 0; k < i; ++k) v -= M[i + 65 * k] * rhs[k]; rhs[i] = v / M[i + 65 * i]; }
            for (int i = n - 1; i >= 0; --i) { double v = rhs[i]; for (int k = i + 1; k < n; ++k) v -= M[k + 65 * i] * rhs[k]; rhs[i] = v / M[i + 65 * i]; }
        }
        __syncthreads();
        return;
    }
    // LU with partial pivoting on a fresh symmetric copy
    __syncthreads();
    load();
    __syncthreads();
    for (int j = 0; j < n; ++j) {
        if (t == 0) { int p = j; double best = fabs(M[j + 65 * j]); for (int i = j + 1; i < n; ++i) { double a = fabs(M[i + 65 * j]); if (a > best) { best = a; p = i; } } *piv = p; if (best == 0.0) atomicCAS(status, 0, 2); }
        __syncthreads();
        const int p = *piv;
        if (p != j) { if (t < n) { double a = M[j + 65 * t]; M[j + 65 * t] = M[p + 65 * t]; M[p + 65 * t] = a; } if (t == 0) { double a = rhs[j]; rhs[j] = rhs[p]; rhs[p] = a; } }
        __syncthreads();
        double f = 0;
        if (t > j && t < n) { f = M[t + 65 * j] / M[j + 65 * j]; }
        __syncthreads();
        if (t > j && t < n) { for (int c2 = j + 1; c2 < n; ++c2) M[t + 65 * c2] -= f * M[j + 65 * c2]; rhs[t] -= f * rhs[j]; }
        __syncthreads();
    }
    if (t == 0) for (int i = n - 1; i >= 0; --i) { double v = rhs[i]; for (int k = i + 1; k < n; ++k) v -= M[i + 65 * k] * rhs[k]; rhs[i] = v / M[i + 65 * i]; }
    __syncthreads();
}
__global__ __launch_bounds__(64) void small_solve_kernel(const double* __restrict__ S, double* __restrict__ s, int n, int npad, int* __restrict__ status) {
    __shared__ double M[64 * 65]; __shared__ double rhs[64]; __shared__ int piv; __shared__ int ok;
    small_solve_lds(S, npad, 0.0, s, nullptr, n, M, rhs, &piv, &ok, status);
    if ((int)threadIdx.x < n) s[threadIdx.x] = rhs[threadIdx.x];
}
// The LM trial of the small dense system (nlls_ctx::tiny_dense) up to its cost sweep, in ONE single-wavefront launch: uniformscaling!(H, lambda), solve!, negate!
// (src/iterators.jl:149-155), the step's statistics (max|x|, x'x, x'Hx, g'x: what nlls_lm_trial caches for nlls_quadform / nlls_step_stats) and, for problems of at
// most TINY_RETRACT_MAX variables, update!(to, from, x) (src/linearsystem.jl:206-213).  A is the full symmetric undamped matrix (leading dimension n).
constexpr int TINY_RETRACT_MAX = 4096;
__global__ __launch_bounds__(64) void tiny_dense_trial_kernel(const double* __restrict__ A, const double* __restrict__ b, const uint32_t* __restrict__ red_boff, double lambda, int n,
                                                              double* __restrict__ x, double* __restrict__ scalars, int* __restrict__ status,
                                                              const int32_t* __restrict__ vkind, const int32_t* __restrict__ vdim, const uint32_t* __restrict__ voff, const uint32_t* __restrict__ vboff,
                                                              int64_t nretract, const double* __restrict__ vfrom, double* __restrict__ vto) {
    __shared__ double M[64 * 65]; __shared__ double rhs[64]; __shared__ double xs[64]; __shared__ int piv; __shared__ int ok;
    const int t = threadIdx.x;
    if (t < 5) status[t] = 0;
    xs[t] = 0.0;
    __syncthreads();
    small_solve_lds(A, n, lambda, b, red_boff, n, M, rhs, &piv, &ok, status);
    const uint32_t bo = t < n ? red_boff[t] : 0u;
    const double xv = t < n ? -rhs[t] : 0.0, bt = t < n ? b[bo] : 0.0;
    if (t < n) { x[bo] = xv; xs[bo] = xv; }
    __syncthreads();
    double Ax = 0;                                                   // row bo of A times x (A is symmetric and stored in full)
    if (t < n) for (int j = 0; j < n; ++j) Ax += A[(size_t)bo + (size_t)n * j] * xs[j];
    const bool anynan = __any(is_nan_bits(xv));
    const double mx = wave_max(fabs(xv)), xx = wave_sum(xv * xv), gx = wave_sum(bt * xv), xAx = wave_sum(xv * Ax);
    if (t == 0) {
        scalars[1] = anynan ? __longlong_as_double(0x7ff8000000000000LL) : mx; scalars[2] = xx; scalars[4] = xAx + lambda * xx; scalars[5] = gx;
        scalars[8] = xAx; scalars[9] = xx; scalars[10] = (double)status[0];
    }
    for (int64_t i = t; i < nretract; i += 64) retract_one(vkind, vdim, voff, vboff, i, vfrom, xs, vto);
}

// ---------------------------------------------------------------------------------------------------
// blocked Cholesky of the (bordered) reduced system: S col-major, ld = npad, lower triangle
// ---------------------------------------------------------------------------------------------------
// diagonal block: unblocked right-looking LDL' in LDS, 256 threads.  On exit the block holds the unit-lower
// L below the diagonal and D on it.  (LDL' rather than LL': the damped reduced system of a gauge-free
// bundle adjustment is only barely definite; like the reference's LDLFactorizations, no pivot sign is required.)
__global__ __launch_bounds__(256) void ldlt_diag_kernel(double* __restrict__ S, int npad, int k, int* __restrict__ status) {
    __shared__ double M[NB * (NB + 1)];
    double* D = S + (size_t)k * NB + (size_t)npad * k * NB;
    const int t = threadIdx.x;
    for (int e = t; e < NB * NB; e += 256) { const int i = e % NB, j = e / NB; M[i + (NB + 1) * j] = D[(size_t)i + (size_t)npad * j]; }
    __syncthreads();
    for (int j = 0; j < NB; ++j) {
        double d = M[j + (NB + 1) * j];
        if (d == 0.0 || d != d) { if (t == 0) atomicCAS(status, 0, 1 + k * NB + j); d = 1.0; }
        const double id = 1.0 / d;
        // trailing update with the un-scaled column u: M(i,c) -= u_i * u_c / d, j < c <= i
        const int m = NB - 1 - j;
        for (int e = t; e < m * m; e += 256) { const int i = j + 1 + e % m, c2 = j + 1 + e / m; if (i >= c2) M[i + (NB + 1) * c2] -= M[i + (NB + 1) * j] * M[c2 + (NB + 1) * j] * id; }
        __syncthreads();
        if (t > j && t < NB) M[t + (NB + 1) * j] *= id;
        if (t == 0) M[j + (NB + 1) * j] = d;
        __syncthreads();
    }
    for (int e = t; e < NB * NB; e += 256) { const int i = e % NB, j = e / NB; if (i >= j) D[(size_t)i + (size_t)npad * j] = M[i + (NB + 1) * j]; }
}
// panel: W = A_ik * L_kk^-T (unit diagonal) and L_ik = W * D_k^-1 for each 64-row block i > k; one workgroup
// (64 threads, thread = row) per block.  L goes back into S, W (= L*D) into the panel workspace for the update.
__global__ __launch_bounds__(64) void trsm_panel_kernel(double* __restrict__ S, double* __restrict__ W, int npad, int k) {
    __shared__ double L[NB * (NB + 1)];
    const double* D = S + (size_t)k * NB + (size_t)npad * k * NB;
    const int t = threadIdx.x; const int ib = k + 1 + blockIdx.x;
    for (int e = t; e < NB * NB; e += 64) { const int i = e % NB, j = e / NB; L[i + (NB + 1) * j] = D[(size_t)i + (size_t)npad * j]; }
    __syncthreads();
    double* P = S + (size_t)ib * NB + t + (size_t)npad * k * NB;   // row t of the block, stride npad between columns
    double* Wr = W + (size_t)ib * NB + t;
    double xr[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) xr[j] = P[(size_t)npad * j];
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        double v = xr[j];
#pragma unroll
        for (int l = 0; l < j; ++l) v -= xr[l] * L[j + (NB + 1) * l];
        xr[j] = v;
    }
#pragma unroll
    for (int j = 0; j < NB; ++j) { Wr[(size_t)npad * j] = xr[j]; P[(size_t)npad * j] = xr[j] / L[j + (NB + 1) * j]; }
}
// trailing update on the matrix cores: C_ij -= W_i * L_j' (W = L*D) for all k < j <= i; one 64x64 tile per workgroup,
// 4 waves x (2x2) v_mfma_f64_16x16x4_f64 accumulators.
__global__ __launch_bounds__(256) void syrk_update_kernel(double* __restrict__ S, const double* __restrict__ W, int npad, int k, int nblk) {
    __shared__ double Pi[NB * LDT];   // Pi[r + LDT*kk]
    __shared__ double Pj[NB * LDT];
    const int T = nblk - k - 1;
    // linear tile index -> (ti >= tj)
    int tix = blockIdx.x; int ti = (int)((sqrt(8.0 * tix + 1.0) - 1.0) * 0.5);
    while (ti * (ti + 1) / 2 > tix) --ti;
    while ((ti + 1) * (ti + 2) / 2 <= tix) ++ti;
    const int tj = tix - ti * (ti + 1) / 2;
    (void)T;
    const int ib = k + 1 + ti, jb = k + 1 + tj;
    const double* Gi = W + (size_t)ib * NB;                       // W = L*D rows of block i
    const double* Gj = S + (size_t)jb * NB + (size_t)npad * k * NB;
    const int t = threadIdx.x;
    for (int e = t; e < NB * NB; e += 256) { const int r = e % NB, c2 = e / NB; Pi[r + LDT * c2] = Gi[(size_t)r + (size_t)npad * c2]; Pj[r + LDT * c2] = Gj[(size_t)r + (size_t)npad * c2]; }
    __syncthreads();
    const int w = t >> 6, lane = t & 63;
    const int r0 = (w & 1) * 32, c0 = (w >> 1) * 32;
    double4_t acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b2 = 0; b2 < 2; ++b2) acc[a][b2] = double4_t{0, 0, 0, 0};
    const int li = lane & 15, lk = lane >> 4;
#pragma unroll 4
    for (int kk = 0; kk < NB; kk += 4) {
        double av[2], bv[2];
#pragma unroll
        for (int a = 0; a < 2; ++a) av[a] = Pi[r0 + 16 * a + li + LDT * (kk + lk)];
#pragma unroll
        for (int b2 = 0; b2 < 2; ++b2) bv[b2] = Pj[c0 + 16 * b2 + li + LDT * (kk + lk)];
        // the product is formed TRANSPOSED (operands swapped): the accumulator then has the ROW of C on the lane index, and a store
        // instruction covers 16 consecutive rows of one column of the column-major S (128 contiguous bytes) instead of 16 columns
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b2 = 0; b2 < 2; ++b2) acc[a][b2] = __builtin_amdgcn_mfma_f64_16x16x4f64(bv[b2], av[a], acc[a][b2], 0, 0, 0);
    }
    // C/D layout of the f64 MFMA: col = lane & 15, row = (lane >> 4) + 4*reg -- of the TRANSPOSED tile: C row = lane & 15
    double* Cg = S + (size_t)ib * NB + (size_t)npad * jb * NB;
    double cold[2][2][4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b2 = 0; b2 < 2; ++b2)
#pragma unroll
            for (int r = 0; r < 4; ++r) cold[a][b2][r] = Cg[(size_t)(r0 + 16 * a + li) + (size_t)npad * (c0 + 16 * b2 + lk + 4 * r)];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b2 = 0; b2 < 2; ++b2)
#pragma unroll
            for (int r = 0; r < 4; ++r) Cg[(size_t)(r0 + 16 * a + li) + (size_t)npad * (c0 + 16 * b2 + lk + 4 * r)] = cold[a][b2][r] - acc[a][b2][r];
}
// The trailing update for ONE or TWO panels at a time (NK = 1, 2): C_ij -= sum_q W_{k0+q},i * L_{k0+q},j'.  With two panels per pass every tile of
// the trailing matrix is read and written half as often -- that read-modify-write of S is what bounds the update (its operands, 3 MB per
// panel, stay in L2).  narrow != 0: only the block column jb0 (the next panel: all that its factorisation waits for), one workgroup per
// row block; else the triangle of blocks >= jb0.
template <int NK>
__global__ __launch_bounds__(256) void syrk_update2_kernel(double* __restrict__ S, const double* __restrict__ W0, const double* __restrict__ W1, int npad, int k0, int jb0, int narrow, int wq = -1, int wstrip = 0) {
    __shared__ double Pi[NB * LDT];   // Pi[r + LDT*kk]
    __shared__ double Pj[NB * LDT];
    int ti, tj;
    if (narrow) { ti = blockIdx.x; tj = 0; }
    else { const int tix = blockIdx.x; ti = (int)((sqrt(8.0 * tix + 1.0) - 1.0) * 0.5); while (ti * (ti + 1) / 2 > tix) --ti; while ((ti + 1) * (ti + 2) / 2 <= tix) ++ti; tj = tix - ti * (ti + 1) / 2; }
    // (windowed factorisation, wq >= 0: logical 64-row block t of the step is block jb0 + t inside the band window and wstrip + (t - wq) in the bottom strip)
    const int ib = (wq < 0 || ti < wq) ? jb0 + ti : wstrip + (ti - wq), jb = (wq < 0 || tj < wq) ? jb0 + tj : wstrip + (tj - wq);
    const int t = threadIdx.x, w = t >> 6, lane = t & 63, li = lane & 15, lk = lane >> 4;
    const int r0 = (w & 1) * 32, c0 = (w >> 1) * 32;
    double4_t acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b2 = 0; b2 < 2; ++b2) acc[a][b2] = double4_t{0, 0, 0, 0};
#pragma unroll
    for (int q = 0; q < NK; ++q) {
        const double* Gi = (q == 0 ? W0 : W1) + (size_t)ib * NB;                         // W = L*D rows of block i, panel k0 + q
        const double* Gj = S + (size_t)jb * NB + (size_t)npad * (k0 + q) * NB;           // L rows of block j, panel k0 + q
        if (q > 0) __syncthreads();
        {   // every load first, then the LDS stores: a copy loop global -> LDS waits for each load before it issues the next (16 dependent round trips
            // per operand were most of this kernel's 18.7 us)
            constexpr int NQ = NB * NB / 256; double vi[NQ], vj[NQ];
#pragma unroll
            for (int u = 0; u < NQ; ++u) { const int e = t + 256 * u, r = e % NB, c2 = e / NB; vi[u] = Gi[(size_t)r + (size_t)npad * c2]; vj[u] = Gj[(size_t)r + (size_t)npad * c2]; }
#pragma unroll
            for (int u = 0; u < NQ; ++u) { const int e = t + 256 * u, r = e % NB, c2 = e / NB; Pi[r + LDT * c2] = vi[u]; Pj[r + LDT * c2] = vj[u]; }
        }
        __syncthreads();
#pragma unroll 4
        for (int kk = 0; kk < NB; kk += 4) {
            double av[2], bv[2];
#pragma unroll
            for (int a = 0; a < 2; ++a) av[a] = Pi[r0 + 16 * a + li + LDT * (kk + lk)];
#pragma unroll
            for (int b2 = 0; b2 < 2; ++b2) bv[b2] = Pj[c0 + 16 * b2 + li + LDT * (kk + lk)];
            // (formed TRANSPOSED, operands swapped: the accumulator then has the ROW of C on the lane index -- see syrk_update_kernel)
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b2 = 0; b2 < 2; ++b2) acc[a][b2] = __builtin_amdgcn_mfma_f64_16x16x4f64(bv[b2], av[a], acc[a][b2], 0, 0, 0);
        }
    }
    double* Cg = S + (size_t)ib * NB + (size_t)npad * jb * NB;
    double cold[2][2][4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b2 = 0; b2 < 2; ++b2)
#pragma unroll
            for (int r = 0; r < 4; ++r) cold[a][b2][r] = Cg[(size_t)(r0 + 16 * a + li) + (size_t)npad * (c0 + 16 * b2 + lk + 4 * r)];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b2 = 0; b2 < 2; ++b2)
#pragma unroll
            for (int r = 0; r < 4; ++r) Cg[(size_t)(r0 + 16 * a + li) + (size_t)npad * (c0 + 16 * b2 + lk + 4 * r)] = cold[a][b2][r] - acc[a][b2][r];
}
// The two-panel trailing update with 128 x 128 output tiles: C -= W_k0 L_k0' + W_{k0+1} L_{k0+1}' (K = 128).  EIGHT wavefronts, each a 64 x 32
// block = 4 x 2 accumulator tiles of v_mfma_f64_16x16x4_f64 (6 LDS operand reads per 8 MFMAs; the 64 x 64 kernel above needs 4 per 4 and four
// times the operand traffic per flop).  114 registers: two workgroups per CU are four waves per SIMD -- with four waves per tile (64 x 64 each,
// 213 registers, two waves per SIMD) the same tile took 8 % longer: every workgroup alternates between its MFMA loop and phases in which it
// only waits (first operands, the read-modify-write of C at the end), and two waves per SIMD leave the matrix pipe idle whenever both wait.
// The operands go through LDS in chunks of 16 columns, double buffered: the global loads of chunk c + 1 are in flight while chunk c is
// multiplied.  Products are formed transposed (operands swapped) so that the read-modify-write of S is coalesced.  Tiles start at 64-block
// jb0; when the number of 64-blocks behind it is odd the last tile row / column is half empty: those wavefronts idle.  The blocks above the
// diagonal of a diagonal tile are not needed either.
// What bounds a pass (tools/dense/syrk_test.hip, 1081 tiles = the first pass of a 6000-dof system, 130 us): two workgroups share a CU at 70 k
// cycles per tile each (the MFMAs of a tile are 32.8 k cycles per SIMD: the steady state is matrix-bound), so a CU works through its 4.2 tiles
// in three rounds; the read-modify-write of C adds 19 us, the operand loads 18 us (each alone; lone workgroup: loop 34.5 k cycles without,
// 46.4 k with the operand loads).  Persistent workgroups with dynamic tile fetch and a half-tile stagger between the two workgroups of a CU
// were measured there too: no gain (129 us).
constexpr int S128_KC = 16, S128_LD = 144;
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) void syrk_update128_kernel(double* __restrict__ S, const double* __restrict__ W0, const double* __restrict__ W1, int npad, int k0, int jb0, int firstcol, int wq = -1, int wstrip = 0) {
    __shared__ double As[2][S128_KC * S128_LD], Bs[2][S128_KC * S128_LD];
    int ti, tj;                                                // firstcol: only the first tile column (what the next two panels wait for)
    if (firstcol == 1) { ti = blockIdx.x; tj = 0; }
    else { const int tix = blockIdx.x; ti = (int)((sqrt(8.0 * tix + 1.0) - 1.0) * 0.5); while (ti * (ti + 1) / 2 > tix) --ti; while ((ti + 1) * (ti + 2) / 2 <= tix) ++ti; tj = tix - ti * (ti + 1) / 2;
           if (firstcol == 2) { ++ti; ++tj; } }                  // firstcol == 2: everything BUT the first tile column (the look-ahead factorisation: that column went ahead)
    // (windowed, wq >= 0: logical 128-row block t is row jb0 NB + 128 t inside the band window, 128 (wstrip + t - wq) in the bottom strip)
    const int I0 = (wq < 0 || ti < wq) ? jb0 * NB + 128 * ti : 128 * (wstrip + ti - wq), J0 = (wq < 0 || tj < wq) ? jb0 * NB + 128 * tj : 128 * (wstrip + tj - wq);
    const int t = threadIdx.x, w = __builtin_amdgcn_readfirstlane(t >> 6), lane = t & 63, li = lane & 15, lk = lane >> 4;
    const int r0w = (w & 1) * 64, c0w = (w >> 1) * 32;
    const bool active = I0 + r0w < npad && J0 + c0w < npad && !(ti == tj && c0w >= r0w + 64);
    double4_t acc[4][2];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b2 = 0; b2 < 2; ++b2) acc[a][b2] = double4_t{0, 0, 0, 0};
    // copy roles: thread t moves row (t & 127) of both operands, every fourth column of a chunk (kq, kq + 4, ..)
    const int cr = t & 127, kq = t >> 7;
    const int arow = I0 + cr < npad ? I0 + cr : npad - 1, brow = J0 + cr < npad ? J0 + cr : npad - 1;
    constexpr int NCP = S128_KC / 4, CPP = NB / S128_KC;       // columns per thread and chunk; chunks per panel
    double ra[NCP], rb[NCP];
    auto gload = [&](int chunk) {                              // chunk: panel (chunk / CPP), columns S128_KC * (chunk % CPP) ..
        const int q = chunk / CPP, col0 = (chunk % CPP) * S128_KC;
        const double* Wq = q == 0 ? W0 : W1;
        const double* Ga = Wq + (size_t)arow + (size_t)npad * col0;
        const double* Gb = S + (size_t)brow + (size_t)npad * ((size_t)(k0 + q) * NB + col0);
#pragma unroll
        for (int i = 0; i < NCP; ++i) { ra[i] = Ga[(size_t)npad * (kq + 4 * i)]; rb[i] = Gb[(size_t)npad * (kq + 4 * i)]; }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NCP; ++i) { As[buf][(kq + 4 * i) * S128_LD + cr] = ra[i]; Bs[buf][(kq + 4 * i) * S128_LD + cr] = rb[i]; }
    };
    constexpr int NCH = 2 * NB / S128_KC;
    gload(0); lstore(0);
    __syncthreads();
#pragma unroll 1
    for (int ch = 0; ch < NCH; ++ch) {
        const int buf = ch & 1;
        if (ch + 1 < NCH) gload(ch + 1);
        if (active) {
#pragma unroll
            for (int kk = 0; kk < S128_KC; kk += 4) {
                double av[4], bv[2];
#pragma unroll
                for (int a = 0; a < 4; ++a) av[a] = As[buf][(kk + lk) * S128_LD + r0w + 16 * a + li];
#pragma unroll
                for (int b2 = 0; b2 < 2; ++b2) bv[b2] = Bs[buf][(kk + lk) * S128_LD + c0w + 16 * b2 + li];
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int b2 = 0; b2 < 2; ++b2) acc[a][b2] = __builtin_amdgcn_mfma_f64_16x16x4f64(bv[b2], av[a], acc[a][b2], 0, 0, 0);
            }
        }
        if (ch + 1 < NCH) lstore(buf ^ 1);
        __syncthreads();
    }
    if (!active) return;
    // C/D layout of the f64 MFMA of the TRANSPOSED tile: C row = lane & 15 (+ 16 a), column = (lane >> 4) + 4 r (+ 16 b).
    // (A no-return atomic add per entry -- every entry has one writer per pass -- measured slower than this read-modify-write.)
    double* Cg = S + (size_t)(I0 + r0w) + (size_t)npad * (J0 + c0w);
#pragma unroll
    for (int b2 = 0; b2 < 2; ++b2) {
        double cold[4][4];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int r = 0; r < 4; ++r) cold[a][r] = Cg[(size_t)(16 * a + li) + (size_t)npad * (16 * b2 + lk + 4 * r)];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int r = 0; r < 4; ++r) Cg[(size_t)(16 * a + li) + (size_t)npad * (16 * b2 + lk + 4 * r)] = cold[a][r] - acc[a][b2][r];
    }
}
// backward substitution L' x = z (unit diagonal), block by block from the bottom.  z = D^-1 L^-1 s is row n of the factor.
// step 1 (one workgroup per 64-column block kb, many row blocks): partial[kb][j] = sum_{i > kb block} L[i][kb*64+j] * x[i]
__global__ __launch_bounds__(256) void bwd_gemv_kernel(const double* __restrict__ S, int npad, int kb, int n, const double* __restrict__ x, double* __restrict__ acc) {
    // grid.x = number of row blocks below kb; each adds its 64-vector contribution atomically
    __shared__ double red[4][NB];
    const int ib = kb + 1 + blockIdx.x; const int t = threadIdx.x; const int j = t & 63, q = t >> 6;
    const double* P = S + (size_t)ib * NB + (size_t)npad * ((size_t)kb * NB + j);
    double v = 0;
    for (int i = q * 16; i < q * 16 + 16; ++i) { const int gi = ib * NB + i; if (gi < n) v += P[i] * x[gi]; }
    red[q][j] = v; __syncthreads();
    if (q == 0) atomicAdd(&acc[kb * NB + j], red[0][j] + red[1][j] + red[2][j] + red[3][j]);
}
// step 2: x_k = L_kk^-T (y_k - acc_k), single wave
__global__ __launch_bounds__(64) void bwd_diag_kernel(const double* __restrict__ S, int npad, int kb, int n, const double* __restrict__ acc, double* __restrict__ x) {
    __shared__ double L[NB * (NB + 1)]; __shared__ double r[NB];
    const double* D = S + (size_t)kb * NB + (size_t)npad * kb * NB; const int t = threadIdx.x;
    for (int e = t; e < NB * NB; e += 64) { const int i = e % NB, j = e / NB; L[i + (NB + 1) * j] = D[(size_t)i + (size_t)npad * j]; }
    const int g = kb * NB + t;
    r[t] = (g < n) ? S[(size_t)n + (size_t)npad * g] - acc[g] : 0.0;   // y lives in row n of the factor
    __syncthreads();
    if (t == 0) for (int i = NB - 1; i >= 0; --i) { if (kb * NB + i >= n) { r[i] = 0; continue; } double v = r[i]; for (int l = i + 1; l < NB && kb * NB + l < n; ++l) v -= L[l + (NB + 1) * i] * r[l]; r[i] = v; }
    __syncthreads();
    if (g < n) x[g] = r[t];
}

// ---------------------------------------------------------------------------------------------------
// fast_bAb(H + lambda I, v) and dot(b, v)   src/utils.jl:71-106, src/iterators.jl:52,163
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void quadform_blocks_kernel(const double* __restrict__ A, const SchurCopy* __restrict__ blk, int64_t nblk,
                                                              const double* __restrict__ v, const uint8_t* __restrict__ mask, double* __restrict__ partials) {
    quadform_blocks_body(A, blk, nblk, v, mask, partials, (int)blockIdx.x, (int)gridDim.x);
}
template <int DV>
__global__ __launch_bounds__(256) void quadform_points_kernel(const double* __restrict__ A, const int64_t* __restrict__ ediag, const uint32_t* __restrict__ eboff,
                                                              const uint32_t* __restrict__ members, int64_t nm, const double* __restrict__ tE,
                                                              const double* __restrict__ x, double* __restrict__ partials) {
    quadform_points_body<DV>(A, ediag, eboff, members, nm, tE, x, partials, (int)blockIdx.x, (int)gridDim.x);
}
template <int DV>
__global__ __launch_bounds__(256) void post_solve_kernel(PostSolveArgs a) { post_roles_body<DV>(a, (int)blockIdx.x); }
// out[1] = max|x| (NaN if any entry is), out[2] = x'x, out[4] = x'(H + lambda I)x, out[5] = g'x, out[8] = x'Hx, out[9] = (masked) x'x,
// out[10] = factorisation status
NLLS_DEV void post_solve_finish_body(const double* __restrict__ partials, int np, const double* __restrict__ part2, int np2,
                                     double lambda, double* __restrict__ out, const int* __restrict__ status, double (*red)[4]) {
    double a = 0, m = 0, nan = 0, ss = 0, vv = 0, bv = 0;
    for (int i = threadIdx.x; i < np; i += 256) a += partials[i];
    for (int i = threadIdx.x; i < np2; i += 256) { m = fmax(m, part2[5 * i]); nan = fmax(nan, part2[5 * i + 1]); ss += part2[5 * i + 2]; vv += part2[5 * i + 3]; bv += part2[5 * i + 4]; }
    a = wsum(a); ss = wsum(ss); vv = wsum(vv); bv = wsum(bv);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { m = fmax(m, __shfl_xor(m, o)); nan = fmax(nan, __shfl_xor(nan, o)); }
    if ((threadIdx.x & 63) == 0) { const int w = threadIdx.x >> 6; red[0][w] = a; red[1][w] = m; red[2][w] = nan; red[3][w] = ss; red[4][w] = vv; red[5][w] = bv; }
    __syncthreads();
    if (threadIdx.x == 0) {
        a = red[0][0] + red[0][1] + red[0][2] + red[0][3]; ss = red[3][0] + red[3][1] + red[3][2] + red[3][3];
        vv = red[4][0] + red[4][1] + red[4][2] + red[4][3]; bv = red[5][0] + red[5][1] + red[5][2] + red[5][3];
        m = fmax(fmax(red[1][0], red[1][1]), fmax(red[1][2], red[1][3])); nan = fmax(fmax(red[2][0], red[2][1]), fmax(red[2][2], red[2][3]));
        out[1] = nan > 0 ? __longlong_as_double(0x7ff8000000000000LL) : m; out[2] = ss;
        out[4] = a + lambda * vv; out[5] = bv; out[8] = a; out[9] = vv;
        out[10] = (double)status[0];                             // the factorisation status rides home with the scalars (one copy)
    }
}
__global__ __launch_bounds__(256) void post_solve_finish_kernel(const double* __restrict__ partials, int np, const double* __restrict__ part2, int np2,
                                                                double lambda, double* __restrict__ out, const int* __restrict__ status) {
    __shared__ double red[6][4];
    post_solve_finish_body(partials, np, part2, np2, lambda, out, status, red);
}
// the two one-workgroup reductions that end an LM trial in ONE launch: workgroup 0 sums the cost partials (the same order as
// reduce_partials_kernel: the totals are bit-identical), workgroup 1 finishes the step statistics
// Workgroups 2 .. (zr.n + 1), when the look-ahead sweep follows (nlls_lm_trial): the zero fill of the rows that sweep accumulates into with atomics (zero_ranges_kernel's
// work -- a launch of its own in front of every other sweep).  Nothing reads A or b between this launch and that sweep.
struct ZeroRanges { double* A; const int64_t* off; const uint32_t* len; double* b; const uint32_t* boff; const uint32_t* blen; int n; };
__global__ __launch_bounds__(256) void trial_finish_kernel(const double* __restrict__ cpart, int64_t ncp, const double* __restrict__ partials, int np,
                                                           const double* __restrict__ part2, int np2, double lambda, double* __restrict__ out, const int* __restrict__ status,
                                                           double* __restrict__ host_out, double seq, ZeroRanges zr, double* __restrict__ stamps) {
    __shared__ double red[6][4];
    if (blockIdx.x >= 2) {
        const int r = (int)blockIdx.x - 2;
        const int64_t o = zr.off[r]; const uint32_t l = zr.len[r];
        for (uint32_t i = threadIdx.x; i < l; i += 256) zr.A[o + i] = 0.0;
        const uint32_t bo = zr.boff[r], bl = zr.blen[r];
        for (uint32_t i = threadIdx.x; i < bl; i += 256) zr.b[bo + i] = 0.0;
        return;
    }
    if (blockIdx.x == 0) reduce_partials_body(cpart, ncp, out, &red[0][0]);
    else post_solve_finish_body(partials, np, part2, np2, lambda, out, status, red);
    // the scalars also go straight to the pinned host mirror (device-visible, coherent host memory): no copy command behind this launch.
    // Each workgroup then publishes the trial's sequence number: the host spins on the two numbers instead of sleeping in a stream
    // synchronisation (its wake-up costs more than the kernels of this size it waits for).
    if (host_out && threadIdx.x == 0) {
        if (blockIdx.x == 0) host_out[0] = out[0];
        else { host_out[1] = out[1]; host_out[2] = out[2]; host_out[4] = out[4]; host_out[5] = out[5]; host_out[8] = out[8]; host_out[9] = out[9]; host_out[10] = out[10]; }
        if (blockIdx.x == 1) time_stamp(stamps, 3);
        __threadfence_system();
        reinterpret_cast<volatile double*>(host_out)[32 + blockIdx.x] = seq;
    }
}
__global__ __launch_bounds__(256) void quadform_dense_kernel(const double* __restrict__ A, int n, const double* __restrict__ v, double* __restrict__ partials) {
    __shared__ double red[4];
    double acc = 0;
    for (int j = blockIdx.x * 256 + threadIdx.x; j < n; j += gridDim.x * 256) { double c2 = 0; for (int i = 0; i < n; ++i) c2 += A[(size_t)i + (size_t)n * j] * v[i]; acc += c2 * v[j]; }
    acc = wsum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partials[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
// partial (v'v, b'v) per workgroup
__global__ __launch_bounds__(256) void dot2_partial_kernel(const double* __restrict__ b, const double* __restrict__ v, const double* __restrict__ mask, int64_t n, double* __restrict__ part) {
    __shared__ double red[2][4];
    double vv = 0, bv = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) { const double m = mask ? mask[i] : 1.0; const double x = v[i]; vv += m * x * x; bv += m * b[i] * x; }
    vv = wsum(vv); bv = wsum(bv);
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = vv; red[1][threadIdx.x >> 6] = bv; }
    __syncthreads();
    if (threadIdx.x == 0) { part[2 * blockIdx.x] = red[0][0] + red[0][1] + red[0][2] + red[0][3]; part[2 * blockIdx.x + 1] = red[1][0] + red[1][1] + red[1][2] + red[1][3]; }
}
// out[slot] = sum(partials) + lambda * v'v ; out[slot+1] = b'v
__global__ __launch_bounds__(256) void quadform_finish_kernel(const double* __restrict__ partials, int np, const double* __restrict__ part2, int np2,
                                                              double lambda, double* __restrict__ out, int slot) {
    __shared__ double red[3][4];
    double a = 0, vv = 0, bv = 0;
    for (int i = threadIdx.x; i < np; i += 256) a += partials[i];
    for (int i = threadIdx.x; i < np2; i += 256) { vv += part2[2 * i]; bv += part2[2 * i + 1]; }
    a = wsum(a); vv = wsum(vv); bv = wsum(bv);
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = a; red[1][threadIdx.x >> 6] = vv; red[2][threadIdx.x >> 6] = bv; }
    __syncthreads();
    if (threadIdx.x == 0) {
        a = red[0][0] + red[0][1] + red[0][2] + red[0][3]; vv = red[1][0] + red[1][1] + red[1][2] + red[1][3]; bv = red[2][0] + red[2][1] + red[2][2] + red[2][3];
        out[slot] = a + lambda * vv; out[slot + 1] = bv;
        if (slot == 4) { out[8] = a; out[9] = vv; }             // the step's raw parts: x'Ax and x'x (nlls_solve caches them)
    }
}

// ---------------------------------------------------------------------------------------------------
// assembly of the reduced system from the supernodes' slabs, straight into the block cyclic reduction's tiles:
// one wavefront per block pair of S (per rhs segment), shares summed in the order the structure lists them.
// The tiles have been zero-filled by the previous solve's back-substitution launch (or the allocation).
// ---------------------------------------------------------------------------------------------------
struct GatherArgs { const GatherJob* jobs; const GatherCon* cons; int64_t njobs; const double* slab; const double* A; const double* b; const double* Cinv; int dv; double lambda; BcrGeom g; int* status; };
__device__ __forceinline__ void gather_store(const BcrGeom& g, int r, int c, double v) {      // S(r, c), r >= c
    const int NT = g.NT, bsz = 16 * NT;
    if (r >= g.n_band) {
        const int br = r - g.n_band;
        if (c >= g.n_band) { const int bc = c - g.n_band; g.ws[g.ocp + br * 16 + bc] = v; g.ws[g.ocp + bc * 16 + br] = v; }
        else { const int k = c / bsz, K = (c - k * bsz) >> 4; g.ws[g.oBR + ((size_t)k * NT + K) * 256 + br * 16 + (c & 15)] = v; }
        return;
    }
    const int kr = r / bsz, kc = c / bsz, I = (r - kr * bsz) >> 4, K = (c - kc * bsz) >> 4;
    if (kr == kc) {
        double* t = g.ws + g.oD + ((size_t)kr * (NT * (NT + 1) / 2) + I * (I + 1) / 2 + K) * 256;
        t[(r & 15) * 16 + (c & 15)] = v; if (I == K) t[(c & 15) * 16 + (r & 15)] = v;
        if (r == c) g.ws[g.odg + r] = fabs(v);                // the original diagonal (pivot floor of undamped solves), as bcr_convert_kernel records it
    } else g.ws[g.oA + ((size_t)kr * NT * NT + I * NT + K) * 256 + (r & 15) * 16 + (c & 15)] = v;     // kr == kc + 1: the block size covers the bandwidth
}
__global__ __launch_bounds__(256) void schur_gather_kernel(GatherArgs a) {
    const int64_t j = (int64_t)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); if (j >= a.njobs) return;
    const GatherJob J = a.jobs[j]; const int lane = threadIdx.x & 63;
    const BcrGeom& g = a.g;
    if (J.kind == 2) {                                         // identity on the padding behind the band (the last block's unused columns)
        for (int r = g.n_band + lane; r < g.N * 16 * g.NT; r += 64) { const int bsz = 16 * g.NT, k = r / bsz, I = (r - k * bsz) >> 4;
            g.ws[g.oD + ((size_t)k * (g.NT * (g.NT + 1) / 2) + I * (I + 1) / 2 + I) * 256 + (r & 15) * 17] = 1.0; }
        return;
    }
    const int rows = J.rows, ne = rows * J.cols;
    // (every lane walks the same loops -- the shares' descriptors travel by read-lane from the lane that loaded them --; a lane without an element of its own reads element 0 and stores nothing)
    for (int e0 = 0; e0 < ne; e0 += 64) {
        const int e = e0 + lane; const bool inr = e < ne; const int ee = inr ? e : 0;
        const int b2 = ee / rows, a2 = ee - b2 * rows;
        const bool diag = J.kind == 0 && J.r0 == J.c0;
        const bool live = inr && !(diag && a2 < b2);             // diagonal block: the lower triangle (mirrored by the store)
        double v = 0.0;
        if (J.kind == 0) { if (J.copy_off >= 0) v = J.copy_trans ? a.A[J.copy_off + b2 + (int)J.cols * a2] : a.A[J.copy_off + a2 + rows * b2]; if (diag && a2 == b2) v += a.lambda; }
        else v = a.b[J.boff + a2];
        // shares: the descriptors of up to 64 of them come with ONE coalesced load (lane u takes share u) and reach every lane through read-lanes; the values are then requested
        // eight at a time, all in flight together (round 6: four at a time behind uniform descriptor loads was a chain of dependent round trips, 16.7 us at BASELINE config 4)
        for (uint32_t c0 = J.cbeg; c0 < J.cend; c0 += 64) {
            const uint32_t nc = min(64u, J.cend - c0);
            const GatherCon mine = a.cons[c0 + ((uint32_t)lane < nc ? (uint32_t)lane : 0u)];
            for (uint32_t u0 = 0; u0 < nc; u0 += 8) {
                double t[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int src = (int)min(u0 + (uint32_t)u, nc - 1);
                    GatherCon k; k.off = (uint32_t)__builtin_amdgcn_readlane((int)mine.off, src); k.ld = (uint32_t)__builtin_amdgcn_readlane((int)mine.ld, src);
                    k.aux = (uint32_t)__builtin_amdgcn_readlane((int)mine.aux, src); k.cinv = 0;
                    t[u] = k.aux ? a.slab[k.off + b2 + k.ld * a2] : a.slab[k.off + a2 + k.ld * b2];   // aux: the share lies above the diagonal of S in reduced order -- its transpose is wanted
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) if (u0 + (uint32_t)u < nc) v -= t[u];
            }
        }
        if (!live) continue;
        if (J.kind == 0) gather_store(g, (int)J.r0 + a2, (int)J.c0 + b2, v);
        else {                                                 // the rhs row: row nbd of the border / rhs tiles
            const int r = (int)J.r0 + a2, NT = g.NT, bsz = 16 * NT;
            if (r >= g.n_band) g.ws[g.ocp + g.nbd * 16 + (r - g.n_band)] = v;
            else { const int k = r / bsz, K = (r - k * bsz) >> 4; g.ws[g.oBR + ((size_t)k * NT + K) * 256 + g.nbd * 16 + (r & 15)] = v; }
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------
static int herr(nlls_ctx* c, hipError_t e, const char* what) { c->err = std::string(what) + ": " + hipGetErrorString(e); return NLLS_ERR_HIP; }
#define HIPCHK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return herr(c, e_, #expr); } while (0)

int enqueue_quadform(nlls_ctx* c, const double* d_vec, int out_slot) {
    int np = 0;
    if (c->info.is_sparse) {
        // the step of the last solve: the rows of fast-path members come from E_v s, which the back-substitution kept
        const bool reuse = d_vec == c->x.p && c->tE_valid && c->n_fast_members > 0;
        np = (int)std::min<int64_t>((c->nblk * QF_COLS + 255) / 256, 768); if (np < 1) np = 1;
        hipLaunchKernelGGL(quadform_blocks_kernel, dim3(np), dim3(256), 0, c->stream, c->A.p, c->d_blk.p, c->nblk, d_vec,
                           reuse ? c->d_blk_slowmask.p : (c->nranks > 1 ? c->d_blk_mask.p : (const uint8_t*)nullptr), c->partials.p);
        if (reuse) {
            const int np3 = (int)std::max<int64_t>(1, std::min<int64_t>((c->n_fast_members + 255) / 256, 256));
#define LAUNCH_QP(DV) hipLaunchKernelGGL((quadform_points_kernel<DV>), dim3(np3), dim3(256), 0, c->stream, c->A.p, c->d_elim_diag.p, c->d_elim_boff.p, c->d_fast_members.p, \
                c->n_fast_members, c->tE.p, d_vec, c->partials.p + np)
            if (c->fast_dv == 3) LAUNCH_QP(3); else if (c->fast_dv == 2) LAUNCH_QP(2); else LAUNCH_QP(1);
#undef LAUNCH_QP
            np += np3;
        }
    } else {
        np = (int)std::min<int64_t>((c->info.ndof + 255) / 256, 1024); if (np < 1) np = 1;
        hipLaunchKernelGGL(quadform_dense_kernel, dim3(np), dim3(256), 0, c->stream, c->A.p, (int)c->info.ndof, d_vec, c->partials.p);
    }
    const int np2 = (int)std::max<int64_t>(1, std::min<int64_t>((c->info.ndof + 255) / 256, 256));
    double* part2 = c->partials.p + 1024;
    hipLaunchKernelGGL(dot2_partial_kernel, dim3(np2), dim3(256), 0, c->stream, c->b.p, d_vec, c->nranks > 1 ? c->d_dof_mask.p : (const double*)nullptr, c->info.ndof, part2);
    hipLaunchKernelGGL(quadform_finish_kernel, dim3(1), dim3(256), 0, c->stream, c->partials.p, np, part2, np2, c->lambda, c->scalars.p, out_slot);
    HIPCHK(hipGetLastError());
    return NLLS_OK;
}

// the arguments of the step-statistics roles (nlls_post.hpp) for the step of the last solve; retract_to >= 0: with the retraction role
PostSolveArgs post_solve_args(nlls_ctx* c, int retract_to, int retract_from) {
    const bool reuse = c->tE_valid && c->n_fast_members > 0;
    PostSolveArgs a{};
    const bool lazy = c->nranks > 1 && !c->reduced_summed;      // the reduced rows of A.data and b hold this rank's share only: they count on every rank
    a.A = c->A.p; a.blk = reuse ? (lazy ? c->d_blk_slow_lazy.p : c->d_blk_slow.p) : c->d_blk.p; a.nblk = reuse ? (lazy ? c->nblk_slow_lazy : c->nblk_slow) : c->nblk;
    a.blkmask = reuse ? (const uint8_t*)nullptr : (c->nranks > 1 ? (lazy ? c->d_blk_mask_lazy.p : c->d_blk_mask.p) : (const uint8_t*)nullptr);
    a.ediag = c->d_elim_diag.p; a.eboff = c->d_elim_boff.p; a.members = c->n_fast_members == (int64_t)c->d_elim_diag.n ? (const uint32_t*)nullptr : c->d_fast_members.p; a.nm = c->n_fast_members; a.tE = c->tE.p;
    a.x = c->x.p; a.b = c->b.p; a.dofmask = c->nranks > 1 ? c->d_dof_mask.p : (const double*)nullptr; a.dofmask_b = lazy ? c->d_dof_mask_lazy.p : (const double*)nullptr; a.ndof = c->info.ndof;
    a.np = (int)std::max<int64_t>(1, std::min<int64_t>((a.nblk * QF_COLS + 255) / 256, 768));
    a.np3 = reuse ? (int)std::max<int64_t>(1, std::min<int64_t>((c->n_fast_members + 255) / 256, 256)) : 0;
    a.np2 = (int)std::max<int64_t>(1, std::min<int64_t>((c->info.ndof + 255) / 256, 512));      // (5 partials each, behind the quadratic form's at 1024: ends at 3584 < TRIAL_COST_POFS)
    a.partials = c->partials.p; a.part2 = c->partials.p + 1024;
    a.stamps = c->stamp_ptr();
    a.nretract = 0;
    if (retract_to >= 0 && c->info.nvar > 0) {
        a.nretract = (int)((c->info.nvar + 255) / 256); a.vkind = c->d_var_kind.p; a.vdim = c->d_var_dim.p; a.voff = c->d_var_off.p; a.vboff = c->d_var_boff.p;
        a.nvar = c->info.nvar; a.vfrom = vars_ptr(c, retract_from); a.vto = vars_ptr(c, retract_to);
    }
    c->ps_np = a.np + a.np3; c->ps_np2 = a.np2;
    return a;
}
__global__ void status_to_scalar_kernel(const int* __restrict__ status, double* __restrict__ out) { *out = (double)status[0]; }
// step statistics + quadratic form of the step of the last solve (what nlls_solve / nlls_lm_trial / nlls_trial_local
// precompute): one launch + one finishing workgroup on sparse systems, the separate kernels otherwise
int enqueue_post_solve(nlls_ctx* c, int retract_to, int retract_from, bool finish) {
    if (!c->info.is_sparse) {
        if (retract_to >= 0) { int rc0 = enqueue_retract(c, retract_to, retract_from); if (rc0 != NLLS_OK) return rc0; }
        int rc = enqueue_step_stats(c); if (rc != NLLS_OK) return rc;
        rc = enqueue_quadform(c, c->x.p, 4); if (rc != NLLS_OK) return rc;
        hipLaunchKernelGGL(status_to_scalar_kernel, dim3(1), dim3(1), 0, c->stream, c->d_status.p, c->scalars.p + 10);
        return NLLS_OK;
    }
    PostSolveArgs a = post_solve_args(c, retract_to, retract_from);
    const dim3 grid((unsigned)(a.np + a.np3 + a.np2 + a.nretract));
    if (c->fast_dv == 3) hipLaunchKernelGGL((post_solve_kernel<3>), grid, dim3(256), 0, c->stream, a);
    else if (c->fast_dv == 2) hipLaunchKernelGGL((post_solve_kernel<2>), grid, dim3(256), 0, c->stream, a);
    else hipLaunchKernelGGL((post_solve_kernel<1>), grid, dim3(256), 0, c->stream, a);
    if (finish) hipLaunchKernelGGL(post_solve_finish_kernel, dim3(1), dim3(256), 0, c->stream, c->partials.p, a.np + a.np3, a.part2, a.np2, c->lambda, c->scalars.p, c->d_status.p);
    HIPCHK(hipGetLastError());
    return NLLS_OK;
}
// what follows the solve in an LM trial (src/iterators.jl:155-163)
int enqueue_lm_trial_tail(nlls_ctx* c, int to, int from) {
    // (matrix-free trial: the back-substitution launch has retracted, taken the trial point's cost and left the step statistics -- one finishing launch sums its rows)
    if (c->mf_step) { c->retract_done = false; return enqueue_mf_trial_finish(c); }
    if (!c->info.is_sparse) { int rc = enqueue_post_solve(c, to, from); if (rc != NLLS_OK) return rc; return enqueue_sweep_cost(c, to); }
    int rc; int64_t ncp = 0;
    if (c->retract_done) {
        // the retraction went with the back-substitution launch: the statistics roles ride in the cost sweep's (first) launch -- no launch of their own
        c->retract_done = false;
        PostSolveArgs a = post_solve_args(c, -1, -1); a.dv = c->fast_dv; bool taken = false;
        rc = enqueue_sweep_cost(c, to, TRIAL_COST_POFS, &ncp, &a, &taken); if (rc != NLLS_OK) return rc;
        if (!taken) { rc = enqueue_post_solve(c, -1, -1, false); if (rc != NLLS_OK) return rc; }      // (no launch of the cost sweep could carry them)
    } else {
    rc = enqueue_post_solve(c, to, from, false); if (rc != NLLS_OK) return rc;
    rc = enqueue_sweep_cost(c, to, TRIAL_COST_POFS, &ncp); if (rc != NLLS_OK) return rc;
    }
    // (the look-ahead sweep follows: its zero fill rides here -- enqueue_sweep_gradhess skips the launch once)
    ZeroRanges zr{};
    if (c->tail_zero_for_lookahead && c->nzero > 0) { zr = ZeroRanges{c->A.p, c->d_zero_off.p, c->d_zero_len.p, c->b.p, c->d_zero_b_off.p, c->d_zero_b_len.p, (int)c->nzero}; c->heavy_rows_zeroed = true; }
    hipLaunchKernelGGL(trial_finish_kernel, dim3(2 + (unsigned)zr.n), dim3(256), 0, c->stream, c->partials.p + TRIAL_COST_POFS, ncp, c->partials.p, c->ps_np, c->partials.p + 1024, c->ps_np2,
                       c->lambda, c->scalars.p, c->d_status.p, c->h_scalars_dev, (double)(++c->trial_seq), zr, c->stamp_ptr());
    HIPCHK(hipGetLastError());
    return NLLS_OK;
}

// the supernodes' slabs -> the tiles of the block cyclic reduction (the tiles outside the pattern must be zero: the previous solve's back-substitution, or a memset here)
int enqueue_gather(nlls_ctx* c) {
    if (!c->gather_ready || !c->bcr.ready) { c->err = "gather index missing"; return NLLS_ERR_NOT_READY; }
    const BcrGeom& g = c->bcr.geom;
    if (!c->tiles_zeroed) HIPCHK(hipMemsetAsync(g.ws + g.oD, 0, sizeof(double) * (g.oBR + (size_t)g.N * g.NT * 256 - g.oD), c->stream));
    c->tiles_zeroed = false;
    GatherArgs ga{c->d_gjobs.p, c->d_gcons.p, c->n_gjobs, c->slab.p, c->A.p, c->b.p, c->Cinv.p, c->fast_dv, c->lambda, c->bcr.geom, c->d_status.p};
    hipLaunchKernelGGL(schur_gather_kernel, dim3((unsigned)((c->n_gjobs + 3) / 4)), dim3(256), 0, c->stream, ga);
    HIPCHK(hipGetLastError());
    return NLLS_OK;
}

// local phase: assemble this rank's share of [S | s] (rank 0 also contributes the reduced-reduced blocks,
// lambda*I and b_R); under sharding the buffer is then summed over ranks
template <bool TSP>
static int enqueue_solve_local_t(nlls_ctx* c) {
    using LAY = SLayoutT<TSP>;
    const int n = (int)c->nred; if (n == 0) return NLLS_OK;
    if (c->mf_use) return enqueue_mf_solve_local(c);               // the matrix-free trial: the supernodes evaluate their cost blocks themselves (nlls_mf.hip)
    const LAY L = make_layout<TSP>(c); const int npad = TSP ? 0 : L.npad;
    // lazy stage 0 (collective route, reduced rows not summed over ranks): EVERY rank adds its share of the reduced-reduced blocks and of b_R to its
    // share of [S | s] -- the one sum over ranks that follows completes both; the damping is rank 0's
    const bool lazy = c->nranks > 1 && !c->reduced_summed;
    const bool band = c->solve_mode == SOLVE_BAND; const bool lead = c->nranks == 1 || c->rank == 0 || lazy;
    const double lambda_rr = (c->nranks == 1 || c->rank == 0) ? c->lambda : 0.0;
    if (c->elim_slab) {
        // slab + gather assembly (deterministic): (C_v + lambda I)^-1, the supernodes' shares into their slabs, one gather into the tiles
        const int64_t nel = (int64_t)c->d_elim_diag.n;
        const int64_t n60 = c->n_slab60, nnar = c->n_slabnar, nwid = c->n_slabwide;    // (small supernodes are not here: the gather forms their shares itself)
#define LAUNCH_SLAB(DV) do { \
            hipLaunchKernelGGL((schur_cinv_kernel<DV>), dim3((unsigned)((nel + 255) / 256)), dim3(256), 0, c->stream, c->A.p, c->d_elim_diag.p, c->d_elim_dim.p, nel, c->lambda, c->Cinv.p, c->d_status.p); \
            if (n60 > 0) hipLaunchKernelGGL((schur_elim_wave_kernel<DV, 1, 2>), dim3((unsigned)n60), dim3(64 * ELIM_NW), 0, c->stream, c->A.p, c->b.p, \
                c->d_elim_ptr.p, c->d_elim_nbr.p, c->d_elim_diag.p, c->d_elim_boff.p, c->d_elim_group.p, c->d_slab_groups.p, c->Cinv.p, c->slab.p, c->d_slab_off.p); \
            if (nnar > 0) hipLaunchKernelGGL((schur_elim_wave_kernel<DV, 1, 3>), dim3((unsigned)nnar), dim3(64 * ELIM_NW), 0, c->stream, c->A.p, c->b.p, \
                c->d_elim_ptr.p, c->d_elim_nbr.p, c->d_elim_diag.p, c->d_elim_boff.p, c->d_elim_group.p, c->d_slab_groups.p + n60, c->Cinv.p, c->slab.p, c->d_slab_off.p + n60); \
            if (nwid > 0) hipLaunchKernelGGL((schur_elim_wave_kernel<DV, 2, 3>), dim3((unsigned)nwid), dim3(64 * ELIM_NW), 0, c->stream, c->A.p, c->b.p, \
                c->d_elim_ptr.p, c->d_elim_nbr.p, c->d_elim_diag.p, c->d_elim_boff.p, c->d_elim_group.p, c->d_slab_groups.p + n60 + nnar, c->Cinv.p, c->slab.p, c->d_slab_off.p + n60 + nnar); } while (0)
        // (the status reset rides in the gather launch; schur_cinv_kernel may flag a bad pivot before it: reset first)
        HIPCHK(hipMemsetAsync(c->d_status.p, 0, sizeof(int32_t) * 5, c->stream));
        if (c->fast_dv == 3) LAUNCH_SLAB(3); else if (c->fast_dv == 2) LAUNCH_SLAB(2); else LAUNCH_SLAB(1);
#undef LAUNCH_SLAB
        return enqueue_gather(c);
    }
    const bool one_prepare = lead && c->info.is_sparse && c->ncopy > 0;      // status reset, s and the reduced-reduced blocks in one launch
    // ONE launch for the whole assembly (schur_elim_all_kernel): every eliminated block on the fast path with both kinds of supernode present, one rank,
    // [S | s] carrying the right-hand side as a row (band / dense layouts).  NLLS_ELIM_SPLIT=1 keeps the three launches (A/B).
    const int64_t nfast_narrow = c->n_fast_narrow, nfast_wide = c->n_fast_groups - c->n_fast_narrow;
    const bool all_in_one = one_prepare && !c->elim_split && c->nranks == 1 && c->elim_mfma && c->n_slow_groups == 0 && nfast_narrow > 0 && nfast_wide > 0 &&
                            c->solve_mode != SOLVE_SMALL && c->fast_dv >= 1 && c->fast_dv <= 3 && (int64_t)c->d_elim_diag.n == c->n_fast_members;
    if (all_in_one) {
        if (!c->status_known_zero) HIPCHK(hipMemsetAsync(c->d_status.p, 0, sizeof(int32_t) * 5, c->stream));
        c->status_known_zero = false;
        if (!c->S_zeroed) HIPCHK(hipMemsetAsync(c->S.p, 0, sizeof(double) * (c->s_elems + (size_t)((band || TSP) ? n : npad)), c->stream));
        c->S_zeroed = false;
        const int ninit = (std::max(npad, n) + 255) / 256;
        PrepArgs pa{c->d_red_boff.p, c->d_copy.p, c->lambda, ninit, (uint32_t)c->n_fast_groups, c->d_status.p, c->stamp_ptr()};
        const dim3 grid((unsigned)(c->n_fast_groups + ninit + c->ncopy));
#define LAUNCH_ALL(DV) hipLaunchKernelGGL((schur_elim_all_kernel<DV, LAY>), grid, dim3(256), 0, c->stream, c->A.p, c->b.p, c->d_elim_desc.p, c->d_elim_rc.p, c->Cinv.p, L, c->s_ptr(), (uint32_t)nfast_narrow, pa)
        if (c->fast_dv == 3) LAUNCH_ALL(3); else if (c->fast_dv == 2) LAUNCH_ALL(2); else LAUNCH_ALL(1);
#undef LAUNCH_ALL
        HIPCHK(hipGetLastError());
        return NLLS_OK;
    }
    c->status_known_zero = false;
    if (!one_prepare) HIPCHK(hipMemsetAsync(c->d_status.p, 0, sizeof(int32_t) * 5, c->stream));
    if (!c->S_zeroed) HIPCHK(hipMemsetAsync(c->S.p, 0, sizeof(double) * (c->s_elems + (size_t)((band || TSP) ? n : npad)), c->stream));
    c->S_zeroed = false;
    if (c->nranks > 1) HIPCHK(hipMemsetAsync(c->x.p, 0, sizeof(double) * c->info.ndof, c->stream));
    if (one_prepare) {
        const int ninit = (std::max(npad, n) + 255) / 256;
        hipLaunchKernelGGL(schur_prepare_kernel<LAY>, dim3((unsigned)(ninit + c->ncopy)), dim3(256), 0, c->stream, L, c->s_ptr(), c->b.p, c->d_red_boff.p,
                           c->A.p, c->d_copy.p, lambda_rr, ninit, c->d_status.p);
    } else if (lead) {
        hipLaunchKernelGGL(schur_init_kernel<LAY>, dim3((std::max(npad, n) + 255) / 256), dim3(256), 0, c->stream, L, c->s_ptr(), c->b.p, c->d_red_boff.p);
        if (c->info.is_sparse) {
            if (c->ncopy > 0) hipLaunchKernelGGL(schur_copy_kernel<LAY>, dim3((unsigned)c->ncopy), dim3(64), 0, c->stream, L, c->A.p, c->d_copy.p, lambda_rr);
        } else {
            const int64_t n2 = (int64_t)n * n;
            hipLaunchKernelGGL(dense_to_S_kernel, dim3((unsigned)((n2 + 255) / 256)), dim3(256), 0, c->stream, c->S.p, c->A.p, c->lambda, n, npad);
        }
    }
    if (c->nelim_groups > 0) {
        if (c->n_slow_groups > 0) {
            // (more than 64 KB of dynamic LDS has to be asked for once per process)
            static size_t lds_granted = 0; const size_t want = std::max(c->elim_lds_acc, c->elim_lds_noacc);
            if (want > lds_granted) { HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&schur_elim_kernel<LAY>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)want)); lds_granted = want; }
            const int64_t nacc = c->n_slow_acc, nno = c->n_slow_groups - nacc;
            if (nacc > 0) hipLaunchKernelGGL(schur_elim_kernel<LAY>, dim3((unsigned)nacc), dim3(64), c->elim_lds_acc, c->stream, c->A.p, c->b.p, c->d_elim_ptr.p, c->d_elim_nbr.p,
                               c->d_elim_diag.p, c->d_elim_boff.p, c->d_elim_dim.p, c->d_elim_group.p, c->d_slow_groups.p, c->lambda, c->max_elim_dim, c->slow_nd_acc, 1, L, c->s_ptr(), c->d_status.p);
            if (nno > 0) hipLaunchKernelGGL(schur_elim_kernel<LAY>, dim3((unsigned)nno), dim3(64), c->elim_lds_noacc, c->stream, c->A.p, c->b.p, c->d_elim_ptr.p, c->d_elim_nbr.p,
                               c->d_elim_diag.p, c->d_elim_boff.p, c->d_elim_dim.p, c->d_elim_group.p, c->d_slow_groups.p + nacc, c->lambda, c->max_elim_dim, c->slow_nd_noacc, 0, L, c->s_ptr(), c->d_status.p);
        }
#define LAUNCH_TILED(DV) do { const int64_t nel = (int64_t)c->d_elim_diag.n; \
            hipLaunchKernelGGL((schur_cinv_kernel<DV>), dim3((unsigned)((nel + 255) / 256)), dim3(256), 0, c->stream, c->A.p, c->d_elim_diag.p, c->d_elim_dim.p, nel, c->lambda, c->Cinv.p, c->d_status.p); \
            const int64_t n60 = c->n_fast_n60, nnar = c->n_fast_narrow - c->n_fast_n60, nwid = c->n_fast_groups - c->n_fast_narrow;   /* d_fast_groups: nd <= 60, then the other narrow supernodes, then the wide ones */ \
            if (c->elim_mfma && n60 + nnar > 0 && nwid > 0) { hipLaunchKernelGGL((schur_elim_fused_kernel<DV, LAY>), dim3((unsigned)(n60 + nnar + nwid)), dim3(256), 0, c->stream, c->A.p, c->b.p, \
                c->d_elim_desc.p, c->d_elim_rc.p, c->Cinv.p, L, c->s_ptr(), (uint32_t)(n60 + nnar)); break; } \
            if (c->elim_mfma) { if (n60 + nnar > 0) hipLaunchKernelGGL((schur_elim_mfma_kernel<DV, LAY>), dim3((unsigned)(n60 + nnar)), dim3(64 * ELIM_MFMA_NW), 0, c->stream, c->A.p, c->b.p, \
                c->d_elim_desc.p, c->d_elim_rc.p, c->Cinv.p, L, c->s_ptr()); } else { \
            if (n60 > 0) hipLaunchKernelGGL((schur_elim_tiled_kernel<DV, 1, 2, LAY>), dim3((unsigned)n60), dim3(192), 0, c->stream, c->A.p, c->b.p, \
                c->d_elim_desc.p, c->d_elim_rc.p, c->Cinv.p, L, c->s_ptr()); \
            if (nnar > 0) hipLaunchKernelGGL((schur_elim_tiled_kernel<DV, 1, 3, LAY>), dim3((unsigned)nnar), dim3(256), 0, c->stream, c->A.p, c->b.p, \
                c->d_elim_desc.p + n60, c->d_elim_rc.p, c->Cinv.p, L, c->s_ptr()); } \
            if (nwid > 0) hipLaunchKernelGGL((schur_elim_tiled_kernel<DV, 2, 3, LAY>), dim3((unsigned)nwid), dim3(256), 0, c->stream, c->A.p, c->b.p, \
                c->d_elim_desc.p + c->n_fast_narrow, c->d_elim_rc.p, c->Cinv.p, L, c->s_ptr()); } while (0)
        if (c->n_fast_groups > 0) {
            if (c->fast_dv == 3) LAUNCH_TILED(3); else if (c->fast_dv == 2) LAUNCH_TILED(2); else if (c->fast_dv == 1) LAUNCH_TILED(1);
        }
#undef LAUNCH_TILED
    }
    HIPCHK(hipGetLastError());
    return NLLS_OK;
}

int enqueue_solve_local(nlls_ctx* c) { return c->solve_mode == SOLVE_TSPARSE ? enqueue_solve_local_t<true>(c) : enqueue_solve_local_t<false>(c); }

// the reduced system itself: factorisation + both substitutions; its solution lands in s (c->s_ptr())
int enqueue_reduced_solve(nlls_ctx* c) {
    const int n = (int)c->nred; if (n == 0) return NLLS_OK;
    const SLayout L = make_layout(c); const int npad = L.npad, nblk = npad / NB;
    const bool band = c->solve_mode == SOLVE_BAND;
    if (c->solve_mode == SOLVE_SMALL) {
        hipLaunchKernelGGL(small_solve_kernel, dim3(1), dim3(64), 0, c->stream, c->S.p, c->s_ptr(), n, npad, c->d_status.p);
    } else if (c->solve_mode == SOLVE_TSPARSE) {
        // (an undamped step of a gauge-free problem: vanished pivots are dropped and counted, the band solver's rule -- see below)
        if (c->tsp.enqueue(c->stream, c->S.p, c->s_ptr(), c->d_status.p, c->lambda == 0.0 ? 1e-11 : c->damped_floor) != NLLS_OK) return herr(c, hipGetLastError(), "tile-sparse reduced solve launch");
    } else if (band && c->bcr.ready) {
        // an UNDAMPED step (Newton, dogleg's Gauss-Newton step) of a gauge-free problem: S is singular -- vanished pivots are dropped
        // (src/iterators.jl:47-115 asks for the Gauss-Newton step; any exact factorisation of a singular system returns rounding / rounding)
        // Rule: a pivot that has lost eleven orders of magnitude against its original diagonal entry is dropped (its unknown gets no step) and
        // COUNTED (status[4] -> nlls_get_solve_stats()[10]); a NaN pivot is not touched and is reported like any bad pivot.  Both assemblies
        // of the tiles (atomics, and the deterministic slab + gather) take it; the chain and dense solvers have no floor.
        // Round 5: the SAME rule under damping (c->damped_floor = 1e-11 unless NLLS_FLAG_NO_PIVOT_FLOOR).  A damped pivot is at least lambda, so the rule is silent while
        // lambda / |original diagonal| > 1e-11; below that -- the late iterations of a converged, gauge-free problem, lambda at 1e-19 of the diagonal -- the pivots of the
        // gauge directions are rounding noise of either sign, the step along them noise / noise, and whether the trial is accepted a coin toss: at BASELINE config 4
        // 29-31 damped solves for 20 iterations (the oracle's LDL': 24) against 20 with the rule, ending at the oracle's cost to 12 digits.
        const double pivot_floor = c->lambda == 0.0 ? 1e-11 : c->damped_floor;
        const bool tiles_direct = c->elim_slab || c->mf_use;          // (slab + gather assembly: the tiles are in place, no conversion from band storage)
        if (!tiles_direct) c->tiles_zeroed = false;
        if (c->bcr.enqueue(c->stream, tiles_direct ? (const double*)nullptr : c->S.p, c->s_ptr(), c->d_status.p, pivot_floor) != NLLS_OK) return herr(c, hipGetLastError(), "block cyclic reduction launch");
    } else if (band) {
        // bands wider than block cyclic reduction takes (more than 80 columns), and NLLS_FLAG_NO_BCR: the chain kernels of round 1 (nlls_chain.hip)
        const int rc = enqueue_chain_solve(c, L.n_band, L.bw, L.nbd, L.H); if (rc != NLLS_OK) return rc;
    } else {
        // blocked right-looking LDL', 64 columns at a time: the panel (diagonal block on the matrix cores with look-ahead, the rows below
        // it as tile products: dense_panel_kernel, nlls_bcr.hip) and the MFMA trailing update
        // workspace: W = L Delta of the current panel(s) (npad x 128) | acc (npad) | inv(L_JJ)' of every diagonal tile (backward pass)
        double* const Wbuf = c->Lwork.p; double* const accb = Wbuf + (size_t)npad * 2 * NB; double* LiD = accb + npad;
        double* const Dfac = LiD + (size_t)(npad / 16) * 256 + 256;    // the factored diagonal blocks, one slot of 128 x 128 per 64-block, until dense_dcopy_all_kernel moves them into S
        double* W0 = Wbuf; double* W1 = Wbuf + (size_t)npad * NB;
        int k = 0;
        if (c->dense_window) {
            // WINDOWED: the reduced system is a wide band (after the reverse Cuthill-McKee ordering of the upload) + border rows, stored densely.  The same
            // 128-column panels and trailing updates, restricted to what a banded LDL' touches: below panel p the 128-row blocks p + 1 .. whi - 1 (fill stays
            // inside the band: rows up to 128 p + 127 + bw) and the strip of border / right-hand-side rows at the bottom.  O(n w^2) instead of n^3 / 3.
            const int NB128 = npad / 128, strip128 = (int)(c->n_band / 128);
            for (int p = 0; p < NB128; ++p) {
                const int whi = std::min(NB128, (128 * p + 127 + c->bw) / 128 + 1);            // first 128-row block BEHIND the band of this panel
                DenseWin w; w.nwin = std::max(0, std::min(whi, NB128) - (p + 1)); w.strip = std::max(strip128, p + 1 + w.nwin); w.ntot = w.nwin + std::max(0, NB128 - w.strip);
                launch_dense_panel(c->stream, c->S.p, Wbuf, LiD, npad, p, c->d_status.p, 1, Dfac, w);
                if (w.ntot <= 0) continue;
                if (w.ntot >= c->dense_t128_min) hipLaunchKernelGGL(syrk_update128_kernel, dim3(w.ntot * (w.ntot + 1) / 2), dim3(512), 0, c->stream, c->S.p, W0, W1, npad, 2 * p, 2 * p + 2, 0, w.nwin, w.strip);
                else { const int T = 2 * w.ntot; hipLaunchKernelGGL(syrk_update2_kernel<2>, dim3(T * (T + 1) / 2), dim3(256), 0, c->stream, c->S.p, W0, W1, npad, 2 * p, 2 * p + 2, 0, 2 * w.nwin, 2 * w.strip); }
            }
            launch_dense_dcopy_all(c->stream, c->S.p, Dfac, npad, NB128, 2 * NB128, 0);
            k = nblk;
        } else if (c->dense_t128) {
            // 128-column panels (dense_panel_kernel<8, 2>: one launch factors what used to be panel k, a narrow update of block column k + 1 and
            // panel k + 1), each followed by ONE update of everything behind it with K = 128 (128 x 128 tiles; 64 x 64 for the small tail)
            for (; k + 1 < nblk; k += 2) {
                launch_dense_panel(c->stream, c->S.p, Wbuf, LiD, npad, k / 2, c->d_status.p, 1, Dfac);
                const int T = nblk - k - 2;
                if (T <= 0) continue;
                const int T128 = (T + 1) / 2;
                if (T128 >= c->dense_t128_min) hipLaunchKernelGGL(syrk_update128_kernel, dim3(T128 * (T128 + 1) / 2), dim3(512), 0, c->stream, c->S.p, W0, W1, npad, k, k + 2, 0);
                else hipLaunchKernelGGL(syrk_update2_kernel<2>, dim3(T * (T + 1) / 2), dim3(256), 0, c->stream, c->S.p, W0, W1, npad, k, k + 2, 0);
            }
            const int nwide = k / 2;
            if (k < nblk) { launch_dense_panel(c->stream, c->S.p, W0, LiD, npad, k, c->d_status.p, 0, Dfac); ++k; }     // an odd last 64-column panel: nothing behind it
            launch_dense_dcopy_all(c->stream, c->S.p, Dfac, npad, nwide, 2 * nwide, nblk - 2 * nwide);
        } else {
            // two 64-column panels per pass: panel k, a NARROW update of block column k + 1 only, panel k + 1, then one update with both (K = 128)
            for (; k < nblk; k += 2) {
                launch_dense_panel(c->stream, c->S.p, W0, LiD, npad, k, c->d_status.p, 0, Dfac);
                if (k + 1 >= nblk) break;
                hipLaunchKernelGGL(syrk_update2_kernel<1>, dim3(nblk - k - 1), dim3(256), 0, c->stream, c->S.p, W0, W0, npad, k, k + 1, 1);
                launch_dense_panel(c->stream, c->S.p, W1, LiD, npad, k + 1, c->d_status.p, 0, Dfac);
                const int T = nblk - k - 2;
                if (T > 0) hipLaunchKernelGGL(syrk_update2_kernel<2>, dim3(T * (T + 1) / 2), dim3(256), 0, c->stream, c->S.p, W0, W1, npad, k, k + 2, 0);
            }
            launch_dense_dcopy_all(c->stream, c->S.p, Dfac, npad, 0, 0, nblk);
        }
        // backward substitution into acc / s (x)
        if (c->dense_fused_bwd) {
            // ONE launch for the whole substitution (+ one for the explicit inverses of the 128 x 128 diagonal blocks, into the slots the factored
            // diagonal blocks have just left, and the sentinel in x)
            launch_dense_bwd_fused(c->stream, c->S.p, LiD, Dfac, npad, n, c->s_ptr(), c->d_status.p);
            HIPCHK(hipGetLastError());
            return NLLS_OK;
        }
        double* acc = accb;
        HIPCHK(hipMemsetAsync(acc, 0, sizeof(double) * npad, c->stream));
        const int nb_real = (n + NB - 1) / NB;
        // one launch per block: the last block alone, then "push block s into the blocks above it and solve block s - 1"
        // (two blocks per launch -- every workgroup solving block s - 1 redundantly -- was measured: 15.9 us per launch against 2 x 6.7)
        launch_dense_bwd_diag(c->stream, c->S.p, LiD, npad, nb_real - 1, n, acc, c->s_ptr());
        for (int sblk = nb_real - 1; sblk >= 1; --sblk) launch_dense_bwd_step(c->stream, c->S.p, LiD, npad, sblk, n, acc, c->s_ptr());
    }
    HIPCHK(hipGetLastError());
    return NLLS_OK;
}

// finish: factor the (summed) reduced system, solve it, back-substitute this rank's eliminated blocks
int enqueue_solve_finish(nlls_ctx* c) {
    c->retract_done = false;
    const int n = (int)c->nred; if (n == 0) return NLLS_OK;
    { const int rc = enqueue_reduced_solve(c); if (rc != NLLS_OK) return rc; }
    if (c->phase_on && c->phase_ev.size() >= 6) (void)hipEventRecord(c->phase_ev[3], c->stream);      // (phase timing: the reduced solve ends)
    // x = -solution (folded into the fast back-substitution launch when there is one)
    // (replicate_xr: a sharded LM trial keeps the reduced part of the step on every rank -- each retracts the cameras and its own
    //  points itself, no all-reduce of x)
    const int write_red = (c->nranks == 1 || c->rank == 0 || c->replicate_xr) ? 1 : 0;
    if (c->n_fast_groups == 0) hipLaunchKernelGGL(scatter_reduced_kernel, dim3((n + 255) / 256), dim3(256), 0, c->stream, c->s_ptr(), c->d_red_boff.p, n, c->x.p, write_red);
    const int64_t nel_local = (int64_t)(c->d_elim_diag.n);
    if (nel_local > 0) {
        const size_t lds = sizeof(double) * ((size_t)c->max_elim_dim * c->max_elim_dim + c->max_elim_dim);
        // members of fast-path supernodes reuse the inverses of schur_cinv_kernel; the rest factor their own block
        const int64_t nslow = c->n_fast_groups > 0 ? (int64_t)c->d_slow_blocks.n : nel_local;
        if (nslow > 0)
            hipLaunchKernelGGL(schur_backsub_kernel, dim3((unsigned)nslow), dim3(64), lds, c->stream, c->A.p, c->b.p, c->d_elim_ptr.p, c->d_elim_nbr.p,
                               c->d_elim_diag.p, c->d_elim_boff.p, c->d_elim_dim.p, c->n_fast_groups > 0 ? c->d_slow_blocks.p : (const uint32_t*)nullptr,
                               c->lambda, c->max_elim_dim, c->s_ptr(), c->x.p);
        // (single rank, one_prepare path: s is rewritten in full by schur_prepare_kernel, so only S itself has to be zero)
        const bool zero_S = c->nranks == 1 && c->solve_mode == SOLVE_BAND && c->info.is_sparse && c->ncopy > 0;
        const unsigned nextra = zero_S ? 160 : 32;
        // what the spare workgroups zero-fill for the next solve: the band storage of S, or (slab + gather assembly) the tiles the gather writes into
        double* zptr = c->S.p; int64_t zcount = zero_S ? (int64_t)c->s_elems : (int64_t)0;
        const bool tiles_direct = c->elim_slab || c->mf_use;
        if (tiles_direct) { const BcrGeom& g = c->bcr.geom; zptr = g.ws + g.oD; zcount = (int64_t)(g.oBR + (size_t)g.N * g.NT * 256 - g.oD); }
        // an LM trial (nlls_lm_trial sets trial_to / trial_from): the retraction in this launch
        BsfRetract rt{}; unsigned nrestwg = 0; rt.stamps = c->stamp_ptr();
        c->retract_done = false;
        if (c->trial_to >= 0 && (c->post_fuse || c->mf_use) && c->fast_all_euclid && c->nranks == 1 && nslow == 0 && c->info.is_sparse && c->n_fast_groups > 0 && c->info.nvar > 0) {
            rt.on = 1; rt.nrest = (int)c->d_rest_var.n; rt.fast_voff = c->d_fast_voff.p; rt.rest_var = c->d_rest_var.p; rt.rest_red = c->d_rest_red.p;
            rt.vkind = c->d_var_kind.p; rt.vdim = c->d_var_dim.p; rt.voff = c->d_var_off.p; rt.vfrom = vars_ptr(c, c->trial_from); rt.vto = vars_ptr(c, c->trial_to);
            nrestwg = (unsigned)((rt.nrest + 63) / 64); c->retract_done = true;
        }
        if (c->mf_use) {
            if (!rt.on) { c->err = "matrix-free trial without the fused retraction"; return NLLS_ERR_NOT_READY; }
            const int rc = enqueue_mf_backsub(c, rt, write_red, zptr, zcount, nextra, nrestwg); if (rc != NLLS_OK) return rc;
            c->tE_valid = false; c->mf_step = true; c->tiles_zeroed = true;
            return NLLS_OK;
        }
        c->mf_step = false;
#define LAUNCH_BSF(DV) hipLaunchKernelGGL((schur_backsub_fast_kernel<DV>), dim3((unsigned)c->n_fast_groups + nextra + nrestwg), dim3(64), 0, c->stream, c->A.p, c->b.p, c->d_elim_desc.p, c->d_elim_rc.p, \
                c->Cinv.p, c->s_ptr(), c->x.p, c->tE.p, (uint32_t)c->n_fast_groups, c->d_red_boff.p, n, write_red, zptr, zcount, nextra, rt)
        if (c->n_fast_groups > 0) { if (c->fast_dv == 3) LAUNCH_BSF(3); else if (c->fast_dv == 2) LAUNCH_BSF(2); else if (c->fast_dv == 1) LAUNCH_BSF(1); c->tE_valid = true; if (tiles_direct) c->tiles_zeroed = true; else c->S_zeroed = zero_S; }
        else c->retract_done = false;
#undef LAUNCH_BSF
    }
    HIPCHK(hipGetLastError());
    return NLLS_OK;
}

// the end of that trial: the cost partials' sum (the order of reduce_partials_kernel), and the scalars published to the pinned host mirror as trial_finish_kernel does
__global__ __launch_bounds__(TPB) void tiny_trial_finish_kernel(DenseFin fin) {
    __shared__ double red[TPB / 64];
    dense_fin_body(fin, red);
}
// the small dense system's LM trial: damped solve + statistics (+ retraction) in one launch, then the cost sweep; scalars[0..10] as enqueue_lm_trial_tail leaves them
int enqueue_tiny_trial_finish_pending(nlls_ctx* c) {
    if (!c->dense_fin_pending) return NLLS_OK;
    c->dense_fin_pending = false;
    hipLaunchKernelGGL(tiny_trial_finish_kernel, dim3(1), dim3(TPB), 0, c->stream, c->dense_fin);
    HIPCHK(hipGetLastError());
    return NLLS_OK;
}
int enqueue_tiny_dense_trial(nlls_ctx* c, int to, int from, bool lookahead_follows) {
    const int n = (int)c->info.ndof; const int64_t nvar = c->info.nvar;
    const bool fused_retract = nvar <= TINY_RETRACT_MAX;
    c->status_known_zero = false; c->retract_done = false;
    hipLaunchKernelGGL(tiny_dense_trial_kernel, dim3(1), dim3(64), 0, c->stream, c->A.p, c->b.p, c->d_red_boff.p, c->lambda, n, c->x.p, c->scalars.p, c->d_status.p,
                       c->d_var_kind.p, c->d_var_dim.p, c->d_var_off.p, c->d_var_boff.p, fused_retract ? nvar : (int64_t)0, vars_ptr(c, from), vars_ptr(c, to));
    HIPCHK(hipGetLastError());
    if (!fused_retract) { const int rc = enqueue_retract(c, to, from); if (rc != NLLS_OK) return rc; }
    int64_t ncp = 0;
    { const int rc = enqueue_sweep_cost(c, to, TRIAL_COST_POFS, &ncp); if (rc != NLLS_OK) return rc; }
    c->dense_fin = DenseFin{c->partials.p + TRIAL_COST_POFS, ncp, c->scalars.p, c->h_scalars_dev, (double)(++c->trial_seq)}; c->dense_fin_pending = true;
    if (lookahead_follows) return NLLS_OK;                        // (the look-ahead sweep's accumulate launch carries the finishing reduction; nlls_lm_trial launches what is left)
    return enqueue_tiny_trial_finish_pending(c);
}

int enqueue_solve(nlls_ctx* c) {
    int rc = enqueue_solve_local(c); if (rc != NLLS_OK) return rc;
    return enqueue_solve_finish(c);
}

}  // namespace nlls
