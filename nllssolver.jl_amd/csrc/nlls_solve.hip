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

namespace nlls {

constexpr int NB = 64;          // Cholesky panel width
constexpr int LDT = 80;         // LDS leading dimension of a 64-row operand tile (80 = 16 mod 32: conflict-free ds_read_b64)

typedef double double4_t __attribute__((ext_vector_type(4)));

NLLS_DEV double wsum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

// ---------------------------------------------------------------------------------------------------
// reduced system storage.  S is addressed by its lower triangle (i >= j) in reduced order
// [banded part (n_band dof) | border dof (nbd) | rhs row]; row n (= n_band + nbd) carries the right-hand side.
//   dense: col-major, ld = npad                         (general systems; MFMA blocked LDL')
//   band : column j of the banded part holds H = bw+1+nbd+1 entries [S(j..j+bw, j) | S(border, j) | s(j)],
//          followed by the (nbd+1)^2 border corner        (narrow-band systems; persistent-workgroup LDL')
//   tile-sparse: the lower tiles of the filled tile pattern, 128 x 128 column-major each, in a nested-dissection order of its own (nlls_tsp.hip):
//          tsp[0, n) = position of a reduced unknown in that order, tsp[n + I npad + J] = slot of tile (I, J), I >= J  (npad = number of tiles)
// ---------------------------------------------------------------------------------------------------
// The tile-sparse addressing is a TYPE of its own (SLayoutT<true>: one more pointer, two dependent loads per entry): every kernel that assembles [S | s] is
// instantiated for both, and the band / dense instantiations are byte for byte what they were without it (with the branch inside one struct the elimination
// launch of BASELINE config 4 was measured 4 us slower -- registers, not the branch).
struct SLayoutNoMap {}; struct SLayoutMap { const int32_t* tsp; };
template <bool TSP>
struct SLayoutT : std::conditional_t<TSP, SLayoutMap, SLayoutNoMap> {
    double* S; int mode; int n, npad, n_band, bw, nbd, H;
    NLLS_DEV double* at(int i, int j) const {   // i >= j
        if constexpr (TSP) { int pi = this->tsp[i], pj = this->tsp[j]; if (pi < pj) { const int t = pi; pi = pj; pj = t; }      // (the tile order is not the reduced order: the entry lives at (max, min) of the POSITIONS)
            return S + (size_t)this->tsp[n + (pi >> 7) * npad + (pj >> 7)] * (128 * 128) + (pi & 127) + 128 * (pj & 127); }
        if (mode != SOLVE_BAND) return S + (size_t)i + (size_t)npad * j;
        if (i < n_band) return S + (size_t)j * H + (i - j);
        if (j < n_band) return S + (size_t)j * H + bw + 1 + (i - n_band);
        return S + (size_t)n_band * H + (i - n_band) + (size_t)(nbd + 1) * (j - n_band);
    }
    // entry i of the reduced right-hand side while the system is assembled: the factorisations carry it as row n of S
    // (band and dense layouts); only the one-wave solver of tiny systems reads it from the vector s
    NLLS_DEV double* rhs(double* s, int i) const { if constexpr (TSP) return s + i; else return mode == SOLVE_SMALL ? s + i : at(n, i); }
};
using SLayout = SLayoutT<false>;
template <bool T> NLLS_HD constexpr bool LAY_IS_TSP(const SLayoutT<T>&) { return T; }

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
struct BsfRetract { int on, nrest; const uint32_t* fast_voff; const uint32_t* rest_var; const int32_t* rest_red;
                    const int32_t* vkind; const int32_t* vdim; const uint32_t* voff; const double* vfrom; double* vto; };
template <int DV>
__global__ __launch_bounds__(64) void schur_backsub_fast_kernel(const double* __restrict__ A, const double* __restrict__ b,
                                                                const ElimDesc* __restrict__ desc, const uint32_t* __restrict__ rcflat,
                                                                const double* __restrict__ Cinv, const double* __restrict__ xr, double* __restrict__ x, double* __restrict__ tE,
                                                                uint32_t ngroups, const uint32_t* __restrict__ red_boff, int nred, int write_red,
                                                                double* __restrict__ Szero, int64_t nzero, uint32_t nextra, BsfRetract rt) {
    constexpr int MAXC = (72 + 15) / 16;                      // columns per lane (nd <= 72)
    __shared__ uint32_t rc[80];
    const int lane = threadIdx.x, l = lane & 15, gsub = lane >> 4;
    if (blockIdx.x >= ngroups + nextra) {                     // (only with rt.on) one thread per variable that is no member of a fast supernode
        const int j = (int)(blockIdx.x - ngroups - nextra) * 64 + lane; if (j >= rt.nrest) return;
        const uint32_t i = rt.rest_var[j]; const int r0 = rt.rest_red[j]; const int k = rt.vkind[i], d = rt.vdim[i]; const uint32_t o = rt.voff[i];
        if (r0 < 0) { const int st = var_storage(k, d); for (int q = 0; q < st; ++q) rt.vto[o + q] = rt.vfrom[o + q]; return; }      // fixed: copied
        retract_var_fn(k, d, o, rt.vfrom, rt.vto, [&](int q) { return write_red ? -xr[r0 + q] : 0.0; });     // (no staging array: it lived in scratch memory, 1040 bytes per lane of this launch)
        return;
    }
    if (blockIdx.x >= ngroups) {                              // the workgroups behind the supernodes scatter the reduced part: x_R = -s
        for (int i = (blockIdx.x - ngroups) * 64 + lane; i < nred; i += nextra * 64) x[red_boff[i]] = write_red ? -xr[i] : 0.0;
        // ... and leave the reduced system's storage zero-filled for the next solve (nothing reads S any more)
        for (int64_t i = (int64_t)(blockIdx.x - ngroups) * 64 + lane; i < nzero; i += (int64_t)nextra * 64) Szero[i] = 0.0;
        return;
    }
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
struct PrepArgs { const uint32_t* red_boff; const SchurCopy* copies; double lambda; int ninit; uint32_t nfast; int* status; };
template <int DV, class LAY = SLayout>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void schur_elim_all_kernel(const double* __restrict__ A, const double* __restrict__ b,
                                                              const ElimDesc* __restrict__ desc, const uint32_t* __restrict__ rcflat,
                                                              double* __restrict__ Cinv, LAY L, double* __restrict__ s, uint32_t nnarrow, PrepArgs pa) {
    if (blockIdx.x >= pa.nfast) {
        const int w = (int)(blockIdx.x - pa.nfast);
        if (w == 0 && threadIdx.x == 0) pa.status[4] = 0;                       // (pivots dropped by the floor: only the panels of this solve add to it)
        if (w < pa.ninit) {
            const int i = w * 256 + threadIdx.x;
            if (i < L.n) { atomicAdd(L.rhs(s, i), b[pa.red_boff[i]]); return; }
            if (!LAY_IS_TSP(L) && L.mode != SOLVE_BAND && i < L.npad) { s[i] = 0.0; L.S[(size_t)i + (size_t)L.npad * i] = (i == L.n) ? 1e300 : 1.0; }
            return;
        }
        const SchurCopy cp = pa.copies[w - pa.ninit];
        for (int e = threadIdx.x; e < cp.rows * cp.cols; e += 256) {
            const int i = e % cp.rows, j = e / cp.rows;
            double v = A[cp.off + e];
            if (cp.r == cp.c) { if (i < j) continue; if (i == j) v += pa.lambda; atomicAdd(L.at(cp.r + i, cp.c + j), v); }
            else if (cp.r > cp.c) atomicAdd(L.at(cp.r + i, cp.c + j), v);
            else atomicAdd(L.at(cp.c + j, cp.r + i), v);
        }
        return;
    }
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
__global__ __launch_bounds__(64) void small_solve_kernel(const double* __restrict__ S, double* __restrict__ s, int n, int npad, int* __restrict__ status) {
    __shared__ double M[64 * 65]; __shared__ double rhs[64]; __shared__ int piv; __shared__ int ok;
    const int t = threadIdx.x;
    for (int e = t; e < n * n; e += 64) { const int i = e % n, j = e / n; M[i + 65 * j] = (i >= j) ? S[(size_t)i + (size_t)npad * j] : S[(size_t)j + (size_t)npad * i]; }
    if (t < n) rhs[t] = s[t];
    if (t == 0) ok = 1;
    __syncthreads();
    // Cholesky (right-looking), thread = row
    for (int j = 0; j < n; ++j) {
        const double d = M[j + 65 * j];
        if (!(d > 0)) { if (t == 0) ok = 0; }
        __syncthreads();
        if (!ok) break;
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
    if (ok) {
        if (t == 0) {
            for (int i = 0; i < n; ++i) { double v = rhs[i]; for (int k = 0; k < i; ++k) v -= M[i + 65 * k] * rhs[k]; rhs[i] = v / M[i + 65 * i]; }
            for (int i = n - 1; i >= 0; --i) { double v = rhs[i]; for (int k = i + 1; k < n; ++k) v -= M[k + 65 * i] * rhs[k]; rhs[i] = v / M[i + 65 * i]; }
        }
        __syncthreads();
        if (t < n) s[t] = rhs[t];
        return;
    }
    // LU with partial pivoting on a fresh symmetric copy
    __syncthreads();
    for (int e = t; e < n * n; e += 64) { const int i = e % n, j = e / n; M[i + 65 * j] = (i >= j) ? S[(size_t)i + (size_t)npad * j] : S[(size_t)j + (size_t)npad * i]; }
    if (t < n) rhs[t] = s[t];
    __syncthreads();
    for (int j = 0; j < n; ++j) {
        if (t == 0) { int p = j; double best = fabs(M[j + 65 * j]); for (int i = j + 1; i < n; ++i) { double a = fabs(M[i + 65 * j]); if (a > best) { best = a; p = i; } } piv = p; if (best == 0.0) atomicCAS(status, 0, 2); }
        __syncthreads();
        const int p = piv;
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
    if (t < n) s[t] = rhs[t];
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
// bordered-band LDL' + both triangular solves in ONE persistent workgroup (narrow-band reduced systems,
// e.g. the camera chain of sequential bundle adjustment: 6000 dof, half bandwidth 65).
// A banded factorisation is a chain of n dependent pivots: latency- not throughput-bound.  Design:
//  * the active window (columns j+1..j+bw) lives in REGISTERS: lane t, slot q owns SEG consecutive entries of one
//    window column for that column's whole life (bw pivots), so the rank-1 update costs one LDS read + one FMA per
//    entry and no LDS write; only the next pivot column is published to LDS (double-buffered), one barrier per pivot;
//  * border rows (dense rows ordered last) and the rhs ride along as extra rows (tiny LDS arrays);
//  * columns stream in from HBM through an LDS ring, prefetched into registers one chunk ahead behind an LDS-only
//    barrier (a __syncthreads() would drain vmcnt and put HBM latency on the per-pivot critical path);
//  * the backward pass runs in "axpy" form on ONE wave: each lane keeps the partially reduced unknowns of its rows
//    in registers, x_i is broadcast with v_readlane, no cross-lane reduction; waves 1-3 stage the factor.
// ---------------------------------------------------------------------------------------------------
struct BandArgs { const double* Sb; double* Lb; double* xr; int n_band, bw, nbd, H, CH, PFC, RC, NSC; int* status; };

template <int SEG, int NSLOT>
__global__ __launch_bounds__(256) void band_ldlt_solve_kernel(BandArgs a) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n_band = a.n_band, bw = a.bw, nbd = a.nbd, H = a.H, CH = a.CH, RC = a.RC, nbr = nbd + 1, NSC = a.NSC;
    const int RCW = bw + 1;                       // window columns (the pivot column + bw columns it updates)
    const int PV = 2 * NSC * SEG + 2 * SEG;       // published pivot column: band entries, then zeros (covers dc + e0 + SEG of any slot)
    double* W = sm;                               // RC * H landing ring of columns in global layout [band | border | rhs]
    double* piv = W + (size_t)RC * H;             // 2 * PV   published pivot columns (double-buffered)
    const int BWS = RCW + 1;                      // one spare slot: the entering column is written while the pivot's is still read
    double* Bw = piv + 2 * PV;                    // BWS * nbr  border rows + rhs of the window columns (slot = column % BWS)
    double* Cl = Bw + (size_t)BWS * nbr;          // nbr x nbr border corner (col-major, lower), last row = rhs
    double* xb = Cl + nbr * nbr;                  // nbr
    const double* corner_g = a.Sb + (size_t)n_band * H;
    for (int e = tid; e < nbr * nbr; e += 256) Cl[e] = corner_g[e];
    for (int e = tid; e < 2 * PV; e += 256) piv[e] = 0.0;
    // ---- ownership: pair id = tid + 256 q -> (window column slot cs, segment s)
    int own_cs[NSLOT], own_s[NSLOT];
#pragma unroll
    for (int q = 0; q < NSLOT; ++q) { const int pid = tid + 256 * q; own_cs[q] = (pid < RCW * NSC) ? pid / NSC : -1; own_s[q] = pid % NSC; }
    const int ncorner = nbr * (nbr + 1) / 2;
    int cr = 0, cr2 = 0;                          // this lane's corner element (r >= r2), lanes < ncorner
    { int it = tid; while (cr2 < nbr && it >= nbr - cr2) { it -= nbr - cr2; ++cr2; } cr = cr2 + it; }
    // ---- initial landing ring: chunks 0 .. PFC-1
    const int chunk_elems = CH * H;
    for (int m = 0; m < a.PFC; ++m) {
        const int c0 = m * CH;
        for (int idx = tid; idx < chunk_elems; idx += 256) { const int c2 = c0 + idx / H; W[(size_t)(c0 % RC) * H + idx] = (c2 < n_band) ? a.Sb[(size_t)c0 * H + idx] : 0.0; }
    }
    __syncthreads();
    // ---- initial window: columns 0..bw into registers / Bw, column 0 published
    double reg[NSLOT][SEG];
#pragma unroll
    for (int q = 0; q < NSLOT; ++q) {
        const int cs = own_cs[q];                 // column c = cs for the first window
#pragma unroll
        for (int m = 0; m < SEG; ++m) { const int e = own_s[q] * SEG + m; reg[q][m] = (cs >= 0 && e <= bw && cs < n_band) ? W[(size_t)(cs % RC) * H + e] : 0.0; }
        if (cs == 0) {
#pragma unroll
            for (int m = 0; m < SEG; ++m) piv[own_s[q] * SEG + m] = reg[q][m];
        }
    }
    for (int e = tid; e < RCW * nbr; e += 256) { const int c2 = e / nbr, r = e - c2 * nbr; Bw[e] = (c2 < n_band) ? W[(size_t)(c2 % RC) * H + bw + 1 + r] : 0.0; }
    double pf[12];
    int jb = 0, jc = 0, mchunk = 0, pb = 0;                     // j % BWS, j % CH, j / CH, pivot buffer
    int wnew = RCW % RC;                                        // ring slot of column j + RCW
    int dcq[NSLOT], e0q[NSLOT];                                 // per slot: column offset from the pivot, first entry
#pragma unroll
    for (int q = 0; q < NSLOT; ++q) { dcq[q] = own_cs[q] >= 0 ? own_cs[q] : -(1 << 28); e0q[q] = own_s[q] * SEG; }
    const unsigned long long t_begin = __builtin_amdgcn_s_memtime();   // diagnostics only (nlls_get_solve_stats)
    for (int j = 0; j < n_band; ++j) {
        // the pivot column j is published.  Raw barrier behind an LDS-only wait (see header).
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        const double* col = piv + pb * PV;        // band entries of column j (entry 0 = d), zeros behind
        double* nxt = piv + (pb ^ 1) * PV;
        const double* bcol = Bw + (size_t)jb * nbr;   // border rows + rhs of column j
        double d = col[0];
        if (d == 0.0 || d != d) { if (tid == 0) atomicCAS(a.status, 0, 1 + j); d = 1.0; }
        const double id = 1.0 / d;
        if (jc == 0) {                                         // issue the prefetch of a chunk PFC ahead (registers)
            const int c0 = (mchunk + a.PFC) * CH;
#pragma unroll
            for (int k = 0; k < 12; ++k) { const int idx = tid + 256 * k; const int c2 = c0 + idx / H;
                pf[k] = (idx < chunk_elems && c2 < n_band) ? a.Sb[(size_t)c0 * H + idx] : 0.0; }
        }
        // ---- rank-1 update of the register window:  (column j+dc)[e] -= col[dc+e] * col[dc] / d   (branch-free:
        //      an inactive slot multiplies by 0; sources past the band part are the zero pad of the published column)
#pragma unroll
        for (int q = 0; q < NSLOT; ++q) {
            const int dc = dcq[q], e0 = e0q[q];
            const int dci = dc > 0 ? dc : 0;
            const double cdc = col[dci];
            const double l = (dc > 0 && e0 <= bw - dc) ? cdc * id : 0.0;
            const double* src = col + dci + e0;
#pragma unroll
            for (int m = 0; m < SEG; ++m) reg[q][m] = fma(-src[m], l, reg[q][m]);
            if (dc == 1) {                                      // next pivot column: publish
#pragma unroll
                for (int m = 0; m < SEG; ++m) nxt[e0 + m] = reg[q][m];
            }
            if (dc == 0) {                                      // the pivot column's slot now takes column j + RCW
                const bool have = j + RCW < n_band; const double* wsrc = W + (size_t)wnew * H + e0;
#pragma unroll
                for (int m = 0; m < SEG; ++m) reg[q][m] = (have && e0 + m <= bw) ? wsrc[m] : 0.0;
            }
            dcq[q] = (dc == 0) ? bw : dc - 1;
        }
        // ---- border rows + rhs of the window columns (lanes 0..bw-1), border corner (lanes < ncorner)
        if (tid < bw) {
            const int dc = tid + 1; const double l = col[dc] * id;
            int cs = jb + dc; if (cs >= BWS) cs -= BWS;
            double* dst = Bw + (size_t)cs * nbr;
            for (int r = 0; r < nbr; ++r) dst[r] -= bcol[r] * l;
        }
        if (tid < ncorner) Cl[cr + nbr * cr2] -= bcol[cr] * bcol[cr2] * id;
        // ---- factor column j: D on top, L below (fire and forget)
        if (tid < H) a.Lb[(size_t)j * H + tid] = (tid == 0) ? d : (tid <= bw ? col[tid] : bcol[tid - bw - 1]) * id;
        // the spare border slot takes column j + RCW (first touched at the next pivot)
        if (tid >= 64 && tid < 64 + nbr) { int cs = jb + RCW; if (cs >= BWS) cs -= BWS; Bw[(size_t)cs * nbr + (tid - 64)] = (j + RCW < n_band) ? W[(size_t)wnew * H + bw + 1 + (tid - 64)] : 0.0; }
        if (jc == CH - 1) {                                    // land the prefetched chunk: its ring slots held columns already in registers
            const int c0 = (mchunk + a.PFC) * CH;
#pragma unroll
            for (int k = 0; k < 12; ++k) { const int idx = tid + 256 * k; if (idx < chunk_elems) W[(size_t)(c0 % RC) * H + idx] = pf[k]; }
        }
        if (++jb == BWS) jb = 0;
        if (++wnew == RC) wnew = 0;
        if (++jc == CH) { jc = 0; ++mchunk; }
        pb ^= 1;
    }
    __syncthreads();
    const unsigned long long t_factor = __builtin_amdgcn_s_memtime();
    // ---- border corner: LDL' of the nbd x nbd block with the rhs row riding along, then the border unknowns
    if (tid == 0) {
        for (int j = 0; j < nbd; ++j) {
            double d = Cl[j + nbr * j];
            if (d == 0.0 || d != d) { atomicCAS(a.status, 0, 1 + n_band + j); d = 1.0; }
            for (int c2 = j + 1; c2 < nbd; ++c2) { const double f = Cl[c2 + nbr * j] / d; for (int i = c2; i < nbr; ++i) Cl[i + nbr * c2] -= Cl[i + nbr * j] * f; }
            for (int i = j + 1; i < nbr; ++i) Cl[i + nbr * j] /= d;
            Cl[j + nbr * j] = d;
        }
        for (int r = nbd - 1; r >= 0; --r) { double v = Cl[nbd + nbr * r]; for (int r2 = r + 1; r2 < nbd; ++r2) v -= Cl[r2 + nbr * r] * xb[r2]; xb[r] = v; a.xr[n_band + r] = v; }
    }
    __threadfence();
    __syncthreads();
    // ---- backward pass  L' x = z  (unit diagonal), rows n_band-1 .. 0, in axpy form on wave 0.
    // Lane l keeps the partially reduced unknown of its rows r = l (mod 64) in registers, one per 64-row block
    // (block index mod 3: the window [i-bw, i] spans at most three blocks for bw <= 127).  Waves 1-3 stage the factor
    // columns (global layout) into the ring W, one chunk ahead; one barrier per chunk of CH rows.
    const int M = (n_band + CH - 1) / CH;
    auto stage = [&](int m, int t0, int nt) {
        if (m < 0) return;
        const int c0 = m * CH; const int ncol = min(CH, n_band - c0);
        double* dst = W + (size_t)(c0 % RC) * H; const size_t g0 = (size_t)c0 * H;
        for (int idx = t0; idx < ncol * H; idx += nt) dst[idx] = a.Lb[g0 + idx];
    };
    const int PFB = a.PFC;                                     // chunks that must be resident below the current one
    for (int m = M - 1; m >= M - PFB && m >= 0; --m) stage(m, tid, 256);
    __syncthreads();
    // za/zb/zc: this lane's rows in blocks B, B-1, B-2 of the current row i (rotated when i crosses a 64-row block)
    double za = 0, zb = 0, zc = 0;
    auto zinit = [&](int ringslot) { const double* c2 = W + (size_t)ringslot * H; double v = c2[bw + 1 + nbd]; for (int q = 0; q < nbd; ++q) v -= c2[bw + 1 + q] * xb[q]; return v; };
    if (wave == 0) {   // rows [n_band-1-bw, n_band-1] start as z = rhs entry - border part; later rows are initialised when they enter the window
        const int Btop = (n_band - 1) >> 6;
        for (int r = n_band - 1; r >= max(0, n_band - 1 - bw); --r) if ((r & 63) == lane) { const double v = zinit(r % RC); const int k = Btop - (r >> 6); if (k == 0) za = v; else if (k == 1) zb = v; else zc = v; }
    }
    int rin_slot = ((n_band - 2 - bw) % RC + RC) % RC;          // ring slot of the row entering the window next (i - 1 - bw)
    for (int m = M - 1; m >= 0; --m) {
        if (wave == 0) {
            const int hi = min(n_band, (m + 1) * CH) - 1;
            int i = hi;
            while (i >= m * CH) {
                const int B = i >> 6; const int lo = max(m * CH, B << 6);       // rows [lo, i] share block B
                // per-lane LDS offsets of L(i, r_k), r_k = 64 (B - k) + lane: they decrease by one entry per step
                int e_a = i - ((B << 6) + lane), e_b = e_a + 64, e_c = e_a + 128;
                const int ra = (B << 6) + lane, rb = ra - 64, rc3 = ra - 128;
                const double* pa = W + (size_t)((ra % RC + RC) % RC) * H + e_a;
                const double* pbp = W + (size_t)((rb % RC + RC) % RC) * H + e_b;
                const double* pc = W + (size_t)((rc3 % RC + RC) % RC) * H + e_c;
                for (; i >= lo; --i) {
                    const int li = i & 63;
                    const double xi = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(za), li), __builtin_amdgcn_readlane(__double2loint(za), li));
                    if (lane == li) a.xr[i] = xi;
                    const double la = (e_a >= 1 && e_a <= bw) ? *pa : 0.0;
                    const double lb = (e_b <= bw && rb >= 0) ? *pbp : 0.0;
                    const double lc = (e_c <= bw && rc3 >= 0) ? *pc : 0.0;
                    za = fma(-la, xi, za); zb = fma(-lb, xi, zb); zc = fma(-lc, xi, zc);
                    // the row entering the window at the next step: rin = i - 1 - bw
                    const int rin = i - 1 - bw;
                    if (rin >= 0 && (rin & 63) == lane) { const double v = zinit(rin_slot); const int k = B - (rin >> 6); if (k == 0) za = v; else if (k == 1) zb = v; else zc = v; }
                    if (--rin_slot < 0) rin_slot = RC - 1;
                    --e_a; --e_b; --e_c; --pa; --pbp; --pc;
                }
                if (i >= 0 && (i >> 6) != B) { za = zb; zb = zc; zc = 0.0; }   // crossed into block B-1
            }
        } else {
            stage(m - PFB, tid - 64, 192);
        }
        __syncthreads();
    }
    if (tid == 0) { a.status[2] = (int)((t_factor - t_begin) >> 10); a.status[3] = (int)((__builtin_amdgcn_s_memtime() - t_factor) >> 10); }
}

// ---------------------------------------------------------------------------------------------------
// Blocked bordered-band LDL' (block = 16 columns) -- the factorisation half of the band solver for bw <= 80.
// Per block column J:
//   * wave 0 holds EVERY row of the block column (diagonal tile, the NBW sub-diagonal tiles, the border/rhs tile:
//     <= 128 rows, two per lane) in registers and runs the 16 pivots there: the pivot row is broadcast with
//     v_readlane (the diagonal tile is kept fully symmetric, so row k of it supplies all multipliers), no LDS
//     traffic and no barrier inside the block.  The panel W = L*D and 1/D go to LDS (two buffers, by block parity);
//   * the rank-16 trailing update runs tile by tile on the fp64 matrix cores, C(16x16) -= W_I * (W_K / D)', four
//     v_mfma_f64_16x16x4_f64 per tile, and is split by urgency (look-ahead): between the factorisations of J and J+1
//     wave 0 forms only W_1 and updates only the DIAGONAL tile of block column J+1; every other tile-update of block J
//     (the rest of column J+1, columns J+2..J+NBW, the border corner) is done by the helper waves WHILE wave 0
//     factors block J+1;
//   * waves 1-3 also stream the next tile column in from HBM and the previous block's factor out (band layout,
//     consumed by band_backward_tiles_kernel) behind wave 0's factorisation.
// Tiles live in an LDS ring indexed by (block column mod (NBW+2), tile row), rows padded to 17 doubles.
// ---------------------------------------------------------------------------------------------------
// rev / nJs / sep_out: twisted (two-sided) factorisation -- one workgroup takes the band from the top, a second one from
// the bottom (rev = 1: it sees the matrix with rows and columns reversed, still a band), each stops after nJs blocks,
// and what they have accumulated on the separator in between goes to sep_out (band_sep_solve_kernel).
struct BlkArgs { const double* Sb; double* Lb; double* corner_out; double* sep_out; int n_band, bw, nbd, H, NBW, rev, nJs, timing; int* status; };
struct BlkArgs2 { BlkArgs c[2]; };

NLLS_DEV double readlane_d(double x, int k) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), k), __builtin_amdgcn_readlane(__double2loint(x), k));
}

struct BlkLds { double* tiles; double* corner; double* Wp; double* dvec; double* Li; double* dummy; int TW, TR, NBW, H, bw, n_band, nJ, rev; };
constexpr int BLK_P = 17, BLK_TS = 16 * BLK_P;                // padded tile row, doubles per tile
NLLS_DEV double* blk_tile(const BlkLds& S, int K, int ti) { return S.tiles + ((size_t)(K % S.TW) * S.TR + ti) * BLK_TS; }          // (modulo: cold paths only)
NLLS_DEV double* blk_slot_tile(const BlkLds& S, int slot, int ti) { return S.tiles + ((size_t)slot * S.TR + ti) * BLK_TS; }   // slot = column % TW, kept incrementally
NLLS_DEV double* blk_panel(const BlkLds& S, int J) { return S.Wp + (size_t)(J & 1) * S.TR * 16 * BLK_P; }   // W of block J
NLLS_DEV double* blk_d(const BlkLds& S, int J) { return S.dvec + (J & 1) * 32; }                               // D[16], 1/D[16]
NLLS_DEV double* blk_li(const BlkLds& S, int J) { return S.Li + (J & 1) * 16 * BLK_P; }                         // inv(L_JJ)'

// wave 0: LDL' of the 16x16 diagonal tile of block column J, entirely on the matrix cores.  The tile sits in the
// accumulator layout of v_mfma_f64_16x16x4_f64 (register r of lane (li, lk) = A[lk + 4r][li]); it is kept fully
// symmetric, so row k -- ONE register, the 16 lanes with lk = k % 4 -- is the pivot column w.  With every other lane
// zeroed that register is directly a valid A operand (A[i][kk = k % 4] = w_i) and B operand (B[kk][j] = w_j / -d_k):
// one MFMA applies the whole rank-1 update, no cross-lane traffic except the two v_readlanes that fetch d_k.
// A second accumulator starts as the identity and takes the same column operations (transposed): it ends as inv(L),
// so the sub-diagonal tiles need no substitution, W_T = T * inv(L)' is a matrix-core product (blk_panel_tile).
// Serial chain per pivot: readlane d -> v_rcp_f64 + two Newton steps -> scale -> MFMA.
__device__ __forceinline__ void blk_factor(const BlkLds& S, int J, int jslot, int* status, const double4_t* Ain) {
    constexpr int P = BLK_P;
    const int lane = threadIdx.x & 63, li = lane & 15, lk = lane >> 4;
    double4_t A, Bt;                                          // Bt[n][j]: transpose of the identity rows' tile
    {
        const double* t0 = blk_slot_tile(S, jslot, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) { A[r] = Ain ? (*Ain)[r] : t0[(lk + 4 * r) * P + li]; Bt[r] = (lk + 4 * r == li) ? 1.0 : 0.0; }   // (wave 0 made this tile final itself, one phase ago: it is still in its registers)
    }
    double* db = blk_d(S, J);
    int badk = 16;                                            // first pivot of this block that is zero or NaN
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        constexpr int dummy = 0; (void)dummy;
        const int q = k & 3, r = k >> 2;
        const double w = A[r], bt = Bt[r];                    // row k of both tiles lives in the lanes with lk == q
        const double dk = readlane_d(w, 16 * q + k);
        double rdk = __builtin_amdgcn_rcp(dk);
        // operands that do not depend on 1/d: w masked to the rows below the pivot, and the lane masks
        const bool rowq = lk == q;
        const double am = (rowq && li > k) ? w : 0.0;
        const double bm = rowq ? bt : 0.0;
        rdk = fma(fma(-dk, rdk, 1.0), rdk, rdk); rdk = fma(fma(-dk, rdk, 1.0), rdk, rdk);   // Newton: full fp64 accuracy
        if (!(fabs(dk) > 0.0)) badk = badk < k ? badk : k;
        db[k] = dk; db[16 + k] = rdk;                         // every lane, same value
        if (k < 15) {
            A = __builtin_amdgcn_mfma_f64_16x16x4f64(am, am * -rdk, A, 0, 0, 0);      // A[i][j]  -= w_i w_j / d     (i, j > k)
            Bt = __builtin_amdgcn_mfma_f64_16x16x4f64(am, bm * -rdk, Bt, 0, 0, 0);    // Bt[j][i] -= w_j B[i][k] / d (j > k)
        }
    }
    double* Wb = blk_panel(S, J);
    double* Lij = blk_li(S, J);
#pragma unroll
    for (int r = 0; r < 4; ++r) { Wb[(lk + 4 * r) * P + li] = A[r]; Lij[li * P + (lk + 4 * r)] = Bt[r]; }   // Li[j][n] = Bt[n][j] = inv(L)'[j][n]
    if (badk < 16 && lane == 0) atomicCAS(status, 0, 1 + 16 * J + badk);
}
// panel tile ti (1..NBW sub-diagonal, NBW+1 border) of block J:  W = T * inv(L)'  -> panel rows 16*ti..16*ti+15
__device__ __forceinline__ void blk_panel_tile(const BlkLds& S, int J, int jslot, int ti) {
    constexpr int P = BLK_P;
    const int lane = threadIdx.x & 63, li = lane & 15, lk = lane >> 4;
    const double* T = blk_slot_tile(S, jslot, ti); double* Wt = blk_panel(S, J) + (size_t)ti * 16 * P;
    double av[4], bv[4];
    const double* Lij = blk_li(S, J);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) { av[kk] = T[li * P + 4 * kk + lk]; bv[kk] = Lij[(4 * kk + lk) * P + li]; }
    double4_t acc = {0, 0, 0, 0}, acc2 = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[0], bv[0], acc, 0, 0, 0);
    acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[1], bv[1], acc2, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[2], bv[2], acc, 0, 0, 0);
    acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[3], bv[3], acc2, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 4; ++r) Wt[(lk + 4 * r) * P + li] = acc[r] + acc2[r];
}
// wave 0 between two factorisations: panel tile W_1 = T_1 inv(L)' and, straight from the registers, the update of the
// diagonal tile of block column J+1, C -= W_1 (W_1 / D)'.  The panel product is formed TRANSPOSED (operands swapped:
// inv(L) T_1'), because the accumulator layout of W_1' -- register r of lane (li, lk) = W_1[li][lk + 4 r] -- is exactly
// the operand layout the update needs (A[i][k] = W_1[i][k], B[k][j] = W_1[j][k] / d_k): no LDS round trip between the
// two.  W_1 still goes to the panel in LDS (normal layout) for the helpers' tile-updates.
__device__ __forceinline__ void blk_panel_update_diag(const BlkLds& S, int J, int jslot, double4_t& diag_out) {
    constexpr int P = BLK_P;
    const int lane = threadIdx.x & 63, li = lane & 15, lk = lane >> 4;
    const double* T = blk_slot_tile(S, jslot, 1); double* Wt = blk_panel(S, J) + (size_t)16 * P;
    const double* Lij = blk_li(S, J); const double* rd = blk_d(S, J) + 16;
    double av[4], bv[4], rdk[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) { av[kk] = T[li * P + 4 * kk + lk]; bv[kk] = Lij[(4 * kk + lk) * P + li]; rdk[kk] = rd[4 * kk + lk]; }
    int sl = jslot + 1; sl -= sl >= S.TW ? S.TW : 0;
    double* Ct = ((J + 1 < S.nJ) ? S.tiles + (size_t)sl * S.TR * BLK_TS : S.dummy) + lk * P + li;
    double4_t c, c2 = {0, 0, 0, 0};
#pragma unroll
    for (int r = 0; r < 4; ++r) c[r] = Ct[4 * r * P];
    double4_t a1 = {0, 0, 0, 0}, a2 = {0, 0, 0, 0};
    a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(bv[0], av[0], a1, 0, 0, 0);
    a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(bv[1], av[1], a2, 0, 0, 0);
    a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(bv[2], av[2], a1, 0, 0, 0);
    a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(bv[3], av[3], a2, 0, 0, 0);
    double w[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) w[r] = a1[r] + a2[r];            // W_1[li][lk + 4 r]
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(-w[0], w[0] * rdk[0], c, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(-w[1], w[1] * rdk[1], c2, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(-w[2], w[2] * rdk[2], c, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(-w[3], w[3] * rdk[3], c2, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 4; ++r) Wt[li * P + lk + 4 * r] = w[r];
#pragma unroll
    for (int r = 0; r < 4; ++r) { diag_out[r] = c[r] + c2[r]; Ct[4 * r * P] = diag_out[r]; }
}
// tile column K <- band layout in HBM (identity behind the last column); threads t0, t0+nt, ...  Gather form: every
// word of the TR tiles is computed from its (row, column), so the ring slot needs no zero fill and one pass suffices.
// Two halves: blk_land_load issues the HBM loads into registers, blk_land_store puts them into the ring slot; whatever
// runs between the two hides the HBM latency.
constexpr int BLK_T = 512, BLK_HELP = BLK_T - 64;             // threads of the factor kernel; wave 0 factors, the rest help
constexpr int BLK_LANDW = 5;                                  // words per helper thread: ceil(8 tiles * 256 / 448)
__device__ __forceinline__ void blk_land_load(const BlkLds& S, const double* __restrict__ Sb, int K, int t0, int nt, double (&val)[BLK_LANDW]) {
    const int H = S.H, nwords = S.TR * 16 * 16;
#pragma unroll
    for (int q = 0; q < BLK_LANDW; ++q) {
        const int w = t0 + q * nt; val[q] = 0.0;
        if (w < nwords) {
            const int ti = w >> 8, cc = (w >> 4) & 15, r = w & 15;     // r fastest: 16 lanes read 16 consecutive band entries
            int c = 16 * K + cc, e;
            if (ti <= S.NBW) { e = 16 * ti + r - cc; if (e < 0) { c = 16 * K + r; e = -e; } }     // upper part of the diagonal tile: mirror
            else e = S.bw + 1 + r;                                                              // border tile: row r = border index
            if (e < H && (ti > S.NBW || e <= S.bw)) {
                if (!S.rev) val[q] = (c < S.n_band) ? Sb[(size_t)c * H + e] : (e == 0 ? 1.0 : 0.0);
                else if (ti > S.NBW) val[q] = (c < S.n_band) ? Sb[(size_t)(S.n_band - 1 - c) * H + e] : 0.0;          // border / rhs rows of reversed column c
                else val[q] = (c + e < S.n_band) ? Sb[(size_t)(S.n_band - 1 - c - e) * H + e] : ((e == 0 && c >= S.n_band) ? 1.0 : 0.0);   // R(c + e, c) = S(n-1-c, n-1-c-e)
            }
        }
    }
}
__device__ __forceinline__ void blk_land_store(const BlkLds& S, int slot, int t0, int nt, const double (&val)[BLK_LANDW]) {
    constexpr int P = BLK_P;
    double* base = blk_slot_tile(S, slot, 0);
    const int nwords = S.TR * 16 * 16;
#pragma unroll
    for (int q = 0; q < BLK_LANDW; ++q) {
        const int w = t0 + q * nt;
        if (w < nwords) { const int ti = w >> 8, cc = (w >> 4) & 15, r = w & 15; base[(size_t)ti * BLK_TS + r * P + cc] = val[q]; }
    }
}
// The same landing for an INTERIOR tile column (every entry it reads exists: 16 K + 15 + bw < n_band), from a per-thread
// plan made once: a word's LDS offset, its offset in the band array for column 0 and the (signed) stride per column do
// not depend on K.  This is what runs at (almost) every block step; the general form above handles the ends.
struct BlkLandPlan { int loff[BLK_LANDW]; long long goff[BLK_LANDW]; long long gstep; unsigned on; };
__device__ __forceinline__ void blk_land_plan(const BlkLds& S, int t0, int nt, BlkLandPlan& Pl) {
    constexpr int P = BLK_P;
    const int H = S.H, nwords = S.TR * 16 * 16;
    Pl.on = 0; Pl.gstep = S.rev ? -16LL * H : 16LL * H;
#pragma unroll
    for (int q = 0; q < BLK_LANDW; ++q) {
        const int w = t0 + q * nt; Pl.loff[q] = 0; Pl.goff[q] = 0;
        if (w >= nwords) continue;
        const int ti = w >> 8, cc = (w >> 4) & 15, r = w & 15;
        int c = cc, e;
        if (ti <= S.NBW) { e = 16 * ti + r - cc; if (e < 0) { c = r; e = -e; } } else e = S.bw + 1 + r;
        Pl.loff[q] = ti * BLK_TS + r * P + cc;
        if (e < H && (ti > S.NBW || e <= S.bw)) {
            Pl.on |= 1u << q;
            Pl.goff[q] = !S.rev ? (long long)c * H + e : (ti > S.NBW ? (long long)(S.n_band - 1 - c) * H + e : (long long)(S.n_band - 1 - c - e) * H + e);
        }
    }
}
__device__ __forceinline__ void blk_land_load_fast(const double* __restrict__ Sb, int K, const BlkLandPlan& Pl, double (&val)[BLK_LANDW]) {
    const double* col = Sb + K * Pl.gstep;
#pragma unroll
    for (int q = 0; q < BLK_LANDW; ++q) val[q] = (Pl.on >> q & 1) ? col[Pl.goff[q]] : 0.0;
}
__device__ __forceinline__ void blk_land_store_fast(const BlkLds& S, int slot, int t0, int nt, const BlkLandPlan& Pl, const double (&val)[BLK_LANDW]) {
    double* base = blk_slot_tile(S, slot, 0);
    const int nwords = S.TR * 16 * 16;
#pragma unroll
    for (int q = 0; q < BLK_LANDW; ++q) if (t0 + q * nt < nwords) base[Pl.loff[q]] = val[q];
}
// Tile-updates of one block step, numbered u = 0..nup-1: first the NBW+1 tiles of block column J+1 (K = 1: tile rows
// 1..NBW and the border row), then K = 2..NBW (tile rows K..NBW and the border row), last the border corner (K = 0).
// A wave's share of a range of them is the same at every block step, so it is unpacked once.
template <int MAXU> struct BlkUpd { int K[MAXU], woff[MAXU], loff[MAXU], toff[MAXU]; };   // K < 0: none; offsets in doubles
template <int MAXU>
__device__ __forceinline__ void blk_update_list(const BlkLds& S, int first, int count, int w0, int nw, BlkUpd<MAXU>& U) {
    constexpr int P = BLK_P;
    const int NBW = S.NBW, nup = NBW * (NBW + 1) / 2 + NBW + 1;
#pragma unroll
    for (int q = 0; q < MAXU; ++q) {
        const int u = first + w0 + q * nw; U.K[q] = -1; U.woff[q] = 0; U.loff[q] = 0; U.toff[q] = 0;
        if (w0 < 0 || w0 + q * nw >= count) continue;
        int K = 0, pi = NBW + 1, trow = 0;                            // K = 0 marks the border corner
        if (u < nup - 1) { int uu = u; K = 1; while (uu >= NBW - K + 2) { uu -= NBW - K + 2; ++K; }
            pi = (uu == NBW - K + 1) ? NBW + 1 : K + uu; trow = (uu == NBW - K + 1) ? NBW + 1 : uu; }
        else if (u != nup - 1) continue;
        U.K[q] = K; U.woff[q] = pi * 16 * P; U.loff[q] = (K > 0 ? K : NBW + 1) * 16 * P; U.toff[q] = trow * BLK_TS;
    }
}
// apply this wave's tile-updates of block J (panel W and 1/D of block J in LDS; jslot = J % TW) on the matrix cores.
// A tile-update that does not exist at this step works on a spare tile: no predicated stores, no branches.
template <int MAXU>
__device__ __forceinline__ void blk_update(const BlkLds& S, int J, int jslot, const BlkUpd<MAXU>& U) {
    constexpr int P = BLK_P;
    const int lane = threadIdx.x & 63, li = lane & 15, lk = lane >> 4;
    const double* Wb = blk_panel(S, J); const double* rd = blk_d(S, J) + 16;
    double* Ct[MAXU]; double4_t acc[MAXU], acc2[MAXU]; double wv[MAXU][4], lv[MAXU][4];
    double rdk[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) rdk[kk] = rd[4 * kk + lk];
    const int lo = li * P + lk, co = lk * P + li;
    // all operand loads first, then the MFMAs, then the stores: the LDS latency of one tile hides behind the others
#pragma unroll
    for (int q = 0; q < MAXU; ++q) {
        const int K = U.K[q]; const bool ok = K >= 0 && (K == 0 || J + K < S.nJ);
        int sl = jslot + (K > 0 ? K : 0); sl -= sl >= S.TW ? S.TW : 0;
        double* ct = (K > 0) ? S.tiles + (size_t)sl * S.TR * BLK_TS + U.toff[q] : S.corner;
        Ct[q] = (ok ? ct : S.dummy) + co;
        const double* Wt = Wb + U.woff[q] + lo; const double* Lt = Wb + U.loff[q] + lo;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) { wv[q][kk] = -Wt[4 * kk]; lv[q][kk] = Lt[4 * kk] * rdk[kk]; }
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[q][r] = Ct[q][4 * r * P];
        acc2[q] = double4_t{0, 0, 0, 0};
    }
#pragma unroll
    for (int q = 0; q < MAXU; ++q) {
        acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(wv[q][0], lv[q][0], acc[q], 0, 0, 0);
        acc2[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(wv[q][1], lv[q][1], acc2[q], 0, 0, 0);
    }
#pragma unroll
    for (int q = 0; q < MAXU; ++q) {
        acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(wv[q][2], lv[q][2], acc[q], 0, 0, 0);
        acc2[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(wv[q][3], lv[q][3], acc2[q], 0, 0, 0);
    }
#pragma unroll
    for (int q = 0; q < MAXU; ++q) {
#pragma unroll
        for (int r = 0; r < 4; ++r) Ct[q][4 * r * P] = acc[q][r] + acc2[q][r];
    }
}
// factor block column J -> HBM (consumed by band_backward_tiles_kernel).  The backward pass needs
//   x_J = inv(L_JJ)' ( z_J - Lbd_J' xb - sum_K L_{J+K,J}' x_{J+K} ),
// so the tiles are exported PRE-MULTIPLIED by inv(L_JJ):  M_K = L_{J+K,J} inv(L_JJ)  (then x_J = zh_J - sum_K M_K' x_{J+K},
// one matrix-vector stage per block instead of two dependent ones).  The products run on the matrix cores of the helper
// waves, off wave 0's critical path.  Per block, blk_fsize() doubles, tiles row-major and unpadded:
//   [256 (K-1), 256 K)      M_K, K = 1..NBW   [i'][c]
//   then 16                 zh = inv(L_JJ)' z,  z = (L^-1 b) / D the rhs row
//   then nbd x 16           border rows, Mbd = Lbd inv(L_JJ)
NLLS_HD int blk_fsize(int NBW, int nbd) { return NBW * 256 + (nbd + 1) * 16; }
// one wavefront: tile ti (1..NBW, or NBW+1 = the border / rhs tile) of block J
__device__ __forceinline__ void blk_export_tile(const BlkLds& S, double* __restrict__ Lt, int nbd, int J, int ti) {
    constexpr int P = BLK_P;
    const int lane = threadIdx.x & 63, li = lane & 15, lk = lane >> 4;
    const double* Wt = blk_panel(S, J) + (size_t)ti * 16 * P; const double* rd = blk_d(S, J) + 16; const double* Lij = blk_li(S, J);
    double4_t acc = {0, 0, 0, 0}, acc2 = {0, 0, 0, 0};
    double av[4], bv[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) { av[m] = Wt[li * P + 4 * m + lk] * rd[4 * m + lk]; bv[m] = Lij[li * P + 4 * m + lk]; }   // A[i][n] = L[i][n]; B[n][c] = inv(L)'[c][n]
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[0], bv[0], acc, 0, 0, 0);
    acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[1], bv[1], acc2, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[2], bv[2], acc, 0, 0, 0);
    acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[3], bv[3], acc2, 0, 0, 0);
    double* dst = Lt + (size_t)J * blk_fsize(S.NBW, nbd);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int i = lk + 4 * r; const double v = acc[r] + acc2[r];
        if (ti <= S.NBW) dst[(ti - 1) * 256 + i * 16 + li] = v;
        else if (i == nbd) dst[S.NBW * 256 + li] = v;                      // rhs row -> zh
        else if (i < nbd) dst[S.NBW * 256 + 16 + i * 16 + li] = v;         // border rows
    }
}
NLLS_DEV int blk_export_wave(int wave) { return wave < 4 ? wave : wave - 1; }   // waves 1,2,3,5,6,7 -> tiles 1..6 (wave 4 shares wave 0's SIMD)

__global__ __launch_bounds__(BLK_T) void band_blocked_factor_kernel(BlkArgs2 args) {
    const BlkArgs a = args.c[blockIdx.x];
    extern __shared__ __attribute__((aligned(16))) double sm[];
    constexpr int P = BLK_P, TS = BLK_TS;
    const int tid = threadIdx.x, wave = tid >> 6;
    const int n_band = a.n_band, nbd = a.nbd, H = a.H, NBW = a.NBW, nbr = nbd + 1;
    BlkLds S; S.TW = NBW + 2; S.TR = NBW + 2; S.NBW = NBW; S.H = H; S.bw = a.bw; S.n_band = n_band; S.nJ = (n_band + 15) >> 4; S.rev = a.rev;
    S.tiles = sm;                                             // [TW][TR][TS]
    S.corner = S.tiles + (size_t)S.TW * S.TR * TS;            // [TS] border x border (row/col = border index, rhs = nbd)
    S.Wp = S.corner + TS;                                     // [2][TR*16][P]
    S.dvec = S.Wp + 2 * (size_t)S.TR * 16 * P;                // [2][32]
    S.Li = S.dvec + 64;                                       // [2][16][P]: inv(L_JJ)' by block parity
    S.dummy = S.Li + 2 * 16 * P;                              // [TS] spare tile: target of the tile-updates that do not exist at a step
    const int nJ = S.nJ, nJs = a.nJs;                         // blocks of the matrix; blocks this workgroup factors
    for (int i = tid; i < S.TW * S.TR * TS + TS; i += BLK_T) S.tiles[i] = 0.0;
    __syncthreads();
    for (int e = tid; e < nbr * nbr; e += BLK_T) { const int i = e % nbr, j = e / nbr; if (i >= j) { const double v = a.Sb[(size_t)n_band * H + e]; S.corner[i * P + j] = v; S.corner[j * P + i] = v; } }
    {   // the first NBW + 1 tile columns: all HBM loads in flight together, then the LDS stores
        double pv[6][BLK_LANDW];
#pragma unroll
        for (int K = 0; K < 6; ++K) if (K <= NBW && K < nJ) blk_land_load(S, a.Sb, K, tid, BLK_T, pv[K]);
#pragma unroll
        for (int K = 0; K < 6; ++K) if (K <= NBW && K < nJ) blk_land_store(S, K, tid, BLK_T, pv[K]);   // slot = column (K < TW)
    }
    // Tile-updates of a block step (blk_update_list numbering): u = 0 is the DIAGONAL tile of block column J+1 -- the only
    // one the next factorisation waits for: wave 0 applies it itself, right after the panel tile it needs (W_1), and goes
    // on to factor J+1.  All the others -- the rest of column J+1 (needed by the panel step of J+1, one factorisation
    // later), columns J+2..J+NBW and the border corner -- are done by the helper waves while wave 0 factors J+1.
    const int nup = NBW * (NBW + 1) / 2 + NBW + 1;
    constexpr int NW = BLK_T / 64;
    BlkUpd<3> Uh; blk_update_list<3>(S, 1, nup - 1, wave - 1, NW - 1, Uh);
    const unsigned long long t_begin = __builtin_amdgcn_s_memtime();   // diagnostics only (nlls_get_solve_stats)
    int jslot = 0;                                            // J % TW, kept incrementally (no integer division in the loop)
    double4_t diag = {0, 0, 0, 0};                            // wave 0: the next diagonal tile, from its update to its factorisation
    BlkLandPlan plan; blk_land_plan(S, tid - 64, BLK_HELP, plan);
    __syncthreads();                                          // the first tile columns have landed
    for (int J = 0; J < nJs; ++J) {
        const int pslot = jslot == 0 ? S.TW - 1 : jslot - 1;  // slot of column J-1 = slot of column J+NBW+1
        if (wave == 0) blk_factor(S, J, jslot, a.status, J > 0 ? &diag : nullptr);
        else {
            // helpers, behind wave 0's factorisation: block J-1's remaining tile-updates, tile column J+NBW+1 into the
            // ring slot of column J-1 (HBM latency), block J-1's factor out
            double lv[BLK_LANDW]; const int Kl = J + NBW + 1; const bool landing = Kl < nJ, interior = 16 * Kl + 15 + S.bw < n_band;
            if (landing) { if (interior) blk_land_load_fast(a.Sb, Kl, plan, lv); else blk_land_load(S, a.Sb, Kl, tid - 64, BLK_HELP, lv); }
            if (J > 0) { blk_update<3>(S, J - 1, pslot, Uh); if (wave != 4 && blk_export_wave(wave) <= NBW + 1) blk_export_tile(S, a.Lb, nbd, J - 1, blk_export_wave(wave)); }
            if (landing) blk_land_store_fast(S, pslot, tid - 64, BLK_HELP, plan, lv);
        }
        __syncthreads();                                      // (B) diagonal tile factored; block J-1's updates all applied: column J is final
        if (wave == 0) blk_panel_update_diag(S, J, jslot, diag);                                     // W_1 and, from the registers, the diagonal tile of column J+1
        else if (wave <= NBW) blk_panel_tile(S, J, jslot, wave + 1);                          // W_2 .. W_{NBW+1}
        __syncthreads();                                      // (C) panel J in LDS; the next diagonal tile is ready
        if (++jslot == S.TW) jslot = 0;
    }
    if (wave > 0) blk_update<3>(S, nJs - 1, (nJs - 1) % S.TW, Uh);   // the last block's remaining updates (border corner, separator)
    if (wave > 0 && wave != 4 && blk_export_wave(wave) <= NBW + 1) blk_export_tile(S, a.Lb, nbd, nJs - 1, blk_export_wave(wave));
    __syncthreads();
    if (a.sep_out) {
        // the NBW tile columns behind the last factored block, with everything this side has subtracted from them:
        // dense [16 NBW][16 NBW] (lower block triangle), then the rhs row
        const int SW = 16 * NBW;
        for (int idx = tid; idx < NBW * NBW * 256; idx += BLK_T) {
            const int t2 = idx >> 8, r = (idx >> 4) & 15, cc = idx & 15, Kc = t2 / NBW, ti = t2 % NBW;
            if (Kc + ti < NBW) a.sep_out[(size_t)(16 * (Kc + ti) + r) * SW + 16 * Kc + cc] = blk_tile(S, nJs + Kc, ti)[r * P + cc];
        }
        for (int idx = tid; idx < SW; idx += BLK_T) a.sep_out[(size_t)SW * SW + idx] = blk_tile(S, nJs + (idx >> 4), NBW + 1)[nbd * P + (idx & 15)];
    }
    for (int e = tid; e < nbr * nbr; e += BLK_T) { const int i = e % nbr, j = e / nbr; a.corner_out[e] = S.corner[i * P + j]; }
    if (tid == 0 && blockIdx.x == 0 && a.timing) a.status[2] = (int)((__builtin_amdgcn_s_memtime() - t_begin) >> 10);
}

// Separator of the twisted factorisation: the columns [cA, cA + ws) between the two sides.  Its matrix is what the top
// side left (sepA), plus what the bottom side left (sepB, in reversed indices), minus the original entries, which both
// sides had loaded; same for the rhs row.  It is written out as a (dense) band system of its own, Hs = ws + 1 entries per
// column, and goes through the same blocked factor / backward kernels (five blocks).
__global__ __launch_bounds__(256) void band_sep_combine_kernel(const double* __restrict__ Sb, const double* __restrict__ sepA, const double* __restrict__ sepB,
                                                               int cA, int ws, int SW, int bw, int nbd, int H, double* __restrict__ Ssep) {
    const int Hs = ws + 1;                                    // entries 0..ws-1: the column from its diagonal down; entry ws: the rhs row
    for (int idx = blockIdx.x * 256 + threadIdx.x; idx < ws * Hs; idx += gridDim.x * 256) {
        const int j = idx / Hs, e = idx % Hs, i = j + e;
        double v = 0.0;
        if (e == ws) v = sepA[(size_t)SW * SW + j] + sepB[(size_t)SW * SW + (ws - 1 - j)] - Sb[(size_t)(cA + j) * H + bw + 1 + nbd];
        else if (i < ws) v = sepA[(size_t)i * SW + j] + sepB[(size_t)(ws - 1 - j) * SW + (ws - 1 - i)] - ((e <= bw) ? Sb[(size_t)(cA + j) * H + e] : 0.0);
        Ssep[idx] = v;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) Ssep[(size_t)ws * Hs] = 0.0;   // the 1 x 1 "corner" (rhs x rhs), unused
}

// Border corner + backward pass of the blocked band solver (pre-multiplied tiles from band_blocked_factor_kernel):
//   x_J = zh_J - Mbd_J' xb - sum_{K=1..NBW} M_K' x_{J+K},   J = nJs-1 .. 0
// Wave 0, lane (c, g) = (lane % 16, lane / 16): it forms the part of entry c that comes from rows g, g+4, g+8, g+12 of
// every tile (4 NBW multiply-adds) and parks it in LDS; the four parts of an entry are summed by whoever needs the
// entry next -- lane (c, g) of the next block needs x[g + 4 q], q = 0..3, i.e. 16 parts -- so one LDS round trip per
// block is the whole serial chain, and the far tiles' products (their x is older) are computed while it is in flight.
// Waves 1-3 copy the factor HBM -> LDS with global_load_lds_dwordx4 (no registers, BWD_AHEAD blocks ahead, block B by
// wave 1 + B % 3); a wave retires a block with a counted s_waitcnt just before the barrier that hands it to wave 0.
// rev / nJs / xnext: the two sides of the twisted factorisation (a side's unknowns behind its last block are the
// separator's, xnext = their index in xr seen from this side; -1: nothing behind the last block).
struct BwdArgs { double* Lt; const double* corner_in; double* xr; int n_band, nbd, NBW, rev, nJs, xnext, nxnext, timing; int* status; };
struct BwdArgs2 { BwdArgs c[2]; };
constexpr int BWD_AHEAD = 9, BWD_RING = BWD_AHEAD + 1;         // ring slots = blocks in LDS
NLLS_HD int bwd_slot(int NBW) { return NBW * 256 + 128; }      // doubles copied per block: the tiles, then zh (+ whatever follows)
template <int NBW>
__global__ __launch_bounds__(256) void band_backward_tiles_kernel(BwdArgs2 args) {
    const BwdArgs a = args.c[blockIdx.x];
    extern __shared__ __attribute__((aligned(16))) double sm[];
    constexpr int SLOT = NBW * 256 + 128, NI = SLOT / 128;     // NI wave-wide 16-byte copies per block
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, c = lane & 15, g = lane >> 4;
    const int n_band = a.n_band, nbd = a.nbd, nbr = nbd + 1;
    const int nJ = a.nJs, fs = blk_fsize(NBW, nbd);           // blocks of this side
    double* ring = sm;                            // [BWD_RING + 1][SLOT]; the extra slot takes the copies of blocks that do not exist
    double* red = ring + (size_t)(BWD_RING + 1) * SLOT;   // [2][16][4]: the four parts of the 16 entries of a block, by block parity
    double* xs = red + 128;                       // [8][16]: x of the last blocks (ring by block index), read by the helper waves
    double* farp = xs + 128;                      // [2][64]: the far tiles' (K >= 3) share of a block's parts, by block parity
    double* Cl = farp + 128;                      // nbr x nbr border corner (col-major, lower), last row = rhs
    double* xb = Cl + nbr * nbr;                  // nbr
    for (int e = tid; e < nbr * nbr; e += 256) Cl[e] = a.corner_in[e];
    __syncthreads();
    const unsigned long long t_begin = __builtin_amdgcn_s_memtime();
    if (tid == 0) {
        for (int j = 0; j < nbd; ++j) {
            double d = Cl[j + nbr * j];
            if (d == 0.0 || d != d) { atomicCAS(a.status, 0, 1 + n_band + j); d = 1.0; }
            for (int c2 = j + 1; c2 < nbd; ++c2) { const double f = Cl[c2 + nbr * j] / d; for (int i = c2; i < nbr; ++i) Cl[i + nbr * c2] -= Cl[i + nbr * j] * f; }
            for (int i = j + 1; i < nbr; ++i) Cl[i + nbr * j] /= d;
            Cl[j + nbr * j] = d;
        }
        for (int r = nbd - 1; r >= 0; --r) { double v = Cl[nbd + nbr * r]; for (int r2 = r + 1; r2 < nbd; ++r2) v -= Cl[r2 + nbr * r] * xb[r2]; xb[r] = v; a.xr[n_band + r] = v; }
    }
    __syncthreads();
    if (nbd > 0) {                                // fold the border unknowns into zh, in place:  zh_J -= Mbd_J' xb
        for (int idx = tid; idx < nJ * 16; idx += 256) {
            double* p = a.Lt + (size_t)(idx >> 4) * fs + NBW * 256; const int c2 = idx & 15;
            double v = p[c2]; for (int q = 0; q < nbd; ++q) v -= p[16 + q * 16 + c2] * xb[q]; p[c2] = v;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    // block B -> ring slot B % BWD_RING (blocks that do not exist: block 0 -> the spare slot, so that every issue is NI copies)
    auto issue = [&](int B) {
        const double* src = a.Lt + (size_t)(B >= 0 ? B : 0) * fs + 2 * lane;
        double* dst = ring + (size_t)(B >= 0 ? B % BWD_RING : BWD_RING) * SLOT;
#pragma unroll
        for (int i = 0; i < NI; ++i) __builtin_amdgcn_global_load_lds(src + 128 * i, (__attribute__((address_space(3))) void*)(dst + 128 * i), 16, 0, 0);
    };
    // x behind the last block (the separator's unknowns under the twisted factorisation, else zeros) into the xs ring
    for (int i = tid; i < 16 * NBW; i += 256) { const int blk = nJ + (i >> 4);
        xs[(blk & 7) * 16 + (i & 15)] = (a.xnext >= 0 && i < a.nxnext) ? a.xr[a.rev ? a.xnext - i : a.xnext + i] : 0.0; }
    __syncthreads();
    // The far tiles (K >= 3) of block B only need x of blocks B+3.., known two iterations before wave 0 gets to block B:
    // the wave that copied block B in forms their share of the parts right after the copy has landed (x from the xs ring)
    // and leaves it in farp -- wave 0's chain per block is then two tiles, not NBW.
    auto far3 = [&](int B) {
        if constexpr (NBW >= 3) {
            const double* Bt = ring + (size_t)(B % BWD_RING) * SLOT;
            double f = 0.0;
#pragma unroll
            for (int K = NBW; K >= 3; --K)
#pragma unroll
                for (int q = 0; q < 4; ++q) f = fma(Bt[(K - 1) * 256 + (g + 4 * q) * 16 + c], xs[((B + K) & 7) * 16 + g + 4 * q], f);
            farp[(B & 1) * 64 + lane] = f;
        }
    };
    if (wave > 0) {
        for (int B = nJ - 1; B > nJ - 1 - BWD_AHEAD; --B) if (1 + ((B % 3) + 3) % 3 == wave) issue(B);
        if (1 + (nJ - 1) % 3 == wave) { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * NI) : "memory"); far3(nJ - 1); }   // block nJ-1 has landed
    }
    // at the top of iteration J: xq[0][q] = x_{J+2}[g + 4 q].  Behind the last block: the separator's unknowns (twisted
    // factorisation), else nothing; xfirst = the block right behind the end (what the first iteration gets for x_{J+1})
    auto behind = [&](int i) { return (a.xnext >= 0 && i < a.nxnext) ? a.xr[a.rev ? a.xnext - i : a.xnext + i] : 0.0; };
    double xq[1][4], xfirst[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) { xfirst[q] = behind(g + 4 * q); xq[0][q] = behind(16 + g + 4 * q); }
    double zq[4] = {0, 0, 0, 0};                  // zh of the block whose parts are in flight, entries g + 4 q
    auto finish = [&](int Jp, double (&xnew)[4]) {           // x of block Jp from its four parts in LDS
        const double* rp = red + (Jp & 1) * 64;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const double4_t p4 = *reinterpret_cast<const double4_t*>(rp + 4 * (g + 4 * q));
            xnew[q] = zq[q] - ((p4[0] + p4[1]) + (p4[2] + p4[3]));
        }
    };
    // x leaves for HBM from the xs ring, by a helper wave and two iterations late: a global store in wave 0's loop would put
    // a vmcnt(0) -- the store's full HBM latency -- in front of its next LDS read (the compiler orders LDS reads behind
    // every outstanding vector-memory operation in a kernel that uses global_load_lds)
    auto store_x = [&](int Jp, const double* xv16) {          // lanes 0..15
        const int row = 16 * Jp + lane; if (lane < 16 && row < n_band) a.xr[a.rev ? n_band - 1 - row : row] = xv16[lane];
    };
    for (int J = nJ - 1; J >= 0; --J) {
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // block J is in LDS; the slot of block J+1 is free
        if (wave == 0) {
            const double* B0 = ring + (size_t)(J % BWD_RING) * SLOT;
            // every LDS read of the iteration is issued up front (nothing below reads LDS behind a write): the parts of
            // block J+1, the two near tiles, zh and the helpers' share for the far tiles
            const double* rp = red + ((J + 1) & 1) * 64;
            double4_t p4[4]; double m1[4], m2[4], zn[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) p4[q] = *reinterpret_cast<const double4_t*>(rp + 4 * (g + 4 * q));
#pragma unroll
            for (int q = 0; q < 4; ++q) m1[q] = B0[(g + 4 * q) * 16 + c];
            double far = NBW >= 3 ? farp[(J & 1) * 64 + lane] : 0.0;
#pragma unroll
            for (int q = 0; q < 4; ++q) { m2[q] = NBW >= 2 ? B0[256 + (g + 4 * q) * 16 + c] : 0.0; zn[q] = B0[NBW * 256 + g + 4 * q]; }
            double xnew[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) xnew[q] = J < nJ - 1 ? zq[q] - ((p4[q][0] + p4[q][1]) + (p4[q][2] + p4[q][3])) : xfirst[q];   // x_{J+1} (behind the end at first)
            // tile K = 2 (its x is complete) is summed beside the LDS round trip; behind x_{J+1} the chain is two fused
            // multiply-adds deep (two accumulators) and one addition
#pragma unroll
            for (int q = 0; q < 4; ++q) far = fma(m2[q], xq[0][q], far);
            const double pa = fma(m1[1], xnew[1], fma(m1[0], xnew[0], far)), pb = fma(m1[3], xnew[3], m1[2] * xnew[2]);
            red[(J & 1) * 64 + 4 * c + g] = pa + pb;
            if (c == 0) {
#pragma unroll
                for (int q = 0; q < 4; ++q) xs[((J + 1) & 7) * 16 + g + 4 * q] = xnew[q];
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) { xq[0][q] = xnew[q]; zq[q] = zn[q]; }
        } else {
            if (wave == 1 + J % 3 && J + 2 <= nJ - 1) store_x(J + 2, xs + ((J + 2) & 7) * 16);   // published during iteration J+1
            const int B = J - BWD_AHEAD;          // goes into the slot block J+1 has just left
            if (1 + ((B % 3) + 3) % 3 == wave) issue(B);
            if (J >= 1 && 1 + (J - 1) % 3 == wave) { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * NI) : "memory"); far3(J - 1); }   // block J-1 has landed
        }
    }
    if (wave == 0 && nJ > 0) {                    // the last two blocks: x_1 is in the ring (iteration 0), x_0 comes out now
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); double x0[4]; finish(0, x0);
        if (c == 0) {
#pragma unroll
            for (int q = 0; q < 4; ++q) { const int row = g + 4 * q; if (row < n_band) a.xr[a.rev ? n_band - 1 - row : row] = x0[q]; }
        }
        if (nJ > 1) store_x(1, xs + 16);
    }
    if (tid == 0 && blockIdx.x == 0 && a.timing) a.status[3] = (int)((__builtin_amdgcn_s_memtime() - t_begin) >> 10);
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
__global__ __launch_bounds__(256) void trial_finish_kernel(const double* __restrict__ cpart, int64_t ncp, const double* __restrict__ partials, int np,
                                                           const double* __restrict__ part2, int np2, double lambda, double* __restrict__ out, const int* __restrict__ status,
                                                           double* __restrict__ host_out, double seq) {
    __shared__ double red[6][4];
    if (blockIdx.x == 0) reduce_partials_body(cpart, ncp, out, &red[0][0]);
    else post_solve_finish_body(partials, np, part2, np2, lambda, out, status, red);
    // the scalars also go straight to the pinned host mirror (device-visible, coherent host memory): no copy command behind this launch.
    // Each workgroup then publishes the trial's sequence number: the host spins on the two numbers instead of sleeping in a stream
    // synchronisation (its wake-up costs more than the kernels of this size it waits for).
    if (host_out && threadIdx.x == 0) {
        if (blockIdx.x == 0) host_out[0] = out[0];
        else { host_out[1] = out[1]; host_out[2] = out[2]; host_out[4] = out[4]; host_out[5] = out[5]; host_out[8] = out[8]; host_out[9] = out[9]; host_out[10] = out[10]; }
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
    for (int e = lane; e < ne; e += 64) {
        const int b2 = e / rows, a2 = e - b2 * rows;
        const bool diag = J.kind == 0 && J.r0 == J.c0;
        if (diag && a2 < b2) continue;                        // diagonal block: the lower triangle (mirrored by the store)
        double v = 0.0;
        if (J.kind == 0) { if (J.copy_off >= 0) v = J.copy_trans ? a.A[J.copy_off + b2 + (int)J.cols * a2] : a.A[J.copy_off + a2 + rows * b2]; if (diag && a2 == b2) v += a.lambda; }
        else v = a.b[J.boff + a2];
        // shares, four at a time: the descriptors (uniform loads) and then the four values are in flight together
        for (uint32_t c0 = J.cbeg; c0 < J.cend; c0 += 4) {
            GatherCon k[4]; double t[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) k[u] = a.cons[c0 + u < J.cend ? c0 + u : J.cend - 1];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (k[u].ld) t[u] = k[u].aux ? a.slab[k[u].off + b2 + k[u].ld * a2] : a.slab[k[u].off + a2 + k[u].ld * b2];   // aux: the share lies above the diagonal of S in reduced order -- its transpose is wanted
                else {                                          // a member of a small supernode: e_a' (C_v + lambda I)^-1 e_b on the fly
                    const int dv = a.dv; const double* ea = a.A + k[u].off + (size_t)dv * a2; const double* eb = J.kind == 0 ? a.A + k[u].aux + (size_t)dv * b2 : a.b + k[u].aux;
                    const double* ci = a.Cinv + k[u].cinv; double s2 = 0.0;
                    for (int m = 0; m < dv; ++m) { double r2 = 0.0; for (int n2 = 0; n2 < dv; ++n2) r2 = fma(ci[m + dv * n2], eb[n2], r2); s2 = fma(ea[m], r2, s2); }
                    t[u] = s2;
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) if (c0 + u < J.cend) v -= t[u];
        }
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
    hipLaunchKernelGGL(trial_finish_kernel, dim3(2), dim3(256), 0, c->stream, c->partials.p + TRIAL_COST_POFS, ncp, c->partials.p, c->ps_np, c->partials.p + 1024, c->ps_np2,
                       c->lambda, c->scalars.p, c->d_status.p, c->h_scalars_dev, (double)(++c->trial_seq));
    HIPCHK(hipGetLastError());
    return NLLS_OK;
}

template <bool TSP = false>
static SLayoutT<TSP> make_layout(nlls_ctx* c) {
    SLayoutT<TSP> L{}; L.S = c->S.p; L.mode = c->solve_mode; L.n = (int)c->nred; L.npad = c->dense_pad128 ? (((int)c->nred + 1 + 127) / 128) * 128 : (((int)c->nred + 1 + NB - 1) / NB) * NB;
    L.n_band = (int)c->n_band; L.bw = c->bw; L.nbd = c->nbd; L.H = c->band_H;
    if constexpr (TSP) { L.npad = c->tsp.nt; L.tsp = c->tsp.d_map.p; }
    return L;
}

// local phase: assemble this rank's share of [S | s] (rank 0 also contributes the reduced-reduced blocks,
// lambda*I and b_R); under sharding the buffer is then summed over ranks
template <bool TSP>
static int enqueue_solve_local_t(nlls_ctx* c) {
    using LAY = SLayoutT<TSP>;
    const int n = (int)c->nred; if (n == 0) return NLLS_OK;
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
        GatherArgs ga{c->d_gjobs.p, c->d_gcons.p, c->n_gjobs, c->slab.p, c->A.p, c->b.p, c->Cinv.p, c->fast_dv, c->lambda, c->bcr.geom, c->d_status.p};
        hipLaunchKernelGGL(schur_gather_kernel, dim3((unsigned)((c->n_gjobs + 3) / 4)), dim3(256), 0, c->stream, ga);
        HIPCHK(hipGetLastError());
        return NLLS_OK;
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
        PrepArgs pa{c->d_red_boff.p, c->d_copy.p, c->lambda, ninit, (uint32_t)c->n_fast_groups, c->d_status.p};
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
        if (c->bcr.enqueue(c->stream, c->elim_slab ? (const double*)nullptr : c->S.p, c->s_ptr(), c->d_status.p, pivot_floor) != NLLS_OK) return herr(c, hipGetLastError(), "block cyclic reduction launch");
    } else if (band) {
        BandArgs a{}; a.Sb = c->S.p; a.Lb = c->Lwork.p; a.xr = c->s_ptr(); a.n_band = L.n_band; a.bw = L.bw; a.nbd = L.nbd; a.H = L.H; a.CH = c->band_CH; a.status = c->d_status.p;
        a.PFC = (L.bw + 1 + a.CH - 1) / a.CH + 1; a.RC = (a.PFC + 1) * a.CH; a.NSC = (L.bw + 1 + c->band_SEG - 1) / c->band_SEG;
        const int nbr = L.nbd + 1;
        const size_t lds = sizeof(double) * ((size_t)a.RC * L.H + 2 * (size_t)(2 * a.NSC * c->band_SEG + 2 * c->band_SEG) + (size_t)(L.bw + 2) * nbr + (size_t)nbr * nbr + nbr + 8);
        const int NBW = (L.bw + 15) / 16;                // tile rows below the diagonal tile that a block column reaches
        const size_t blk_lds = sizeof(double) * ((size_t)(NBW + 2) * (NBW + 2) * 272 + 272 + 2 * (size_t)(NBW + 2) * 16 * 17 + 64 + 32 * 17 + 272 + 8);
        if (c->band_blocked && NBW <= 5 && (NBW + 2) * 16 <= 128 && L.H <= 96 && blk_lds <= 160 * 1024) {
            const int nJb = (L.n_band + 15) / 16, fsz = blk_fsize(NBW, L.nbd);
            // twisted (two-sided) factorisation: two workgroups, one from each end of the band, meet at a separator of
            // ws columns, bw <= ws <= 16 NBW, so that the sides do not touch each other
            const int kk = (L.n_band - L.bw) / 16, ws = L.n_band - 16 * kk;
            const bool twisted = c->band_twisted && L.nbd == 0 && ws >= L.bw && ws <= 16 * NBW && kk >= 4 * (NBW + 2);
            const int JA = twisted ? (kk + 1) / 2 : nJb, JB = twisted ? kk / 2 : 0, cA = 16 * JA;
            double* corner = c->Lwork.p + (size_t)nJb * fsz + 128;
            double* sepA = corner + 2 * nbr * nbr; double* sepB = sepA + (size_t)(16 * NBW) * (16 * NBW) + 16 * NBW;
            BlkArgs2 bkl{};
            for (int sd = 0; sd < (twisted ? 2 : 1); ++sd) {
                BlkArgs& q = bkl.c[sd]; q.Sb = c->S.p; q.Lb = c->Lwork.p + (size_t)(sd ? JA : 0) * fsz; q.corner_out = corner + sd * nbr * nbr;
                q.sep_out = twisted ? (sd ? sepB : sepA) : nullptr; q.n_band = L.n_band; q.bw = L.bw; q.nbd = L.nbd; q.H = L.H; q.NBW = NBW; q.rev = sd; q.nJs = sd ? JB : JA; q.timing = 1; q.status = c->d_status.p;
            }
            hipLaunchKernelGGL(band_blocked_factor_kernel, dim3(twisted ? 2 : 1), dim3(BLK_T), blk_lds, c->stream, bkl);
            if (twisted) {
                // the separator: a dense ws x ws system in band layout (bandwidth ws - 1), same kernels, one workgroup
                double* Ssep = sepB + (size_t)(16 * NBW) * (16 * NBW) + 16 * NBW; double* Lsep = Ssep + (size_t)ws * (ws + 1) + 8;
                const int nJs2 = (ws + 15) / 16, NBWs = (ws - 1 + 15) / 16;
                hipLaunchKernelGGL(band_sep_combine_kernel, dim3(8), dim3(256), 0, c->stream, (const double*)c->S.p, (const double*)sepA, (const double*)sepB, cA, ws, 16 * NBW, L.bw, L.nbd, L.H, Ssep);
                BlkArgs2 bs{}; BlkArgs& q = bs.c[0]; q.Sb = Ssep; q.Lb = Lsep; q.corner_out = Lsep + (size_t)nJs2 * blk_fsize(NBWs, 0) + 128; q.sep_out = nullptr;
                q.n_band = ws; q.bw = ws - 1; q.nbd = 0; q.H = ws + 1; q.NBW = NBWs; q.rev = 0; q.nJs = nJs2; q.status = c->d_status.p;
                const size_t lds_s = sizeof(double) * ((size_t)(NBWs + 2) * (NBWs + 2) * 272 + 272 + 2 * (size_t)(NBWs + 2) * 16 * 17 + 64 + 32 * 17 + 272 + 8);
                hipLaunchKernelGGL(band_blocked_factor_kernel, dim3(1), dim3(BLK_T), lds_s, c->stream, bs);
                BwdArgs2 b2{}; BwdArgs& r = b2.c[0]; r.Lt = Lsep; r.corner_in = q.corner_out; r.xr = c->s_ptr() + cA; r.n_band = ws; r.nbd = 0; r.NBW = NBWs; r.rev = 0; r.nJs = nJs2; r.xnext = -1; r.nxnext = 0; r.status = c->d_status.p;
                const size_t lds_sb = sizeof(double) * ((size_t)(BWD_RING + 1) * bwd_slot(NBWs) + 384 + 16);
                switch (NBWs) {
                    case 1: hipLaunchKernelGGL(band_backward_tiles_kernel<1>, dim3(1), dim3(256), lds_sb, c->stream, b2); break;
                    case 2: hipLaunchKernelGGL(band_backward_tiles_kernel<2>, dim3(1), dim3(256), lds_sb, c->stream, b2); break;
                    case 3: hipLaunchKernelGGL(band_backward_tiles_kernel<3>, dim3(1), dim3(256), lds_sb, c->stream, b2); break;
                    case 4: hipLaunchKernelGGL(band_backward_tiles_kernel<4>, dim3(1), dim3(256), lds_sb, c->stream, b2); break;
                    default: hipLaunchKernelGGL(band_backward_tiles_kernel<5>, dim3(1), dim3(256), lds_sb, c->stream, b2); break;
                }
            }
            BwdArgs2 bw2{};
            for (int sd = 0; sd < (twisted ? 2 : 1); ++sd) {
                BwdArgs& q = bw2.c[sd]; q.Lt = c->Lwork.p + (size_t)(sd ? JA : 0) * fsz; q.corner_in = corner; q.xr = c->s_ptr(); q.n_band = L.n_band; q.nbd = L.nbd; q.NBW = NBW;
                q.rev = sd; q.nJs = sd ? JB : JA; q.xnext = twisted ? (sd ? cA + ws - 1 : cA) : -1; q.nxnext = ws; q.timing = 1; q.status = c->d_status.p;
            }
            const size_t lds_b = sizeof(double) * ((size_t)(BWD_RING + 1) * bwd_slot(NBW) + 384 + (size_t)nbr * nbr + nbr + 8);
            const dim3 gb(twisted ? 2 : 1);
            switch (NBW) {
                case 1: hipLaunchKernelGGL(band_backward_tiles_kernel<1>, gb, dim3(256), lds_b, c->stream, bw2); break;
                case 2: hipLaunchKernelGGL(band_backward_tiles_kernel<2>, gb, dim3(256), lds_b, c->stream, bw2); break;
                case 3: hipLaunchKernelGGL(band_backward_tiles_kernel<3>, gb, dim3(256), lds_b, c->stream, bw2); break;
                case 4: hipLaunchKernelGGL(band_backward_tiles_kernel<4>, gb, dim3(256), lds_b, c->stream, bw2); break;
                default: hipLaunchKernelGGL(band_backward_tiles_kernel<5>, gb, dim3(256), lds_b, c->stream, bw2); break;
            }
        } else
#define LAUNCH_BAND(SEG, NSLOT) hipLaunchKernelGGL((band_ldlt_solve_kernel<SEG, NSLOT>), dim3(1), dim3(256), lds, c->stream, a)
        if (c->band_SEG == 10 && c->band_NSEG == 2) LAUNCH_BAND(10, 2);
        else if (c->band_SEG == 8 && c->band_NSEG == 1) LAUNCH_BAND(8, 1);
        else if (c->band_SEG == 12 && c->band_NSEG == 2) LAUNCH_BAND(12, 2);
        else if (c->band_SEG == 12 && c->band_NSEG == 4) LAUNCH_BAND(12, 4);
        else LAUNCH_BAND(16, 4);
#undef LAUNCH_BAND
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
        if (c->elim_slab) { const BcrGeom& g = c->bcr.geom; zptr = g.ws + g.oD; zcount = (int64_t)(g.oBR + (size_t)g.N * g.NT * 256 - g.oD); }
        // an LM trial (nlls_lm_trial sets trial_to / trial_from): the retraction in this launch
        BsfRetract rt{}; unsigned nrestwg = 0;
        c->retract_done = false;
        if (c->trial_to >= 0 && c->post_fuse && c->fast_all_euclid && c->nranks == 1 && nslow == 0 && c->info.is_sparse && c->n_fast_groups > 0 && c->info.nvar > 0) {
            rt.on = 1; rt.nrest = (int)c->d_rest_var.n; rt.fast_voff = c->d_fast_voff.p; rt.rest_var = c->d_rest_var.p; rt.rest_red = c->d_rest_red.p;
            rt.vkind = c->d_var_kind.p; rt.vdim = c->d_var_dim.p; rt.voff = c->d_var_off.p; rt.vfrom = vars_ptr(c, c->trial_from); rt.vto = vars_ptr(c, c->trial_to);
            nrestwg = (unsigned)((rt.nrest + 63) / 64); c->retract_done = true;
        }
#define LAUNCH_BSF(DV) hipLaunchKernelGGL((schur_backsub_fast_kernel<DV>), dim3((unsigned)c->n_fast_groups + nextra + nrestwg), dim3(64), 0, c->stream, c->A.p, c->b.p, c->d_elim_desc.p, c->d_elim_rc.p, \
                c->Cinv.p, c->s_ptr(), c->x.p, c->tE.p, (uint32_t)c->n_fast_groups, c->d_red_boff.p, n, write_red, zptr, zcount, nextra, rt)
        if (c->n_fast_groups > 0) { if (c->fast_dv == 3) LAUNCH_BSF(3); else if (c->fast_dv == 2) LAUNCH_BSF(2); else if (c->fast_dv == 1) LAUNCH_BSF(1); c->tE_valid = true; c->S_zeroed = zero_S; }
        else c->retract_done = false;
#undef LAUNCH_BSF
    }
    HIPCHK(hipGetLastError());
    return NLLS_OK;
}

int enqueue_solve(nlls_ctx* c) {
    int rc = enqueue_solve_local(c); if (rc != NLLS_OK) return rc;
    return enqueue_solve_finish(c);
}

}  // namespace nlls
