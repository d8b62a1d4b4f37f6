// nlls_slayout.hpp -- addressing of the reduced system [S | s] and the argument blocks shared by the translation units that assemble it
// (nlls_solve.hip: the materialised elimination; nlls_mf.hip: the matrix-free one).
#pragma once
#include "nlls_wave.hpp"

namespace nlls {

constexpr int NB = 64;          // Cholesky panel width (the dense layout is padded to a multiple of it)

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

template <bool TSP = false>
inline SLayoutT<TSP> make_layout(nlls_ctx* c) {
    SLayoutT<TSP> L{}; L.S = c->S.p; L.mode = c->solve_mode; L.n = (int)c->nred; L.npad = c->dense_pad128 ? (((int)c->nred + 1 + 127) / 128) * 128 : (((int)c->nred + 1 + NB - 1) / NB) * NB;
    L.n_band = (int)c->n_band; L.bw = c->bw; L.nbd = c->nbd; L.H = c->band_H;
    if constexpr (TSP) { L.npad = c->tsp.nt; L.tsp = c->tsp.d_map.p; }
    return L;
}

// the workgroups behind the supernodes of the one-launch assembly: status reset, s += b_R, the reduced-reduced blocks (+ lambda on their diagonals) added into S
struct PrepArgs { const uint32_t* red_boff; const SchurCopy* copies; double lambda; int ninit; uint32_t nfast; int* status; double* stamps; };
template <class LAY>
NLLS_DEV void schur_prep_roles(const double* __restrict__ A, const double* __restrict__ b, const LAY& L, double* __restrict__ s, const PrepArgs& pa, int w) {
    if (w == 0 && threadIdx.x == 0) pa.status[4] = 0;                       // (pivots dropped by the floor: only the panels of this solve add to it)
    if (w < pa.ninit) {
        const int i = w * (int)blockDim.x + threadIdx.x;      // (ninit counts workgroups of the launch's own size)
        if (i < L.n) { atomicAdd(L.rhs(s, i), b[pa.red_boff[i]]); return; }
        if (!LAY_IS_TSP(L) && L.mode != SOLVE_BAND && i < L.npad) { s[i] = 0.0; L.S[(size_t)i + (size_t)L.npad * i] = (i == L.n) ? 1e300 : 1.0; }
        return;
    }
    const SchurCopy cp = pa.copies[w - pa.ninit];
    for (int e = threadIdx.x; e < cp.rows * cp.cols; e += (int)blockDim.x) {
        const int i = e % cp.rows, j = e / cp.rows;
        double v = A[cp.off + e];
        if (cp.r == cp.c) { if (i < j) continue; if (i == j) v += pa.lambda; atomicAdd(L.at(cp.r + i, cp.c + j), v); }
        else if (cp.r > cp.c) atomicAdd(L.at(cp.r + i, cp.c + j), v);
        else atomicAdd(L.at(cp.c + j, cp.r + i), v);
    }
}

// The retraction of an LM trial (update!(to, from, x), src/iterators.jl:155) rides in the back-substitution launch (on != 0): the supernode's workgroup retracts its own
// members from the step it has just formed (Euclidean variables of DV entries: checked at upload), workgroups behind the others retract every
// other variable from the reduced solution itself (x_R = -xr: the scatter of this very launch is not visible to them) -- the cost sweep is then the
// next launch, and the step statistics ride in ITS launch (nlls_post.hpp): no launch of their own for either.
struct BsfRetract { double* stamps; int on, nrest; const uint32_t* fast_voff; const uint32_t* rest_var; const int32_t* rest_red;
                    const int32_t* vkind; const int32_t* vdim; const uint32_t* voff; const double* vfrom; double* vto; };

// the workgroups behind the supernodes of a back-substitution launch (64 threads each; bid counts from the first of them): [0, nextra) scatter the reduced part of the step
// (x_R = -s) and leave the reduced system's storage zero-filled for the next solve; behind them (rt.on) one thread per variable that is no member of a fast supernode
NLLS_DEV void backsub_rest_roles(uint32_t bid, uint32_t nextra, int lane, const double* __restrict__ xr, double* __restrict__ x, const uint32_t* __restrict__ red_boff, int nred, int write_red,
                                 double* __restrict__ Szero, int64_t nzero, const BsfRetract& rt) {
    if (bid >= nextra) {
        const int j = (int)(bid - nextra) * 64 + lane; if (j >= rt.nrest) return;
        const uint32_t i = rt.rest_var[j]; const int r0 = rt.rest_red[j]; const int k = rt.vkind[i], d = rt.vdim[i]; const uint32_t o = rt.voff[i];
        if (r0 < 0) { const int st = var_storage(k, d); for (int q = 0; q < st; ++q) rt.vto[o + q] = rt.vfrom[o + q]; return; }      // fixed: copied
        retract_var_fn(k, d, o, rt.vfrom, rt.vto, [&](int q) { return write_red ? -xr[r0 + q] : 0.0; });     // (no staging array: it lived in scratch memory, 1040 bytes per lane of this launch)
        return;
    }
    for (int i = bid * 64 + lane; i < nred; i += nextra * 64) x[red_boff[i]] = write_red ? -xr[i] : 0.0;
    for (int64_t i = (int64_t)bid * 64 + lane; i < nzero; i += (int64_t)nextra * 64) Szero[i] = 0.0;
}

}  // namespace nlls
