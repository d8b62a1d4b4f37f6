// nlls_bcr.hpp -- block cyclic reduction solver for the bordered-band reduced system (see nlls_bcr.hip)
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>
#include <vector>

#include "../../include/nlls_amd.h"
#include "nlls_devbuf.hpp"

namespace nlls {

struct BcrElim { int32_t i, l, r, pad; };          // block i is eliminated between its active neighbours l and r (-1: none)
// one 16x16 output tile of a level's Schur update:  dst (-)= sum_c  Wx[a_c] Lx[b_c]'   (offsets in doubles into the workspace)
struct BcrUpd { uint32_t dst, mode, nc, pad; uint32_t a[2], b[2]; };   // mode 0: dst -= sum, 1: dst = -sum, 2: dst = sum, 3: dst = one tile product L inv(L_JJ)
// a level's active chain is the arithmetic progression o, o + s, .., o + (m - 1) s; positions first, first + 2, .. of it are eliminated
struct BcrLevel { int nelim = 0, nupd = 0; size_t elim_off = 0, upd_off = 0; int o = 0, s = 1, m = 0, first = 0; };
struct BcrChain { int o, s, m, first; };

// workspace geometry handed to the kernels
struct BcrGeom {
    double* ws; size_t oD, oA, oBR, oWx, oLx, oMx, oMd, oLd, oLi, ocp, oxb, odg;
    int NT, N, nbd, n_band, bw, H;
};

struct BcrSolver {
    int n_band = 0, bw = 0, nbd = 0, H = 0, NT = 0, N = 0;
    bool ready = false;
    std::vector<BcrLevel> levels;                 // elimination levels, the root block last
    DevBuf<double> ws; DevBuf<BcrElim> d_elim; DevBuf<BcrUpd> d_upd;
    bool fused_backward = true;
    BcrGeom geom{};
    size_t panel_lds = 0, back_lds = 0;
    int chrows_slots = 256;                       // a level's panel launch uses fewer X rows per workgroup while its workgroups still fit this many CUs (NLLS_BCR_CHROWS_SLOTS=0: always three)
    int launches = 0;
    int64_t mfma_issued = 0;                      // v_mfma_f64_16x16x4_f64 instructions one solve issues (all workgroups, redundant factorisations included)

    static bool supports(int64_t n_band, int bw, int nbd);
    int build(int64_t n_band, int bw, int nbd, int H, std::string* err, int nt = 0);   // nt > 0: tiles per block chosen by the caller (blocks of 16 nt < bw unknowns that the STRUCTURE keeps block tridiagonal)
    // Sb: band storage [S | corner] as assembled by the Schur elimination (SLayout, mode SOLVE_BAND); xr: n_band + nbd unknowns out
    // pivot_floor > 0 (undamped Newton / dogleg steps on a gauge-free problem: S is singular): a pivot that has lost more than that
    // fraction of its original diagonal entry is treated as infinite -- its unknown comes out 0 instead of (rounding) / (rounding)
    int enqueue(hipStream_t st, const double* Sb, double* xr, int* status, double pivot_floor = 0.0) const;
    void release() { ws.release(); d_elim.release(); d_upd.release(); levels.clear(); ready = false; }
};

// dense reduced system (nlls_solve.hip): panel factorisation of block column k (64 columns) and the backward pass's diagonal block
// Row window of a step of the WINDOWED dense factorisation (a reduced system that an ordering has turned into a wide band, stored densely): below a
// 128-column panel p only the 128-row blocks p + 1 .. p + nwin (the band) and strip .. (the border + right-hand side rows at the bottom) hold anything.
// Logical row block q of the step -> actual 128-row block  q < nwin ? p + 1 + q : strip + (q - nwin);  ntot = blocks of the step.  nwin < 0: no window.
struct DenseWin { int nwin = -1, strip = 0, ntot = 0; };
void launch_dense_panel(hipStream_t st, double* S, double* W, double* LiD, int npad, int k, int* status, int wide, double* Dfac, DenseWin win = DenseWin{});
void launch_dense_dcopy_all(hipStream_t st, double* S, const double* Dfac, int npad, int nwide, int first64, int n64);   // the factored diagonal blocks: slots of Dfac -> S   // wide: a 128-column panel (k counts panels of the width used)
void launch_dense_bwd_diag(hipStream_t st, const double* S, const double* LiD, int npad, int kb, int n, const double* acc, double* x);
void launch_dense_bwd_fused(hipStream_t st, const double* S, const double* LiD, double* Dinv, int npad, int n, double* x, int* status);   // the whole backward substitution in one launch (+ the diagonal blocks' inverses)
void launch_dense_bwd_step(hipStream_t st, const double* S, const double* LiD, int npad, int s, int n, double* acc, double* x);   // push block s's x into the blocks above, solve block s - 1

}  // namespace nlls
