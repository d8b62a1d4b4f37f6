// nlls_tsp.hpp -- the reduced system as a TILE-SPARSE matrix in a nested-dissection order (see nlls_tsp.hip, nlls_nd.cpp)
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>
#include <vector>

#include "../../include/nlls_amd.h"
#include "nlls_devbuf.hpp"

namespace nlls {

constexpr int TSP_TR = 128;                        // rows of a tile (= the width of a panel of the dense LDL' whose kernels the tiles reuse)
constexpr int TSP_TE = TSP_TR * TSP_TR;            // doubles of a tile (column-major)
constexpr int TSP_STRIP = 16 * TSP_TR;             // doubles of a right-hand-side strip (16 rows, row 0 used, column-major with ld 16)

// ---- symbolic phase (host only, nlls_nd.cpp) -------------------------------------------------------------------------------------------
// Nodes = the reduced blocks (cameras), dof[i] unknowns each.  Nested dissection by breadth-first level structures (George's automatic
// nested dissection: the middle level of the structure rooted at a pseudo-peripheral node separates what lies before it from what lies
// behind it), every part / separator packed into tiles of at most TSP_TR unknowns; then the symbolic LDL' of the TILE graph in that order.
struct TspSym {
    int nt = 0, nlevels = 0;
    std::vector<int32_t> tile_of, row_in_tile;     // per node: its tile, its first row inside the tile
    std::vector<int32_t> fill;                     // per tile: rows in use (the rest is padding: identity)
    std::vector<int32_t> parent, level;            // elimination tree over the tiles; level = height above the leaves (tiles of one level do not touch)
    std::vector<std::vector<int32_t>> cstruct;     // per tile k: the tiles i > k with a structurally non-zero tile (i, k) of L, ascending
    int64_t ntiles_lower = 0;                      // diagonal tiles + sum of |cstruct|
    int64_t nupd_products = 0;                     // 128^3 tile products of the whole factorisation (sum over k of |cstruct| (|cstruct| + 1) / 2)
};
// adj: symmetric adjacency among the n non-border nodes (sorted, no self loops); border nodes (coupled to everything) are appended behind them:
// dof.size() = n + nborder.  Returns false when a node has more than TSP_TR unknowns.
bool tsp_symbolic(const std::vector<std::vector<int32_t>>& adj, const std::vector<int32_t>& dof, int nborder, TspSym& out);

// ---- device side (nlls_tsp.hip) ----------------------------------------------------------------------------------------------------------
struct TspPanelJob { int64_t doff, xoff; int32_t k, xld, rx, lead; };      // one workgroup of a level's panel launch: pivot tile k (diagonal tile at doff), 16 rx rows of X at xoff (leading dimension xld)
struct TspCon { int64_t woff, loff; };                                      // one 128^3 product of an update job: C -= W(woff) L(loff)'
struct TspUpdJob { int64_t coff; int32_t con0, ncon, diag, kind; };        // kind 0: 128 x 128 target tile (diag: on the diagonal, the strictly upper 64 x 64 blocks are skipped); kind 1: right-hand-side strip; kind 2: a target tile whose contributions are split over several workgroups (atomic adds)
struct TspTrsmJob { int64_t xoff; int32_t k, pad; };                        // a tile below the pivot tile k as ONE matrix product with inv(L_kk) (levels with many tiles)
struct TspBwdJob { int32_t i, k; int64_t loff; };                           // backward pass: form x_i; k >= 0: push L_ik' x_i (tile at loff) to acc_k; k < 0: store x_i
// scheme of a level's panel: 1 / 2 = one launch, every workgroup factors the pivot tile itself and takes 1 / 2 sixteen-row chunks of a tile below it (few tiles:
// the pivot chain sets the pace); 3 = the pivot tiles alone (+ their strips), their explicit inverses, then every tile below as one matrix product (many tiles)
struct TspLevel { int npanel = 0, nupd = 0, nupd_tile = 0, nbwd = 0, ntrsm = 0, npiv = 0, scheme = 1; size_t panel0 = 0, upd0 = 0, bwd0 = 0, trsm0 = 0; };

struct TspSolver {
    bool ready = false;
    int n = 0, nt = 0;                             // reduced unknowns, tiles
    int64_t nslots = 0;                            // lower tiles stored
    std::vector<TspLevel> levels;
    DevBuf<int32_t> d_map;                         // [tpos (n) | tmap (nt * nt)]: position of a reduced unknown in tile order; slot of tile (i, j), i >= j (-1: structurally zero)
    DevBuf<int32_t> d_ipos;                        // reduced unknown at a tile-order position (-1: padding)
    DevBuf<TspPanelJob> d_panel; DevBuf<TspUpdJob> d_upd; DevBuf<TspCon> d_con; DevBuf<TspBwdJob> d_bwd; DevBuf<TspTrsmJob> d_trsm;
    DevBuf<int32_t> d_plist;                       // pivot tiles level by level (the inverses of a scheme-3 level), then the tiles of the other levels (inverted behind the last level)
    int nrest = 0; size_t rest0 = 0;
    DevBuf<int64_t> d_padpos;                      // offsets (in S) of the padding's diagonal entries
    DevBuf<double> ws;                             // W tiles + strips (as S) | LiD | Dfac | Dinv | xt | acc
    size_t oW = 0, oLiD = 0, oDfac = 0, oDinv = 0, oxt = 0, oacc = 0, odg = 0, omask = 0;
    int quad_max = 160;                            // NLLS_TSP_QUAD_MAX (A/B): target tiles of a level up to which an update workgroup takes a quarter tile
    bool chunk_masks = true;                       // NLLS_TSP_NO_MASKS=1 (A/B): every tile product in full
    int64_t npad_entries = 0;
    int launches = 0; int64_t products = 0;
    size_t s_elems() const { return (size_t)nslots * TSP_TE + (size_t)nt * TSP_STRIP; }     // tiles, then one right-hand-side strip per tile column
    // adj / nborder (as given to tsp_symbolic): the tiles the ASSEMBLY writes into (a coupling of S before fill; the border's rows; every diagonal tile) take the first
    // nslots_assembled slots -- under sharding only that prefix of the tiles is summed over ranks, the fill tiles are zero on every rank until the factorisation
    int build(const TspSym& sym, const std::vector<int32_t>& node_red_off, const std::vector<int32_t>& dof, int n_red, std::string* err,
              const std::vector<std::vector<int32_t>>* adj = nullptr, int nborder = 0);
    int64_t nslots_assembled = 0;
    // S: [tiles | strips] assembled by the elimination through SLayout::at (mode SOLVE_TSPARSE), s: the reduced right-hand side in, the solution out
    // pivot_floor > 0 (undamped Newton / dogleg steps on a gauge-free problem: S is singular): a pivot that has lost more than that fraction of its original
    // diagonal entry is treated as infinite -- its unknown gets no step -- and counted in status[4] (the rule of the band solver, nlls_bcr.hpp)
    int enqueue(hipStream_t st, double* S, double* s, int* status, double pivot_floor = 0.0) const;
    void release() { d_map.release(); d_ipos.release(); d_panel.release(); d_upd.release(); d_con.release(); d_bwd.release(); d_trsm.release(); d_plist.release(); d_padpos.release(); ws.release(); levels.clear(); ready = false; }
};

// (kernels shared with the dense LDL', nlls_bcr.hip)
void launch_tsp_panel(hipStream_t st, double* S, double* W, double* LiD, double* Dfac, const TspPanelJob* jobs, int njobs, int* status, int dch, const double* diag0, double relfloor, unsigned* mask);
void launch_tsp_dinv(hipStream_t st, const double* LiD, const double* Dfac, double* Dinv, const int32_t* list, int nlist, int nt);

}  // namespace nlls
