// nlls_bcr.hip -- the bordered-band reduced system by BLOCK CYCLIC REDUCTION (gfx950).
//
// Same contract as the rest of the solve (src/linearsolver.jl:28-32, src/iterators.jl:149-153): x solves S x = s.
// The banded reduced camera system of a bundle adjustment (6000 dof, half bandwidth 65) is a chain of n dependent
// pivots for any left-to-right LDL'; the twisted two-sided kernel of round 1 still ran 2 x 185 block steps on two of
// the 256 CUs.  Here the band is cut into N blocks of b = 16 NT >= bw columns -- a block tridiagonal matrix with a
// few border rows (dense rows ordered last, and the right-hand side as one more row) -- and reduced level by level:
// every second block of the active chain is eliminated at once (the blocks of one level do not touch each other),
// its two neighbours receive the Schur complement and become neighbours of each other.  log2 N levels instead of
// N steps; inside a level every block, and every tile of every update, is independent work for the whole chip.
// This is LDL' of S in the odd-even (nested dissection) ordering: no pivoting needed for the definite systems
// Levenberg-Marquardt produces, and no atomics anywhere -- the result is bit-reproducible.
//
// Per level, for an eliminated block i with active neighbours l and r:
//   panel   (bcr_panel_kernel, two workgroups per block)   right-looking LDL' of the 16NT-column panel
//            [ D_i ; X ],  X = [ A_il' ; A_ri ; border rows ; rhs row ]   ->  W = L Delta (D part and X rows), 1/Delta
//            16x16 tiles on v_mfma_f64_16x16x4_f64: the diagonal tile is factored in the accumulator layout together
//            with inv(L_JJ), the sub-diagonal tiles are matrix products, the next diagonal tile stays in wave 0's
//            registers (look-ahead), everything else is done by the other seven waves behind wave 0's factorisation;
//   update  (bcr_update_kernel, one wavefront per output tile)   D_l, D_r, the border/rhs rows of l and r, the new
//            coupling A_rl = -W_r Delta^-1 W_l' and the border corner's share, each  sum_J W_P,J Delta_J^-1 W_Q,J';
//   the factor is exported pre-multiplied by inv(L_JJ) (M = L inv(L_JJ)), so that the backward pass
//   (bcr_backward_kernel, levels in reverse) is matrix-vector products only:
//            x_i,J = zh_J - sum_X M_X,J' x_X - sum_{K>J} M_KJ' x_i,K .
#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "nlls_bcr.hpp"
#include "nlls_tsp.hpp"

namespace nlls {

typedef double bdouble4_t __attribute__((ext_vector_type(4)));
typedef double bdouble2_t __attribute__((ext_vector_type(2)));
constexpr int BP = 17, BTS = 16 * BP;             // LDS tile: 16 rows padded to 17 doubles
constexpr int BCR_T = 512;                        // threads of the panel kernel: wave 0 factors, seven waves help
constexpr int BCR_MAXNT = 5;
constexpr unsigned long long BCR_X_SENTINEL = 0x7ff8dead5eed1234ull;   // a NaN with a payload no arithmetic produces: "not published yet"
// A hand-off that never arrives (a workgroup that was not dispatched: dispatch order is no contract) must not hang the queue: the poll is bounded on the CONSTANT
// 100 MHz clock (s_memrealtime -- the shader clock of s_memtime / readcyclecounter moves with the power state) at half a second, and says so with a status code of its
// own (BCR_STATUS_HANDOFF_TIMEOUT: the host reports NLLS_ERR_HIP "hand-off timed out", not a bad pivot)
constexpr unsigned long long BCR_HANDOFF_TICKS = 50000000ull;          // 0.5 s at 100 MHz
constexpr int BCR_STATUS_HANDOFF_TIMEOUT = 0x40000000;

#define BCR_DEV __device__ __forceinline__

BCR_DEV double bcr_readlane(double x, int k) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), k), __builtin_amdgcn_readlane(__double2loint(x), k));
}
// workgroup barrier that waits for this wave's LDS traffic only: the exports (global stores nothing in the kernel reads back) stay in
// flight across it -- __syncthreads() would wait for every one of them to reach memory
BCR_DEV void bcr_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
BCR_DEV int bcr_dtile(int I, int K) { return I * (I + 1) / 2 + K; }      // lower tiles of a block, I >= K
// entries of the assembled reduced system read straight from its BAND storage (column j: [S(j .. j + bw, j) | border rows | rhs], SLayout of nlls_solve.hip): what
// bcr_convert_kernel re-tiles.  The first level of a damped solve reads them here instead (its panels land from the band, its update jobs take their old values
// from it): no conversion launch stands in front of the levels.
// ---------------------------------------------------------------------------------------------------
// band storage -> block tridiagonal tiles.  Tiles are 16x16 row-major ([row][col], 256 doubles):
//   D[k]  : NT(NT+1)/2 lower tiles of block k (diagonal tiles full and symmetric); columns >= n_band: identity
//   A[k]  : NT x NT tiles of S(block k, block k-1)
//   BR[k] : NT tiles, rows 0..nbd-1 the border rows, row nbd the right-hand side, for the columns of block k
//   cp[0] : the border corner (rows/cols = border index; row nbd = the border part of the rhs), full symmetric
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bcr_convert_kernel(BcrGeom g, const double* __restrict__ Sb) {
    const int NT = g.NT, ND = NT * (NT + 1) / 2, per = ND + NT * NT + NT, b = 16 * NT;
    const int bid = blockIdx.x;
    const int a = threadIdx.x & 15, bc = threadIdx.x >> 4;      // a = row inside the tile (fastest: consecutive band entries), bc = column
    if (bid == g.N * per) {
        const int nbr = g.nbd + 1; double v = 0.0;
        if (a < nbr && bc < nbr) { const int hi = a > bc ? a : bc, lo = a > bc ? bc : a; v = Sb[(size_t)g.n_band * g.H + hi + nbr * lo]; }
        g.ws[g.ocp + a * 16 + bc] = v;
        return;
    }
    const int k = bid / per, t = bid % per;
    double v = 0.0; double* dst;
    if (t < ND) {
        int I = 0; while ((I + 1) * (I + 2) / 2 <= t) ++I;
        const int K = t - I * (I + 1) / 2;
        int row = b * k + 16 * I + a, col = b * k + 16 * K + bc;
        if (row < col) { const int tmp = row; row = col; col = tmp; }
        if (row < g.n_band) { const int e = row - col; if (e <= g.bw) v = Sb[(size_t)col * g.H + e]; }
        else v = (row == col) ? 1.0 : 0.0;
        dst = g.ws + g.oD + ((size_t)k * ND + t) * 256;
        if (I == K && a == bc) g.ws[g.odg + (size_t)b * k + 16 * I + a] = fabs(v);      // the original diagonal (pivot floor of singular systems)
    } else if (t < ND + NT * NT) {
        const int tt = t - ND, P = tt / NT, Q = tt % NT;
        if (k > 0) { const int row = b * k + 16 * P + a, col = b * (k - 1) + 16 * Q + bc, e = row - col; if (row < g.n_band && e <= g.bw) v = Sb[(size_t)col * g.H + e]; }
        dst = g.ws + g.oA + ((size_t)k * NT * NT + tt) * 256;
    } else {
        const int K = t - ND - NT * NT, col = b * k + 16 * K + bc;
        if (col < g.n_band && a <= g.nbd) v = Sb[(size_t)col * g.H + g.bw + 1 + a];
        dst = g.ws + g.oBR + ((size_t)k * NT + K) * 256;
    }
    dst[a * 16 + bc] = v;
}

// ---------------------------------------------------------------------------------------------------
// panel kernel: device functions (tile layouts as in band_blocked_factor_kernel, nlls_solve.hip)
// ---------------------------------------------------------------------------------------------------
// LDL' of one 16x16 diagonal tile on the matrix cores.  The tile sits in the accumulator layout of
// v_mfma_f64_16x16x4_f64 (register r of lane (li, lk) = T[lk + 4r][li]) and is kept symmetric, so row k -- one register,
// the 16 lanes with lk = k % 4 -- is the pivot column: with every other lane zeroed it is directly the A operand
// (A[i][kk] = w_i) and, scaled by -1/d, the B operand: one MFMA is the whole rank-1 update.  A second accumulator starts
// as the identity and ends as inv(L).  Out: Wd (row-major, LDS) = the factored tile (L Delta below, Delta on the
// diagonal), Lid = inv(L)' ([k][j] = inv(L)[j][k]), dd[0..15] = Delta, dd[16..31] = 1 / Delta.
// the sixteen pivots of a tile: a chain of dependent vector instructions around the two MFMAs, nothing else.  GUARD: a vanished pivot (|d| <= fl, the floor of its
// unknown) is treated as infinite -- its rank-1 update is multiplied by 0, its unknown comes out 0; a NaN pivot is left alone (it is reported).
// 1 / d from v_rcp_f64 (measured on gfx950: relative error up to 2^-24.4) in ONE cubic step, r (1 + e + e^2) with e = 1 - d r: three dependent instructions (e, e + e^2, r + r p)
// where two Newton steps are four -- the same 1.1e-16 maximum relative error over 4 M random arguments (one Newton step: 2.2e-15).  Every dependent f64 instruction of the
// pivot chain costs ~25 cycles; this is one of seven per pivot.
BCR_DEV double bcr_refine_rcp(double d, double r) { const double e = fma(-d, r, 1.0); return fma(r, fma(e, e, e), r); }
template <bool GUARD>
BCR_DEV void bcr_pivot_chain(bdouble4_t& A, bdouble4_t& Bt, double fl, int li, int lk) {
#pragma unroll
    for (int k = 0; k < 15; ++k) {
        const int q = k & 3, r = k >> 2;
        const double w = A[r], bt = Bt[r];
        bool drop = false; if constexpr (GUARD) drop = (__ballot(fabs(w) <= fl) >> (16 * q + k)) & 1ull;     // (lane 16 q + k holds pivot k and -- li = k -- its floor)
        double dk = bcr_readlane(w, 16 * q + k);
        double rdk = __builtin_amdgcn_rcp(dk);
        const bool rowq = lk == q;
        const double am = (rowq && li > k) ? w : 0.0;
        const double bm = rowq ? bt : 0.0;
        rdk = bcr_refine_rcp(dk, rdk);
        if constexpr (GUARD) rdk = drop ? 0.0 : rdk;
        A = __builtin_amdgcn_mfma_f64_16x16x4f64(am, am * -rdk, A, 0, 0, 0);
        Bt = __builtin_amdgcn_mfma_f64_16x16x4f64(am, bm * -rdk, Bt, 0, 0, 0);
    }
}
template <bool FLOOR>
BCR_DEV void bcr_factor(const double* T0, bool from_regs, const bdouble4_t Ain, double* Wd, double* Lid, double* dd, int* status, int pivbase, bool report, double fl) {
    const int lane = threadIdx.x & 63, li = lane & 15, lk = lane >> 4;
    // FLOOR: fl = the floor of pivot li, a fraction of the ORIGINAL diagonal entry of that unknown (handed in by the caller: see bcr_panel_kernel)
    if constexpr (!FLOOR) fl = 0.0;
    bdouble4_t A, Bt;
#pragma unroll
    for (int r = 0; r < 4; ++r) { const double t = T0[(lk + 4 * r) * BP + li]; A[r] = from_regs ? Ain[r] : t; Bt[r] = (lk + 4 * r == li) ? 1.0 : 0.0; }
    // Delta is read off the diagonal of the finished tile (entry (k, k) is final once pivot k - 1 has been applied), 1 / Delta is
    // formed again by sixteen lanes at once (same instructions, same bits), zero / NaN pivots are looked for there too.
    // The floor (round 5: on EVERY solve, damped ones too) stays off the chain: the tile is factored as if no pivot could vanish, the sixteen pivots it used are held
    // against their floors afterwards -- one vector compare, one ballot -- and only a tile that met a vanished pivot is factored again from the kept copy, guarded
    // (the guard inside the chain, on every pivot of every tile: bcr_panel_kernel 18.9 against 16.2 us per level, seven levels per solve).
    const bdouble4_t A0 = A;
    bcr_pivot_chain<false>(A, Bt, 0.0, li, lk);
    auto diag_of = [&]() { double d = A[0];
#pragma unroll
        for (int r = 1; r < 4; ++r) d = (li >> 2) == r ? A[r] : d;
        return d; };
    double dsel = diag_of();
    bool dropped = false;
    if constexpr (FLOOR) {
        if (__ballot((li & 3) == lk && fabs(dsel) <= fl) != 0) {          // (uniform: rare -- lambda far below the rounding level of the diagonal)
            A = A0;
#pragma unroll
            for (int r = 0; r < 4; ++r) Bt[r] = (lk + 4 * r == li) ? 1.0 : 0.0;
            bcr_pivot_chain<true>(A, Bt, fl, li, lk);
            dsel = diag_of();
            if (fabs(dsel) <= fl) { dsel = 1e300; dropped = true; }
            if (report) { const unsigned long long m = __ballot(dropped && (li & 3) == lk); if (m != 0 && (threadIdx.x & 63) == 0) atomicAdd(status + 4, __popcll(m)); }   // status[4]: pivots dropped by the floor (nlls_get_solve_stats)
        }
    }
    if ((li & 3) == lk) {
        const double rd = bcr_refine_rcp(dsel, __builtin_amdgcn_rcp(dsel));
        dd[li] = dsel; dd[16 + li] = rd;
        if (report && !(fabs(dsel) > 0.0)) atomicCAS(status, 0, 1 + pivbase + li);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) { Wd[(lk + 4 * r) * BP + li] = A[r]; Lid[li * BP + (lk + 4 * r)] = Bt[r]; }
}
// W = T inv(L)'   (one wavefront, one tile)
BCR_DEV void bcr_panel_tile(const double* T, const double* Lid, double* Wt) {
    const int lane = threadIdx.x & 63, li = lane & 15, lk = lane >> 4;
    double av[4], bv[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) { av[kk] = T[li * BP + 4 * kk + lk]; bv[kk] = Lid[(4 * kk + lk) * BP + li]; }
    bdouble4_t acc = {0, 0, 0, 0}, acc2 = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[0], bv[0], acc, 0, 0, 0);
    acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[1], bv[1], acc2, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[2], bv[2], acc, 0, 0, 0);
    acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[3], bv[3], acc2, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 4; ++r) Wt[(lk + 4 * r) * BP + li] = acc[r] + acc2[r];
}
// wave 0 between two factorisations: W_1 = T_1 inv(L)' formed TRANSPOSED (its accumulator layout is the operand layout
// of the update), then  C -= W_1 (W_1 / Delta)'  on the next diagonal tile, which stays in registers.
BCR_DEV void bcr_panel_update_diag(const double* T, const double* Lid, const double* rd, double* Wt, const double* Ct0, bdouble4_t& diag_out) {
    const int lane = threadIdx.x & 63, li = lane & 15, lk = lane >> 4;
    double av[4], bv[4], rdk[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) { av[kk] = T[li * BP + 4 * kk + lk]; bv[kk] = Lid[(4 * kk + lk) * BP + li]; rdk[kk] = rd[4 * kk + lk]; }
    const double* Ct = Ct0 + lk * BP + li;
    bdouble4_t c, c2 = {0, 0, 0, 0};
#pragma unroll
    for (int r = 0; r < 4; ++r) c[r] = Ct[4 * r * BP];
    bdouble4_t a1 = {0, 0, 0, 0}, a2 = {0, 0, 0, 0};
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
    for (int r = 0; r < 4; ++r) Wt[li * BP + lk + 4 * r] = w[r];
#pragma unroll
    for (int r = 0; r < 4; ++r) diag_out[r] = c[r] + c2[r];
}
// MAXU tile-updates  C -= W_I (W_K / Delta)'  by one wavefront: all operand loads, then the MFMAs, then the stores.  Tiles
// are given as offsets into the workgroup's LDS (no pointer arrays: they would live in scratch memory); a job that does
// not exist works on a spare tile -- no branches, no predicated stores.
template <int MAXU>
BCR_DEV void bcr_update_batch(double* S, const int (&C)[MAXU], const int (&Wi)[MAXU], const int (&Wk)[MAXU], int rdoff) {
    const int lane = threadIdx.x & 63, li = lane & 15, lk = lane >> 4;
    const int lo = li * BP + lk, co = lk * BP + li;
    double rdk[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) rdk[kk] = S[rdoff + 4 * kk + lk];
    bdouble4_t acc[MAXU], acc2[MAXU]; double wv[MAXU][4], lv[MAXU][4];
#pragma unroll
    for (int q = 0; q < MAXU; ++q) {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) { wv[q][kk] = -S[Wi[q] + lo + 4 * kk]; lv[q][kk] = S[Wk[q] + lo + 4 * kk] * rdk[kk]; }
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[q][r] = S[C[q] + co + 4 * r * BP];
        acc2[q] = bdouble4_t{0, 0, 0, 0};
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
        for (int r = 0; r < 4; ++r) S[C[q] + co + 4 * r * BP] = acc[q][r] + acc2[q][r];
    }
}
// export of one panel tile in the MFMA operand order (lane, kk) -> [li][4 kk + lk]:  L = W / Delta to dstL (update kernel's B
// operand), W itself to dstW (its A operand) when wanted.  Plain copies: no product on the way out -- the pre-multiplied
// tiles M = L inv(L_JJ) the backward pass wants are formed by spare wavefronts of the update kernel.
BCR_DEV void bcr_export_tile(const double* Wt, const double* rd, double* __restrict__ dstW, double* __restrict__ dstL) {
    const int lane = threadIdx.x & 63, li = lane & 15, lk = lane >> 4;
    double wv[4], lv[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) { wv[m] = Wt[li * BP + 4 * m + lk]; lv[m] = wv[m] * rd[4 * m + lk]; }
    *reinterpret_cast<bdouble4_t*>(dstL + 4 * lane) = bdouble4_t{lv[0], lv[1], lv[2], lv[3]};
    if (dstW) *reinterpret_cast<bdouble4_t*>(dstW + 4 * lane) = bdouble4_t{wv[0], wv[1], wv[2], wv[3]};
}
// inv(L_JJ)' of a factored diagonal tile in the same operand order (the update kernel forms M = L inv(L_JJ) with it)
BCR_DEV void bcr_export_linv(const double* Lid, double* __restrict__ dst) {
    const int lane = threadIdx.x & 63, li = lane & 15, lk = lane >> 4;
    *reinterpret_cast<bdouble4_t*>(dst + 4 * lane) = bdouble4_t{Lid[li * BP + lk], Lid[li * BP + 4 + lk], Lid[li * BP + 8 + lk], Lid[li * BP + 12 + lk]};
}

struct BcrPanelArgs { BcrGeom g; BcrChain ch; int* status; double relfloor; int chrows; double* xr; };   // xr != nullptr: the fused backward pass follows -- the block's unknowns get the sentinel here   // chrows: X rows per workgroup of this launch (1 .. BCR_CH)
// job e of a level, from the level's chain (no descriptor load in front of everything else)
BCR_DEV BcrElim bcr_job(const BcrChain& c, int e) {
    const int idx = c.first + 2 * e, i = c.o + idx * c.s;
    return BcrElim{i, idx >= 1 ? i - c.s : -1, idx + 1 < c.m ? i + c.s : -1, 0};
}
constexpr int BCR_CH = 3;                                     // X rows (16-row tiles) per workgroup
__host__ __device__ constexpr int bcr_nchunks(int NT, int chrows = BCR_CH) { return (2 * NT + 1 + chrows - 1) / chrows; }
// X rows per workgroup of a level's panel launch: as few as still fit `slots` workgroups (one round of the chip), else BCR_CH
constexpr int bcr_level_chrows(int NT, int nelim, int slots) { for (int c2 = 1; c2 < BCR_CH; ++c2) if (bcr_nchunks(NT, c2) * nelim <= slots) return c2; return BCR_CH; }

// The X rows of an eliminated block -- NT tile rows of the left neighbour (A_il'), NT of the right one (A_ri), one of border /
// rhs rows -- are dealt over bcr_nchunks(NT) workgroups, three tile rows each; every one of them factors D_i (identical
// arithmetic, identical bits).  The tile-updates run on the matrix pipes of the three SIMDs wave 0 does not sit on: they, not
// wave 0's pivot chain, would set the pace of a block step with more X rows per workgroup.  The workgroup that holds the border
// row also exports the D part of the factor and reports bad pivots.
template <bool FLOOR>
__global__ __launch_bounds__(BCR_T) void bcr_panel_kernel(BcrPanelArgs a) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const BcrGeom& g = a.g;
    const int NT = g.NT, CHR = a.chrows, NCH = bcr_nchunks(NT, CHR), ND = NT * (NT + 1) / 2, RXT = 2 * NT + 1, PR = NT + BCR_CH;
    const BcrElim job = bcr_job(a.ch, blockIdx.x / NCH); const int ch = blockIdx.x % NCH;
    // this workgroup's X rows: global row index Rg (0..NT-1: left neighbour, NT..2NT-1: right neighbour, 2NT: border / rhs)
    static_assert(BCR_CH == 3, "the row list below is written out for three rows");
    int rw0 = 0, rw1 = 0, rw2 = 0, RX = 0;            // (scalars, not an array: a dynamically indexed array would live in scratch memory)
#pragma unroll
    for (int s2 = 0; s2 < BCR_CH; ++s2) { const int Rg = CHR * ch + s2;
        if (s2 < CHR && Rg <= 2 * NT && (Rg < NT ? job.l >= 0 : (Rg < 2 * NT ? job.r >= 0 : true))) { if (RX == 0) rw0 = Rg; else if (RX == 1) rw1 = Rg; else rw2 = Rg; ++RX; } }
    if (RX == 0) return;
    auto rowRg = [&](int R) { return R == 0 ? rw0 : (R == 1 ? rw1 : rw2); };
    const bool lead = rowRg(RX - 1) == 2 * NT;
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;   // (readfirstlane: the compiler must know the wave index is uniform, or every job decode below becomes per-lane code under exec masks)
    const int hw = wave < 4 ? wave - 1 : wave - 2;    // helper index of waves 1, 2, 3, 5, 6, 7 (wave 4 shares wave 0's SIMD -- its matrix pipe
                                                      // and its vector issue: it stays out of the way while wave 0 works)
    const bool helper = wave != 0 && wave != 4;
    double flc = 0.0;                                 // pivot floor of the tile wave 0 factors next (bcr_factor): the block's floors are put into LDS by the landing
    if (a.xr && lead) {      // the hand-off of the fused backward pass carries no flag: a dependant polls the unknowns themselves until the sentinel is gone
        if (tid < 16 * NT) { const int row = 16 * NT * job.i + tid; if (row < g.n_band) a.xr[row] = __longlong_as_double((long long)BCR_X_SENTINEL); }
        if (job.l < 0 && job.r < 0 && tid < g.nbd) g.ws[g.oxb + tid] = __longlong_as_double((long long)BCR_X_SENTINEL);      // (the root: the border's unknowns)
    }
#ifdef BCR_STAMPS
    const unsigned long long st0 = __builtin_amdgcn_s_memtime(); int stn = 0;
#define BCR_STAMP() do { if (blockIdx.x == gridDim.x - 1 && lane == 0 && wave <= 1 && stn < 20) a.status[16 + 20 * wave + stn++] = (int)(__builtin_amdgcn_s_memtime() - st0); } while (0)   // (status holds 96 ints: slots 16..55 and 56..71 are the instrumented build's)
#else
#define BCR_STAMP() do {} while (0)
#endif
    double* Dt = sm;                                  // [ND] lower tiles of D_i
    double* Xt = Dt + ND * BTS;                       // [RX][NT]
    double* Wp = Xt + BCR_CH * NT * BTS;              // [2][PR][16][BP]: panel W of block column J, by parity; rows 0..NT-1 the D part, NT.. the X rows
    double* dvec = Wp + 2 * PR * 16 * BP;             // [2][32]
    double* Li = dvec + 64;                           // [2][16][BP]
    const int oDt = 0, oXt = ND * BTS, oWp = oXt + BCR_CH * NT * BTS, odv = oWp + 2 * PR * 16 * BP, oSpare = odv + 64 + 2 * 16 * BP;   // the same as offsets; a spare tile last
    double* flds = sm + oSpare + BTS;                 // [16 NT] pivot floors of the block's unknowns (FLOOR).  Through LDS, not a load per tile: wave 0 has export STORES in flight, and a global load
                                                      // behind them waits for all of them (vmcnt counts in order) -- on the pivot chain that was +2.7 us per level
    if constexpr (FLOOR) { if (tid < 16 * NT) flds[tid] = g.ws[g.odg + 16 * NT * job.i + tid] * a.relfloor; }
    // ---- landing.  First (every thread): D_i and block column 0 of the X rows -- all that the first factorisation and the first
    // panel need.  The other X columns are requested by the six helper waves into registers now and put into LDS behind wave 0's
    // first factorisation.  Word (hi, lo) of a tile: the left neighbour's rows come in transposed.
    const double* Agl = g.ws + g.oA + (size_t)job.i * NT * NT * 256;                        // rows: block i, columns: block l
    const double* Agr = g.ws + g.oA + (size_t)(job.r >= 0 ? job.r : 0) * NT * NT * 256;   // rows: block r, columns: block i
    const double* Bg = g.ws + g.oBR + (size_t)job.i * NT * 256;
    const int bsz = 16 * NT;
    auto xsrc = [&](int Rg, int K, int hi, int lo) -> double {
        if (Rg < NT) return Agl[(size_t)(K * NT + Rg) * 256 + hi * 16 + lo];               // X(R,K)[lo][hi] = A_il(K,R)[hi][lo]
        if (Rg < 2 * NT) return Agr[(size_t)((Rg - NT) * NT + K) * 256 + hi * 16 + lo];
        return Bg[(size_t)K * 256 + hi * 16 + lo];
    };
    auto xdst = [&](int R, int Rg, int K, int hi, int lo) -> double* { return Xt + (R * NT + K) * BTS + (Rg < NT ? lo * BP + hi : hi * BP + lo); };
    constexpr int LANDQ = (BCR_CH * (BCR_MAXNT - 1) * 256 + 6 * 64 - 1) / (6 * 64);      // words per helper thread of the deferred part
    double lv[LANDQ];
    const int nB = RX * (NT - 1) * 256;
    if (helper) {
#pragma unroll
        for (int q = 0; q < LANDQ; ++q) {
            const int w = (64 * hw + lane) + q * 384; lv[q] = 0.0;
            if (w < nB) { const int idx = w >> 8, R = idx / (NT - 1), K = 1 + idx - R * (NT - 1); lv[q] = xsrc(rowRg(R), K, (w >> 4) & 15, w & 15); }
        }
    }
    {   // (all loads first, then the LDS stores: one memory round trip, not one per word)
        constexpr int DQ = (BCR_MAXNT * (BCR_MAXNT + 1) / 2 * 256 + BCR_T - 1) / BCR_T, XQ = (BCR_CH * 256 + BCR_T - 1) / BCR_T;
        const double* Dg = g.ws + g.oD + (size_t)job.i * ND * 256;
        double dv[DQ], xv[XQ];
#pragma unroll
        for (int q = 0; q < DQ; ++q) { const int w = tid + q * BCR_T; dv[q] = 0.0;
            if (w < ND * 256) {
                dv[q] = Dg[w]; } }
#pragma unroll
        for (int q = 0; q < XQ; ++q) { const int w = tid + q * BCR_T; xv[q] = w < RX * 256 ? xsrc(rowRg(w >> 8), 0, (w >> 4) & 15, w & 15) : 0.0; }
#pragma unroll
        for (int q = 0; q < DQ; ++q) { const int w = tid + q * BCR_T; if (w < ND * 256) Dt[(w >> 8) * BTS + ((w >> 4) & 15) * BP + (w & 15)] = dv[q]; }
#pragma unroll
        for (int q = 0; q < XQ; ++q) { const int w = tid + q * BCR_T; if (w < RX * 256) *xdst(w >> 8, rowRg(w >> 8), 0, (w >> 4) & 15, w & 15) = xv[q]; }
    }
    __syncthreads();
    BCR_STAMP();
    auto land_rest = [&]() {
#pragma unroll
        for (int q = 0; q < LANDQ; ++q) {
            const int w = (64 * hw + lane) + q * 384;
            if (w < nB) { const int idx = w >> 8, R = idx / (NT - 1), K = 1 + idx - R * (NT - 1); *xdst(R, rowRg(R), K, (w >> 4) & 15, w & 15) = lv[q]; }
        }
    };
    double* const Mdg = g.ws + g.oLd + (size_t)job.i * (NT * (NT - 1) / 2) * 256;   // L tiles below the diagonal
    double* const Lig = g.ws + g.oLi + (size_t)job.i * NT * 256;                     // inv(L_JJ)' per tile column
    // (W and L of the X rows live for one level only -- panel -> update: their slot is the block's position in the level's launch, so
    //  that every level rewrites the same 8 MB instead of each block keeping 220 KB of its own: the solve's footprint in the memory-side
    //  cache is what the next accumulate launch pays for)
    double* const Wxg = g.ws + g.oWx + (size_t)(blockIdx.x / NCH) * RXT * NT * 256;
    double* const Lxg = g.ws + g.oLx + (size_t)(blockIdx.x / NCH) * RXT * NT * 256;
    // tile-updates of block column Jp (panel Jp complete) other than the next diagonal tile, dealt over nh waves
    auto updates = [&](int Jp, int w0, int nh) {
        const int oWprev = oWp + (Jp & 1) * PR * 16 * BP, ord = odv + (Jp & 1) * 32 + 16;
        const int m = NT - 1 - Jp; if (m <= 0) return;
        const int nDj = m * (m + 1) / 2 - 1, ntot = nDj + RX * m;
        for (int u0 = w0; u0 < ntot; u0 += 3 * nh) {
            int C[3], Wi[3], Wk[3];
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const int u = u0 + q * nh; C[q] = oSpare; Wi[q] = oWprev; Wk[q] = oWprev;
                if (u >= ntot) continue;
                if (u < nDj) { int up = u + 1, K = Jp + 1, cnt = m; while (up >= cnt) { up -= cnt; --cnt; ++K; }
                    const int I = K + up; C[q] = oDt + bcr_dtile(I, K) * BTS; Wi[q] = oWprev + I * 16 * BP; Wk[q] = oWprev + K * 16 * BP; }
                else { int v = u - nDj, R = 0; while (v >= m) { v -= m; ++R; } const int K = Jp + 1 + v;
                    C[q] = oXt + (R * NT + K) * BTS; Wi[q] = oWprev + (NT + R) * 16 * BP; Wk[q] = oWprev + K * 16 * BP; }
            }
            bcr_update_batch<3>(sm, C, Wi, Wk, ord);
        }
    };
    auto exports = [&](int Jp, int w0, int nh) {
        const double* Wprev = Wp + (Jp & 1) * PR * 16 * BP; const double* rd = dvec + (Jp & 1) * 32 + 16; const double* Lid = Li + (Jp & 1) * 16 * BP;
        const int nDe = lead ? NT - Jp : 0, ne = nDe + RX;                   // (lead: inv(L_JJ)' and the NT-1-Jp tiles below the diagonal)
        for (int e = w0; e < ne; e += nh) {
            if (e == 0 && lead) bcr_export_linv(Lid, Lig + (size_t)Jp * 256);
            else if (e < nDe) { const int I = Jp + e; bcr_export_tile(Wprev + I * 16 * BP, rd, nullptr, Mdg + (size_t)(I * (I - 1) / 2 + Jp) * 256); }
            else { const int R = e - nDe, Rg = rowRg(R);
                bcr_export_tile(Wprev + (NT + R) * 16 * BP, rd, Wxg + (size_t)(Rg * NT + Jp) * 256, Lxg + (size_t)(Rg * NT + Jp) * 256); }
        }
    };
    bdouble4_t diag = {0, 0, 0, 0};
    for (int J = 0; J < NT; ++J) {
        double* Wb = Wp + (J & 1) * PR * 16 * BP; double* Lid = Li + (J & 1) * 16 * BP; double* db = dvec + (J & 1) * 32;
        if (wave == 0) { if constexpr (FLOOR) flc = flds[16 * J + (lane & 15)];
                         bcr_factor<FLOOR>(Dt + bcr_dtile(J, J) * BTS, J > 0, diag, Wb + J * 16 * BP, Lid, db, a.status, 16 * NT * job.i + 16 * J, lead, flc); }
        else if (helper) { if (J > 0) { BCR_STAMP(); updates(J - 1, hw, 6); BCR_STAMP(); exports(J - 1, hw, 6); } else land_rest(); }
        BCR_STAMP();
#ifdef BCR_STAMPS
        if (blockIdx.x == gridDim.x - 1 && lane == 0 && J < 2) a.status[56 + 8 * J + wave] = (int)(__builtin_amdgcn_s_memtime() - st0);
#endif
        bcr_lds_barrier();                                // diagonal tile factored; block column J final
        BCR_STAMP();
        if (J + 1 < NT) {
            if (wave == 0) bcr_panel_update_diag(Dt + bcr_dtile(J + 1, J) * BTS, Lid, db + 16, Wb + (J + 1) * 16 * BP, Dt + bcr_dtile(J + 1, J + 1) * BTS, diag);
            else if (helper) {
                const int nD = NT - J - 2;
                for (int p = hw; p < nD + RX; p += 6) {
                    if (p < nD) bcr_panel_tile(Dt + bcr_dtile(J + 2 + p, J) * BTS, Lid, Wb + (J + 2 + p) * 16 * BP);
                    else bcr_panel_tile(Xt + ((p - nD) * NT + J) * BTS, Lid, Wb + (NT + p - nD) * 16 * BP);
                }
            }
        } else {
            for (int p = wave; p < RX; p += 8) bcr_panel_tile(Xt + (p * NT + J) * BTS, Lid, Wb + (NT + p) * 16 * BP);
        }
        BCR_STAMP();
        bcr_lds_barrier();                                // panel J in LDS
    }
    exports(NT - 1, wave, 8);
    BCR_STAMP();
}

// ---------------------------------------------------------------------------------------------------
// The same panel machinery for the DENSE reduced system (S column-major, ld = npad, lower triangle; nlls_solve.hip): block
// column k of 64 columns = four tile columns.  Workgroup ch takes the tile rows 4 (k + 1) + 3 ch .. + 2 below the diagonal
// block as its X rows; every workgroup factors the 64 x 64 diagonal block itself (identical bits).  Out: L in place of the
// panel (unit lower, Delta on the diagonal of the diagonal block), W = L Delta of the rows below into the panel workspace
// (what the MFMA trailing update multiplies with), inv(L_JJ)' of the four diagonal tiles for the backward pass.
// Replaces one ldlt_diag_kernel (64 dependent pivots with two barriers each: 73 us) + trsm_panel_kernel (one thread per row,
// uncoalesced: 55 us) launch pair of round 1.
// ---------------------------------------------------------------------------------------------------
struct DensePanelArgs { double* S; double* W; double* LiD; int npad, k, T; int* status; double* Dfac; int wq, wstrip; };   // wq >= 0: windowed -- logical X tile row q (T of them) is tile row  q < wq ? NT (k + 1) + q : wstrip + (q - wq)   // Dfac: scratch for the factored diagonal block (16 NT square, column-major)
// NT tiles = 16 NT columns per panel (4: 64 columns; 8: 128 columns, the width of one pass of the trailing update -- then no narrow update and
// no second panel launch stand between two passes); DCH X tile rows per workgroup (LDS: 8 tiles of width need 2 rows to stay within 160 KB)
// TSP (tile-sparse reduced system, nlls_tsp.hip): the same panel for a pivot TILE of a level of the elimination tree -- the workgroup's job names the diagonal
// tile and the 16-row chunk(s) of a tile below it (or of the right-hand-side strip) by their offsets in the tile storage; W goes to the same offsets of a
// second buffer, the factored diagonal tile and inv(L_JJ)' to the pivot tile's slots.  One launch factors every pivot tile of a level.
struct TspPanelArgs { double* S; double* W; double* LiD; double* Dfac; const TspPanelJob* jobs; int* status; const double* diag0; double relfloor; unsigned* mask; };   // mask[tile slot]: bit c set = the 16-row chunk c of the tile holds a non-zero (the update skips products with chunks that hold none)   // diag0 / relfloor (FLOOR instantiations: undamped solves of a singular system): |original diagonal| in tile order, the fraction of it below which a pivot is dropped
template <int NT, int DCH, bool TSP = false, bool FLOOR = false>
__global__ __launch_bounds__(BCR_T) void dense_panel_kernel(std::conditional_t<TSP, TspPanelArgs, DensePanelArgs> a) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    constexpr int ND = NT * (NT + 1) / 2, PR = NT + DCH;
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int hw = wave < 4 ? wave - 1 : wave - 2; const bool helper = wave != 0 && wave != 4;
    int npad = 0, c0 = 0, RX = 0, kslot = 0, Q0 = 0, xld = 0; bool lead = false, windowed = false; int64_t dbase = 0, xbase = 0;
    if constexpr (TSP) {
        const TspPanelJob jb = a.jobs[blockIdx.x];
        kslot = jb.k; c0 = 16 * NT * jb.k; RX = jb.rx < DCH ? jb.rx : DCH; lead = jb.lead != 0; dbase = jb.doff; xbase = jb.xoff; xld = jb.xld;
    } else {
        const int ch = blockIdx.x; npad = a.npad; c0 = 16 * NT * a.k; kslot = a.k;             // first column of the panel (a.k counts panels of this width)
        // X tile rows of this workgroup: logical rows Q0 .. Q0 + RX - 1 below the diagonal block; unwindowed they are the tile rows NT (k + 1) + Q0 .. of S
        // (a.T then counts ALL tile rows of S), windowed the band's rows followed by the bottom strip (a.T counts the step's logical rows)
        windowed = a.wq >= 0;
        Q0 = DCH * ch; const int Tq = windowed ? a.T : a.T - NT * (a.k + 1);
        RX = Q0 >= Tq ? 0 : (Tq - Q0 < DCH ? Tq - Q0 : DCH);
        lead = ch == 0;
    }
    double flc = 0.0;                                 // pivot floor of the tile wave 0 factors next (bcr_factor)
    auto trow = [&](int R) { if constexpr (TSP) return 0; else { const int q = Q0 + R; return (!windowed || q < a.wq) ? NT * (a.k + 1) + q : a.wstrip + (q - a.wq); } };   // actual tile row of X row R
    if (RX == 0 && !lead) return;
    double* Dt = sm; double* Xt = Dt + ND * BTS; double* Wp = Xt + DCH * NT * BTS; double* dvec = Wp + 2 * PR * 16 * BP; double* Li = dvec + 64;
    const int oDt = 0, oXt = ND * BTS, oWp = oXt + DCH * NT * BTS, odv = oWp + 2 * PR * 16 * BP, oSpare = odv + 64 + 2 * 16 * BP;
    // pivot floors (TSP && FLOOR) of the tile's 16 NT unknowns: requested HERE, before this wavefront has any store in flight (a load behind its export stores would
    // wait for all of them: bcr_panel_kernel), kept across wave 0's lanes -- unknown u in lane u % 64 of register u / 64 -- and handed to the lanes of a tile by a
    // crossbar shuffle (this kernel's LDS is full: 160 KB at NT = 8)
    double fla = 0.0, flb = 0.0;
    if constexpr (TSP && FLOOR) if (wave == 0) { fla = a.diag0[c0 + lane] * a.relfloor; if constexpr (NT > 4) flb = a.diag0[c0 + 64 + lane] * a.relfloor; }
    // ---- landing: element (row a2, column b2) of a tile; consecutive threads walk a column of S (consecutive addresses)
    {
        constexpr int DQ = (ND * 256 + BCR_T - 1) / BCR_T, XQ = (DCH * NT * 256 + BCR_T - 1) / BCR_T;
        double dv[DQ], xv[XQ];
#pragma unroll
        for (int q = 0; q < DQ; ++q) { const int w = tid + q * BCR_T; dv[q] = 0.0;
            if (w < ND * 256) { const int t = w >> 8, b2 = (w >> 4) & 15, a2 = w & 15; int I = 0; while ((I + 1) * (I + 2) / 2 <= t) ++I; const int K = t - I * (I + 1) / 2;
                int row = 16 * I + a2, col = 16 * K + b2; if (row < col) { const int tmp = row; row = col; col = tmp; }
                if constexpr (TSP) dv[q] = a.S[dbase + row + 16 * NT * col]; else dv[q] = a.S[(size_t)(c0 + row) + (size_t)npad * (c0 + col)]; } }
#pragma unroll
        for (int q = 0; q < XQ; ++q) { const int w = tid + q * BCR_T; xv[q] = 0.0;
            if (w < RX * NT * 256) { const int t = w >> 8, b2 = (w >> 4) & 15, a2 = w & 15, R = t / NT, K = t - R * NT;
                if constexpr (TSP) xv[q] = a.S[xbase + (16 * R + a2) + (int64_t)xld * (16 * K + b2)]; else xv[q] = a.S[(size_t)(16 * trow(R) + a2) + (size_t)npad * (c0 + 16 * K + b2)]; } }
        if constexpr (TSP) {
            // which of this workgroup's 16-row chunks hold anything at all (structural zeros stay exact zeros through the factorisation: a zero row of S_ik gets
            // nothing from any update): the update kernels skip the tile products of chunks that hold nothing -- half of all products on a camera grid
            if (xld != 16 && a.mask) {
#pragma unroll
                for (int R = 0; R < DCH; ++R) { bool nz = false;
#pragma unroll
                    for (int q = 0; q < XQ; ++q) { const int w = tid + q * BCR_T; if (w < RX * NT * 256 && (w >> 8) / NT == R && xv[q] != 0.0) nz = true; }
                    if (__any(nz) && lane == 0) atomicOr(&a.mask[xbase >> 14], 1u << (int)(((xbase & 16383) >> 4) + R)); }      // (one atomic per wavefront that saw something: no barrier, no LDS)
            }
        }
#pragma unroll
        for (int q = 0; q < DQ; ++q) { const int w = tid + q * BCR_T; if (w < ND * 256) Dt[(w >> 8) * BTS + (w & 15) * BP + ((w >> 4) & 15)] = dv[q]; }
#pragma unroll
        for (int q = 0; q < XQ; ++q) { const int w = tid + q * BCR_T; if (w < RX * NT * 256) Xt[(w >> 8) * BTS + (w & 15) * BP + ((w >> 4) & 15)] = xv[q]; }
    }
    __syncthreads();
    auto updates = [&](int Jp, int w0, int nh) {
        const int oWprev = oWp + (Jp & 1) * PR * 16 * BP, ord = odv + (Jp & 1) * 32 + 16;
        const int m = NT - 1 - Jp; if (m <= 0) return;
        const int nDj = m * (m + 1) / 2 - 1, ntot = nDj + RX * m;
        for (int u0 = w0; u0 < ntot; u0 += 3 * nh) {
            int C[3], Wi[3], Wk[3];
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const int u = u0 + q * nh; C[q] = oSpare; Wi[q] = oWprev; Wk[q] = oWprev;
                if (u >= ntot) continue;
                if (u < nDj) { int up = u + 1, K = Jp + 1, cnt = m; while (up >= cnt) { up -= cnt; --cnt; ++K; }
                    const int I = K + up; C[q] = oDt + bcr_dtile(I, K) * BTS; Wi[q] = oWprev + I * 16 * BP; Wk[q] = oWprev + K * 16 * BP; }
                else { int v = u - nDj, R = 0; while (v >= m) { v -= m; ++R; } const int K = Jp + 1 + v;
                    C[q] = oXt + (R * NT + K) * BTS; Wi[q] = oWprev + (NT + R) * 16 * BP; Wk[q] = oWprev + K * 16 * BP; }
            }
            bcr_update_batch<3>(sm, C, Wi, Wk, ord);
        }
    };
    // panel tile (row-major W in LDS) of block column Jp -> S (L = W / Delta, column-major) and, for X rows, W -> the panel workspace
    auto store_tile = [&](const double* Wt, const double* rd, int R, int Jp) {
#pragma unroll
        for (int r = 0; r < 4; ++r) { const int e = lane + 64 * r, a2 = e & 15, b2 = e >> 4; const double w = Wt[a2 * BP + b2];
            if constexpr (TSP) { const int64_t o = xbase + (16 * R + a2) + (int64_t)xld * (16 * Jp + b2); a.S[o] = w * rd[b2]; a.W[o] = w; }
            else { const int grow = 16 * trow(R), gcol = c0 + 16 * Jp;
                a.S[(size_t)(grow + a2) + (size_t)npad * (gcol + b2)] = w * rd[b2];
                a.W[(size_t)(grow + a2) + (size_t)npad * (gcol - c0 + b2)] = w; } }
    };
    double* const Dfac = TSP ? a.Dfac + (size_t)kslot * (16 * NT) * (16 * NT) : a.Dfac;
    auto exports = [&](int Jp, int w0, int nh) {
        const double* Wprev = Wp + (Jp & 1) * PR * 16 * BP; const double* rd = dvec + (Jp & 1) * 32 + 16; const double* dd = dvec + (Jp & 1) * 32; const double* Lid = Li + (Jp & 1) * 16 * BP;
        const int nDe = lead ? NT - Jp + 1 : 0, ne = nDe + RX;      // lead: inv(L_JJ)', the diagonal tile itself, the NT-1-Jp tiles below it
        for (int e = w0; e < ne; e += nh) {
            if (lead && e == 0) { double* dst = a.LiD + ((size_t)kslot * NT + Jp) * 256;
#pragma unroll
                for (int r = 0; r < 4; ++r) { const int q = lane + 64 * r; dst[q] = Lid[(q >> 4) * BP + (q & 15)]; } }
            // The factored DIAGONAL BLOCK does not go into S here: every workgroup of the launch lands the original block from S, and with more
            // workgroups than the chip holds at once the late ones would land what the lead has already overwritten.  It goes to the panel's
            // slot of Dfac; dense_dcopy_all_kernel moves all of them into S behind the last panel.
            else if (lead && e == 1) {                               // the diagonal tile: unit lower L below the diagonal, Delta on it
                const double* Wt = Wprev + Jp * 16 * BP;
#pragma unroll
                for (int r = 0; r < 4; ++r) { const int q = lane + 64 * r, a2 = q & 15, b2 = q >> 4;
                    if (a2 >= b2) Dfac[(size_t)(16 * Jp + a2) + (size_t)(16 * NT) * (16 * Jp + b2)] = a2 == b2 ? dd[b2] : Wt[a2 * BP + b2] * rd[b2]; } }
            else if (e < nDe) { const int I = Jp + e - 1; const double* Wt = Wprev + I * 16 * BP;
#pragma unroll
                for (int r = 0; r < 4; ++r) { const int e2 = lane + 64 * r, a2 = e2 & 15, b2 = e2 >> 4;
                    Dfac[(size_t)(16 * I + a2) + (size_t)(16 * NT) * (16 * Jp + b2)] = Wt[a2 * BP + b2] * rd[b2]; } }
            else { const int R = e - nDe; store_tile(Wprev + (NT + R) * 16 * BP, rd, R, Jp); }
        }
    };
    bdouble4_t diag = {0, 0, 0, 0};
    for (int J = 0; J < NT; ++J) {
        double* Wb = Wp + (J & 1) * PR * 16 * BP; double* Lid = Li + (J & 1) * 16 * BP; double* db = dvec + (J & 1) * 32;
        if (wave == 0) { if constexpr (TSP && FLOOR) { flc = __shfl((J & 4) ? flb : fla, 16 * (J & 3) + (lane & 15), 64); bcr_factor<true>(Dt + bcr_dtile(J, J) * BTS, J > 0, diag, Wb + J * 16 * BP, Lid, db, a.status, c0 + 16 * J, lead, flc); }
                         else bcr_factor<false>(Dt + bcr_dtile(J, J) * BTS, J > 0, diag, Wb + J * 16 * BP, Lid, db, a.status, c0 + 16 * J, lead, 0.0); }
        else if (helper && J > 0) { updates(J - 1, hw, 6); exports(J - 1, hw, 6); }
        bcr_lds_barrier();
        if (J + 1 < NT) {
            if (wave == 0) bcr_panel_update_diag(Dt + bcr_dtile(J + 1, J) * BTS, Lid, db + 16, Wb + (J + 1) * 16 * BP, Dt + bcr_dtile(J + 1, J + 1) * BTS, diag);
            else if (helper) {
                const int nD = NT - J - 2;
                for (int p = hw; p < nD + RX; p += 6) {
                    if (p < nD) bcr_panel_tile(Dt + bcr_dtile(J + 2 + p, J) * BTS, Lid, Wb + (J + 2 + p) * 16 * BP);
                    else bcr_panel_tile(Xt + ((p - nD) * NT + J) * BTS, Lid, Wb + (NT + p - nD) * 16 * BP);
                }
            }
        } else {
            for (int p = wave; p < RX; p += 8) bcr_panel_tile(Xt + (p * NT + J) * BTS, Lid, Wb + (NT + p) * 16 * BP);
        }
        bcr_lds_barrier();
    }
    exports(NT - 1, wave, 8);
}
// the lower triangles of the factored diagonal blocks: their slots -> S, ONE launch behind the last panel (nothing reads them before the
// backward pass).  Slot of a panel = the 64-block index of its first column, 128 x 128 doubles each; block b < nwide: the 128-column panel b,
// the others: 64-column panels from 64-block first64 on.
// (sixteen workgroups per block, four elements per thread requested at once: one workgroup walking a block in a loop took 26 us -- 64 dependent round trips)
__global__ __launch_bounds__(256) void dense_dcopy_all_kernel(double* __restrict__ S, const double* __restrict__ Dfac, int npad, int nwide, int first64) {
    const int b = blockIdx.x >> 4, sl = blockIdx.x & 15, nb = b < nwide ? 128 : 64, k64 = b < nwide ? 2 * b : first64 + (b - nwide), c0 = 64 * k64;
    const double* D = Dfac + (size_t)k64 * 128 * 128;
    double v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { const int e = sl * 1024 + u * 256 + (int)threadIdx.x; v[u] = e < nb * nb ? D[e] : 0.0; }
#pragma unroll
    for (int u = 0; u < 4; ++u) { const int e = sl * 1024 + u * 256 + (int)threadIdx.x; if (e < nb * nb) { const int i = e % nb, j = e / nb; if (i >= j) S[(size_t)(c0 + i) + (size_t)npad * (c0 + j)] = v[u]; } }
}
void launch_dense_dcopy_all(hipStream_t st, double* S, const double* Dfac, int npad, int nwide, int first64, int n64) {
    if (nwide + n64 > 0) hipLaunchKernelGGL(dense_dcopy_all_kernel, dim3((unsigned)(nwide + n64) * 16), dim3(256), 0, st, S, Dfac, npad, nwide, first64);
}
template <int NT, int DCH> constexpr size_t dense_panel_lds() { return sizeof(double) * ((size_t)(NT * (NT + 1) / 2 + DCH * NT) * BTS + 2 * (size_t)(NT + DCH) * 16 * BP + 64 + 2 * 16 * BP + BTS); }
static_assert(dense_panel_lds<8, 2>() <= 160 * 1024, "the 128-column panel must fit the LDS of a CU");
// k: index of the panel in units of ITS width (64 or 128 columns)
void launch_dense_panel(hipStream_t st, double* S, double* W, double* LiD, int npad, int k, int* status, int wide, double* Dfac, DenseWin win) {
    const int T = npad / 16;
    DensePanelArgs a{S, W, LiD, npad, k, T, status, Dfac + (size_t)(wide ? 2 * k : k) * 128 * 128, -1, 0};   // the panel's slot
    if (win.nwin >= 0 && wide) {
        // windowed: the step's logical tile rows = 8 per 128-row block of the window and of the strip; one X tile row per workgroup (few rows: the pivot chain sets the pace)
        a.T = 8 * win.ntot; a.wq = 8 * win.nwin; a.wstrip = 8 * win.strip;
        static bool attrw = false;
        if (!attrw) { constexpr int ldsw = (int)dense_panel_lds<8, 1>(); (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&dense_panel_kernel<8, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, ldsw); attrw = true; }
        constexpr size_t ldsq = dense_panel_lds<8, 1>();
        hipLaunchKernelGGL((dense_panel_kernel<8, 1>), dim3((unsigned)std::max(a.T, 1)), dim3(BCR_T), ldsq, st, a);
        return;
    }
    static const int one_row_max = [] { const char* e = getenv("NLLS_DENSE_DCH1"); return e ? atoi(e) : 256; }();   // (one round of a 256-CU chip; NLLS_DENSE_DCH1=0: two rows per workgroup everywhere, for A/B runs: 4.28 instead of 4.19 ms at 6000 dof)
    if (wide && T - 8 * (k + 1) <= one_row_max && T - 8 * (k + 1) > 0) {
        // one X tile row per workgroup while that still fits one round of the chip: less helper work beside the pivot chain
        const int below = T - 8 * (k + 1);
        static bool attr1 = false;
        if (!attr1) { constexpr int ldsw = (int)dense_panel_lds<8, 1>(); (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&dense_panel_kernel<8, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, ldsw); attr1 = true; }
        constexpr size_t lds = dense_panel_lds<8, 1>();
        hipLaunchKernelGGL((dense_panel_kernel<8, 1>), dim3((unsigned)below), dim3(BCR_T), lds, st, a);
    } else if (wide) {
        const int below = T - 8 * (k + 1), nch = below > 0 ? (below + 1) / 2 : 1;
        static bool attr = false;
        if (!attr) { constexpr int ldsw = (int)dense_panel_lds<8, 2>(); (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&dense_panel_kernel<8, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, ldsw); attr = true; }
        constexpr size_t lds = dense_panel_lds<8, 2>();
        hipLaunchKernelGGL((dense_panel_kernel<8, 2>), dim3((unsigned)nch), dim3(BCR_T), lds, st, a);
    } else {
        const int below = T - 4 * (k + 1), nch = below > 0 ? (below + BCR_CH - 1) / BCR_CH : 1;
        constexpr size_t lds = dense_panel_lds<4, BCR_CH>();
        hipLaunchKernelGGL((dense_panel_kernel<4, BCR_CH>), dim3((unsigned)nch), dim3(BCR_T), lds, st, a);
    }
}
// x_k = L_kk^-T (y_k - acc_k) for the 64 unknowns of block column kb, from the four inverted diagonal tiles and the L tiles below
// them (one wavefront; replaces a 64-step lane-serial substitution)
// the tiles of a diagonal block the backward solve needs -- the L tiles below the diagonal tiles (K > J) and the four inverted diagonal
// tiles -- loaded up front: none of them depends on x, so their latency is taken once (beside the push of dense_bwd_step_kernel) instead of
// once per step of the chain
struct DenseBwdTiles { double l[6][4], li[4][4]; };
BCR_DEV void dense_bwd_diag_load(const double* __restrict__ S, const double* __restrict__ LiD, int npad, int kb, DenseBwdTiles& T) {
    const int lane = threadIdx.x & 63, c = lane & 15, gq = lane >> 4, c0 = 64 * kb;
#pragma unroll
    for (int J = 0; J < 4; ++J) {
#pragma unroll
        for (int K = J + 1; K < 4; ++K)
#pragma unroll
            for (int q = 0; q < 4; ++q) T.l[J * 3 - J * (J - 1) / 2 + (K - J - 1)][q] = S[(size_t)(c0 + 16 * K + gq + 4 * q) + (size_t)npad * (c0 + 16 * J + c)];
        const double* Lk = LiD + ((size_t)kb * 4 + J) * 256 + c * 16;
#pragma unroll
        for (int q = 0; q < 4; ++q) T.li[J][q] = Lk[gq + 4 * q];
    }
}
// (accb: the 64 entries of acc that belong to block kb -- in global memory or in LDS)
BCR_DEV void dense_bwd_diag_chain(const double* __restrict__ S, int npad, int kb, int n, const double* accb, double* __restrict__ x, const DenseBwdTiles& T,
                                  double* r, double* xs, double* uu) {
    const int lane = threadIdx.x & 63, c = lane & 15, gq = lane >> 4, c0 = 64 * kb;
    { const int g = c0 + lane; r[lane] = (g < n) ? S[(size_t)n + (size_t)npad * g] - accb[lane] : 0.0; xs[lane] = 0.0; }     // y: row n of the factor
    __syncthreads();
#pragma unroll
    for (int J = 3; J >= 0; --J) {
        double s = 0.0;
#pragma unroll
        for (int K = J + 1; K < 4; ++K) {
#pragma unroll
            for (int q = 0; q < 4; ++q) s = fma(T.l[J * 3 - J * (J - 1) / 2 + (K - J - 1)][q], xs[16 * K + gq + 4 * q], s);
        }
        s += __shfl_xor(s, 16, 64); s += __shfl_xor(s, 32, 64);
        if (gq == 0) uu[c] = r[16 * J + c] - s;
        __syncthreads();
        double xv = 0.0;
#pragma unroll
        for (int q = 0; q < 4; ++q) xv = fma(T.li[J][q], uu[gq + 4 * q], xv);
        xv += __shfl_xor(xv, 16, 64); xv += __shfl_xor(xv, 32, 64);
        if (gq == 0) { const int g = c0 + 16 * J + c; const double v = g < n ? xv : 0.0; xs[16 * J + c] = v; if (g < n) x[g] = v; }
        __syncthreads();
    }
}
__global__ __launch_bounds__(64) void dense_bwd_diag_kernel(const double* __restrict__ S, const double* __restrict__ LiD, int npad, int kb, int n, const double* __restrict__ acc, double* __restrict__ x) {
    __shared__ double r[64], xs[64], uu[16];
    DenseBwdTiles T; dense_bwd_diag_load(S, LiD, npad, kb, T);
    dense_bwd_diag_chain(S, npad, kb, n, acc + 64 * kb, x, T, r, xs, uu);
}
// One launch per step of the backward substitution L' x = z: block s has just been solved (x_s final).  Workgroup j < s pushes its
// contribution into the 64 entries of block j -- acc_j += L(block s, block j)' x_s: every block column is owned by one workgroup per
// launch, so no atomics -- and the workgroup of block s - 1, whose acc is complete with that, goes straight on to solve it (its
// other three wavefronts retire first: the diagonal-block solve is one wavefront's work, and its tiles were requested at the start).
__global__ __launch_bounds__(256) void dense_bwd_step_kernel(const double* __restrict__ S, const double* __restrict__ LiD, int npad, int s, int n, double* __restrict__ acc, double* __restrict__ x) {
    __shared__ double red[4][64]; __shared__ double r[64], xs[64], uu[16];
    const int j = blockIdx.x, t = threadIdx.x, c = t & 63, q = t >> 6;
    DenseBwdTiles T;
    const bool solver = j == s - 1 && q == 0;
    if (solver) dense_bwd_diag_load(S, LiD, npad, s - 1, T);
    const double* P = S + (size_t)s * 64 + (size_t)npad * ((size_t)j * 64 + c);       // column j*64 + c, rows of block s: contiguous
    double v = 0.0;
#pragma unroll 4
    for (int i = 16 * q; i < 16 * q + 16; ++i) { const int gi = s * 64 + i; if (gi < n) v = fma(P[i], x[gi], v); }
    red[q][c] = v;
    __syncthreads();
    if (q != 0) return;
    acc[j * 64 + c] += red[0][c] + red[1][c] + red[2][c] + red[3][c];
    if (j != s - 1) return;
    __threadfence_block();
    dense_bwd_diag_chain(S, npad, s - 1, n, acc + 64 * (s - 1), x, T, r, xs, uu);    // (one wavefront left in this workgroup: its barriers are its own)
}
void launch_dense_bwd_diag(hipStream_t st, const double* S, const double* LiD, int npad, int kb, int n, const double* acc, double* x) {
    hipLaunchKernelGGL(dense_bwd_diag_kernel, dim3(1), dim3(64), 0, st, S, LiD, npad, kb, n, acc, x);
}
void launch_dense_bwd_step(hipStream_t st, const double* S, const double* LiD, int npad, int s, int n, double* acc, double* x) {
    hipLaunchKernelGGL(dense_bwd_step_kernel, dim3((unsigned)s), dim3(256), 0, st, S, LiD, npad, s, n, acc, x);
}

// ---------------------------------------------------------------------------------------------------
// Schur update of a level: one wavefront per output tile, operands straight from the exported panels (L2 / MALL):
//   dst (-)= sum_c  sum_J  Wx[a_c][J] Lx[b_c][J]'          (Lx = Wx / Delta)
// ---------------------------------------------------------------------------------------------------
template <int NT>
__device__ __forceinline__ void bcr_update_body(double* __restrict__ ws, const BcrUpd* __restrict__ jobs, int njobs) {
    const int j = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); if (j >= njobs) return;
    const BcrUpd u = jobs[j];
    const int lane = threadIdx.x & 63, li = lane & 15, lk = lane >> 4;
    if (u.mode == 3) {                                        // one tile of the factor pre-multiplied for the backward pass:  M = L inv(L_JJ), row-major
        const bdouble4_t av = *reinterpret_cast<const bdouble4_t*>(ws + u.a[0] + 4 * lane), bv = *reinterpret_cast<const bdouble4_t*>(ws + u.b[0] + 4 * lane);
        bdouble4_t acc = {0, 0, 0, 0}, acc2 = {0, 0, 0, 0};
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[0], bv[0], acc, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[1], bv[1], acc2, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[2], bv[2], acc, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[3], bv[3], acc2, 0, 0, 0);
        double* dst = ws + u.dst + lk * 16 + li;
#pragma unroll
        for (int r = 0; r < 4; ++r) dst[64 * r] = acc[r] + acc2[r];
        return;
    }
    // every operand load of the job is issued before the first MFMA (one memory round trip per job); a job with one
    // contribution reads the same tiles twice and counts the second pass with weight 0
    bdouble4_t av[2][NT], bv[2][NT];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const int cc = c < (int)u.nc ? c : 0;
        const double* A = ws + u.a[cc] + 4 * lane; const double* B = ws + u.b[cc] + 4 * lane;
#pragma unroll
        for (int J = 0; J < NT; ++J) { av[c][J] = *reinterpret_cast<const bdouble4_t*>(A + 256 * J); bv[c][J] = *reinterpret_cast<const bdouble4_t*>(B + 256 * J); }
    }
    double* dst = ws + u.dst + lk * 16 + li;
    double old[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) old[r] = u.mode == 0 ? dst[64 * r] : 0.0;
    bdouble4_t acc = {0, 0, 0, 0}, acc2 = {0, 0, 0, 0};
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        if (c >= (int)u.nc) break;
#pragma unroll
        for (int J = 0; J < NT; ++J) {
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[c][J][0], bv[c][J][0], acc, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[c][J][1], bv[c][J][1], acc2, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[c][J][2], bv[c][J][2], acc, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[c][J][3], bv[c][J][3], acc2, 0, 0, 0);
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const double v = acc[r] + acc2[r];
        dst[64 * r] = u.mode == 2 ? v : old[r] - v;
    }
}
template <int NT>
__global__ __launch_bounds__(256) void bcr_update_kernel(double* __restrict__ ws, const BcrUpd* __restrict__ jobs, int njobs) { bcr_update_body<NT>(ws, jobs, njobs); }

// ---------------------------------------------------------------------------------------------------
// backward pass of one level: one workgroup per block eliminated at that level
// ---------------------------------------------------------------------------------------------------
struct BcrBackArgs { BcrGeom g; BcrChain ch; double* xr; int root; int* status; const BcrElim* elims; };
// unknowns that cross workgroups INSIDE one launch (FUSED): relaxed agent-scope accesses -- they go past the (per-XCD, mutually incoherent) L2s
// to the memory side, so that no cache has to be written back or invalidated; every word is its own message (8-byte stores are single-copy atomic)
BCR_DEV void bcr_xstore(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
BCR_DEV double bcr_xload(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// NT wavefronts: wave J owns the 16 unknowns of tile column J.  Every load of the factor is issued before anything is
// waited for (one memory round trip per level); the unknowns of the neighbours (and of the border) go through LDS.
// FUSED: ONE launch for the whole backward pass.  Workgroup w takes the block eliminated (N - 1 - w)-th: the root first, then level by level
// down to the first one -- whatever a workgroup waits for belongs to a workgroup with a smaller index, which was dispatched before it, so the
// wait ends whether or not all of them fit the chip at once.  No flags: the unknowns were pre-set to a sentinel by the panel launches (a NaN with a
// payload no arithmetic produces), a block publishes them with single 8-byte stores, and its dependants -- their own factor tiles requested long
// before -- poll the very words they need.
template <int NT, bool FUSED>
__global__ __launch_bounds__(64 * NT) void bcr_backward_kernel(BcrBackArgs a) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const BcrGeom& g = a.g;
    const BcrElim job = FUSED ? a.elims[g.N - 1 - (int)blockIdx.x] : bcr_job(a.ch, blockIdx.x);
    const bool root = FUSED ? blockIdx.x == 0 : a.root != 0;
    constexpr int RXT = 2 * NT + 1, NO = NT * (NT - 1) / 2, b = 16 * NT, NTH = 64 * NT;
    const int nbd = g.nbd, nbr = nbd + 1;
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, c = lane & 15, gq = lane >> 4;
    double* xs = sm;                               // [RXT][16]
    double* red = xs + RXT * 16;                   // [NT][4][16]
    double* tt = red + NT * 64;                    // [NT][16]
    double* xi = tt + NT * 16;                     // [NT][16]
    double* Mdl = xi + NT * 16;                    // [NO][256]
    double* Cl = Mdl + NO * 256;                   // [16][16] corner (root only)
    double* xb = Cl + 256;                         // [16]
    // this wave's column of the X part of the factor: rows gq, gq + 4, .. of every tile (absent neighbours: block i's own tiles, weight 0)
    const double* Mxg = g.ws + g.oMx + (size_t)job.i * RXT * NT * 256 + (size_t)wave * 256 + c;
    double mx[RXT][4];
#pragma unroll
    for (int R = 0; R < RXT; ++R) {
#pragma unroll
        for (int q = 0; q < 4; ++q) mx[R][q] = Mxg[(size_t)R * NT * 256 + (gq + 4 * q) * 16];
    }
    // the unknowns of the neighbours and of the border that this block's rows multiply: requested NOW, beside the factor's tiles (one memory
    // round trip per level instead of two; the root forms the border unknowns itself first and takes the loop below)
    double xpre = 0.0;
    if (!FUSED && !root && tid < RXT * 16) {
        const int R = tid >> 4, q = tid & 15;
        if (R < NT) { if (job.l >= 0) { const int row = b * job.l + 16 * R + q; if (row < g.n_band) xpre = a.xr[row]; } }
        else if (R < 2 * NT) { if (job.r >= 0) { const int row = b * job.r + 16 * (R - NT) + q; if (row < g.n_band) xpre = a.xr[row]; } }
        else xpre = q < nbd ? g.ws[g.oxb + q] : (q == nbd ? -1.0 : 0.0);
    }
    {   // the block's own triangle: global -> registers -> LDS, every load in flight at once
        constexpr int MQ = (NO * 256 + NTH - 1) / NTH;
        const double* Mdg = g.ws + g.oMd + (size_t)job.i * NO * 256;
        double mdv[MQ > 0 ? MQ : 1];
#pragma unroll
        for (int q = 0; q < MQ; ++q) { const int w = tid + q * NTH; mdv[q] = w < NO * 256 ? Mdg[w] : 0.0; }
#pragma unroll
        for (int q = 0; q < MQ; ++q) { const int w = tid + q * NTH; if (w < NO * 256) Mdl[w] = mdv[q]; }
    }
    if (FUSED && !root) {
        // wait for the neighbours' and the border's unknowns: every lane polls the very word it needs until the sentinel is gone (the factor's loads above are
        // in flight meanwhile) -- one memory round trip per hop, where "unknowns, then a flag" needed three
        if (tid < RXT * 16) {
            const int R = tid >> 4, q = tid & 15; const double* src = nullptr;
            if (R < NT) { if (job.l >= 0) { const int row = b * job.l + 16 * R + q; if (row < g.n_band) src = a.xr + row; } }
            else if (R < 2 * NT) { if (job.r >= 0) { const int row = b * job.r + 16 * (R - NT) + q; if (row < g.n_band) src = a.xr + row; } }
            else if (q < nbd) src = g.ws + g.oxb + q; else xpre = q == nbd ? -1.0 : 0.0;
            if (src) {   // (bounded: half a second on the constant 100 MHz clock, then the solve is flagged with BCR_STATUS_HANDOFF_TIMEOUT instead of hanging the queue)
                const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
                for (;;) { xpre = bcr_xload(src); if ((unsigned long long)__double_as_longlong(xpre) != BCR_X_SENTINEL) break;
                    __builtin_amdgcn_s_sleep(1);
                    if (__builtin_amdgcn_s_memrealtime() - t0 > BCR_HANDOFF_TICKS) { atomicCAS(a.status, 0, BCR_STATUS_HANDOFF_TIMEOUT); break; } }
            }
        }
    }
    if (root) {
        if (nbd > 0) {
            // corner = cp[0] - sum_i cp[1 + i], blocks in index order (fixed order: reproducible)
            // (round 5: the N blocks' shares dealt over the lanes -- each entry of rows 0 .. nbd, the rows read below, as `parts` partial sums formed side by side and combined in a
            //  fixed order: reproducible.  One thread per entry stepping through the N shares was a chain of N dependent loads on the path every other block of the backward pass
            //  waits for: 28 us at BASELINE config 5 -- the adaptive kernel's variable is the border -- where six levels without a border take ~18.)
            {
                const int ne = 16 * nbr; int parts = NTH / ne; if (parts > 8) parts = 8; if (parts * ne > NT * 64) parts = (NT * 64) / ne;     // (the partial sums go through `red`: NT * 64 doubles, not in use yet)
                if (parts >= 2) {
                    const int e = tid % ne, pt = tid / ne;
                    if (pt < parts) {
                        double v0 = 0.0, v1 = 0.0; int i = pt;
                        for (; i + parts < g.N; i += 2 * parts) { v0 += g.ws[g.ocp + (size_t)(1 + i) * 256 + e]; v1 += g.ws[g.ocp + (size_t)(1 + i + parts) * 256 + e]; }
                        if (i < g.N) v0 += g.ws[g.ocp + (size_t)(1 + i) * 256 + e];
                        red[pt * ne + e] = v0 + v1;
                    }
                    __syncthreads();
                    if (tid < ne) { double v = g.ws[g.ocp + tid]; for (int q = 0; q < parts; ++q) v -= red[q * ne + tid]; Cl[tid] = v; }
                } else {
                    for (int e = tid; e < 256; e += NTH) { double v = g.ws[g.ocp + e]; for (int i = 0; i < g.N; ++i) v -= g.ws[g.ocp + (size_t)(1 + i) * 256 + e]; Cl[e] = v; }
                }
            }
            __syncthreads();
            if (tid == 0) {                        // LDL' of the nbd x nbd corner with the rhs row riding along (Cl[row][col])
                for (int j = 0; j < nbd; ++j) {
                    double d = Cl[j * 16 + j];
                    if (d == 0.0 || d != d) { atomicCAS(a.status, 0, 1 + g.n_band + j); d = 1.0; }
                    for (int c2 = j + 1; c2 < nbd; ++c2) { const double f = Cl[c2 * 16 + j] / d; for (int i = c2; i < nbr; ++i) Cl[i * 16 + c2] -= Cl[i * 16 + j] * f; }
                    for (int i = j + 1; i < nbr; ++i) Cl[i * 16 + j] /= d;
                    Cl[j * 16 + j] = d;
                }
                for (int r = nbd - 1; r >= 0; --r) { double v2 = Cl[nbd * 16 + r]; for (int r2 = r + 1; r2 < nbd; ++r2) v2 -= Cl[r2 * 16 + r] * xb[r2]; xb[r] = v2; }
            }
            __syncthreads();
            if (tid < nbd) { if (FUSED) bcr_xstore(g.ws + g.oxb + tid, xb[tid]); else g.ws[g.oxb + tid] = xb[tid]; a.xr[g.n_band + tid] = xb[tid]; }
        }
    } else if (!FUSED && tid < nbd) xb[tid] = g.ws[g.oxb + tid];
    __syncthreads();
    if (!root) { if (tid < RXT * 16) xs[tid] = xpre; }
    else for (int t = tid; t < RXT * 16; t += NTH) {
        const int R = t >> 4, q = t & 15; double v = 0.0;
        if (R < NT) { if (job.l >= 0) { const int row = b * job.l + 16 * R + q; if (row < g.n_band) v = a.xr[row]; } }
        else if (R < 2 * NT) { if (job.r >= 0) { const int row = b * job.r + 16 * (R - NT) + q; if (row < g.n_band) v = a.xr[row]; } }
        else v = q < nbd ? xb[q] : (q == nbd ? -1.0 : 0.0);      // the rhs row is a border row whose unknown is -1
        xs[t] = v;
    }
    __syncthreads();
    {
        double acc = 0.0;
#pragma unroll
        for (int R = 0; R < RXT; ++R) {
#pragma unroll
            for (int q = 0; q < 4; ++q) acc = fma(mx[R][q], xs[16 * R + gq + 4 * q], acc);
        }
        red[(wave * 4 + gq) * 16 + c] = acc;
    }
    __syncthreads();
    if (tid < NT * 16) { const int J = tid >> 4, cc = tid & 15; tt[tid] = -((red[(J * 4 + 0) * 16 + cc] + red[(J * 4 + 1) * 16 + cc]) + (red[(J * 4 + 2) * 16 + cc] + red[(J * 4 + 3) * 16 + cc])); }
    __syncthreads();
    // the block's own triangle: x_J = t_J - sum_{K > J} M_KJ' x_K, J = NT-1 .. 0: wave 0 alone (the others retire: the barrier of each step
    // then only orders this wave's own LDS traffic)
    if (wave != 0) return;
    for (int J = NT - 1; J >= 0; --J) {
        {
            double s = 0.0;
            for (int K = J + 1; K < NT; ++K) {
                const double* M = Mdl + (K * (K - 1) / 2 + J) * 256 + c;
#pragma unroll
                for (int q = 0; q < 4; ++q) s = fma(M[(gq + 4 * q) * 16], xi[16 * K + gq + 4 * q], s);
            }
            s += __shfl_xor(s, 16, 64); s += __shfl_xor(s, 32, 64);
            if (gq == 0) {
                const double x = tt[16 * J + c] - s; xi[16 * J + c] = x;
                const int row = b * job.i + 16 * J + c; if (row < g.n_band) { if (FUSED) bcr_xstore(a.xr + row, x); else a.xr[row] = x; }
            }
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------
// DENSE reduced system: the backward substitution L' x = z in ONE launch (was one launch per 64-column block: 93 dependent launches of 6.7 us at
// 6000 dof).  128-column blocks; workgroup w owns block j = NBB - 1 - w: it accumulates  acc_j = sum_{s > j} L(s, j)' x_s  in registers, hop by hop
// as the blocks below it publish their unknowns, the tile of the NEXT hop requested before this one's unknowns are waited for; then
// x_j = inv(L_jj)' (z_j - acc_j)  with the EXPLICIT inverse of the unit-lower diagonal block (dense_dinv_kernel: once per solve, all blocks at once)
// -- one matrix-vector product instead of eight dependent tile steps.  The hand-off carries no flag: x is pre-set to a sentinel (a NaN with a
// payload no arithmetic produces) by the launch in front, a block's unknowns are single 8-byte relaxed agent-scope stores past the (per-XCD,
// mutually incoherent) L2s, and a dependant polls the very words it needs -- one memory round trip per hop instead of three (unknowns out, flag
// out, flag seen, unknowns in).  Whatever a workgroup waits for belongs to a workgroup with a smaller index.
// ---------------------------------------------------------------------------------------------------
constexpr int DBB = 128;
// inverse of every 128 x 128 unit-lower diagonal block of the factor (column-major, ld = npad) by 16 x 16 tiles on the matrix cores:
//   X_JJ = inv(L_JJ) (exported by the panels: LiD),   X_IJ = -X_II sum_{K = J}^{I-1} L_IK X_KJ,   tile row by tile row, in place in LDS
// (row I of L is read for the last time when row I of X is formed).  The sum leaves the matrix cores in the accumulator layout, which IS the B
// operand layout of the product with X_II: no LDS round trip between the two.  Tiles beyond npad: identity.  Out: Dinv[b] column-major 128 x 128,
// ones on the diagonal, zeros above it.  Also: the sentinel into x[0, n).
// ONE: a single block b1 whose factored tiles are still in its panel's scratch slot (Lslot, column-major 128 x 128: the look-ahead factorisation inverts a block
// the moment it is factored -- the rows below it are then ONE matrix product with the inverse); no sentinel.
// ONE == 2 (tile-sparse reduced system): the pivot tiles list[blockIdx.x], each one's factored block in its own slot (Lslot + 128 x 128 per tile), no sentinel.
template <int ONE>
__global__ __launch_bounds__(512) void dense_dinv_kernel(const double* __restrict__ S, const double* __restrict__ LiD, double* __restrict__ Dinv, double* __restrict__ x, int npad, int n, int b1, const double* __restrict__ Lslot,
                                                         const int32_t* __restrict__ list = nullptr) {
    extern __shared__ __attribute__((aligned(16))) double sm[];          // the 36 lower tiles of the block, [16][BP] each
    const int b = ONE == 1 ? b1 : (ONE == 2 ? list[blockIdx.x] : (int)blockIdx.x), t = threadIdx.x, c0 = DBB * b;
    if (ONE == 2) Lslot += (size_t)b * DBB * DBB;
    if (!ONE && t < DBB && c0 + t < n) x[c0 + t] = __longlong_as_double((long long)BCR_X_SENTINEL);
    for (int e0 = t; e0 < DBB * DBB; e0 += 8 * 512) {      // consecutive threads walk a column of S; eight loads in flight per thread
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int e = e0 + 512 * u, i = e & 127, j = e >> 7, I = i >> 4, J = j >> 4; v[u] = 0.0;
            if (I == J) { const int gt = 8 * b + I; v[u] = gt < npad / 16 ? LiD[(size_t)gt * 256 + (i & 15) + 16 * (j & 15)] : ((i & 15) == (j & 15) ? 1.0 : 0.0); }
            else if (I > J && c0 + i < npad) v[u] = ONE ? Lslot[(size_t)i + (size_t)DBB * j] : S[(size_t)(c0 + i) + (size_t)npad * (c0 + j)]; }
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int e = e0 + 512 * u, i = e & 127, j = e >> 7, I = i >> 4, J = j >> 4;
            if (I >= J) sm[bcr_dtile(I, J) * BTS + (i & 15) * BP + (j & 15)] = v[u]; }
    }
    __syncthreads();
    const int J = __builtin_amdgcn_readfirstlane(t >> 6), lane = t & 63, li = lane & 15, lk = lane >> 4;
    for (int I = 1; I < 8; ++I) {
        bdouble4_t res = {0, 0, 0, 0}, res2 = {0, 0, 0, 0};
        if (J < I) {
            bdouble4_t acc = {0, 0, 0, 0}, acc2 = {0, 0, 0, 0};
            for (int K = J; K < I; ++K) {
                const double* A = sm + bcr_dtile(I, K) * BTS; const double* B = sm + bcr_dtile(K, J) * BTS;
                double av[4], bv[4];
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) { av[kk] = A[li * BP + 4 * kk + lk]; bv[kk] = B[(4 * kk + lk) * BP + li]; }
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[0], bv[0], acc, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[1], bv[1], acc2, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[2], bv[2], acc, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[3], bv[3], acc2, 0, 0, 0);
            }
            const double* D = sm + bcr_dtile(I, I) * BTS;
            double dv[4];
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) dv[kk] = D[li * BP + 4 * kk + lk];
            res = __builtin_amdgcn_mfma_f64_16x16x4f64(dv[0], acc[0] + acc2[0], res, 0, 0, 0);        // (register kk of the sum = rows 4 kk + lk: the B operand of k-slice kk)
            res2 = __builtin_amdgcn_mfma_f64_16x16x4f64(dv[1], acc[1] + acc2[1], res2, 0, 0, 0);
            res = __builtin_amdgcn_mfma_f64_16x16x4f64(dv[2], acc[2] + acc2[2], res, 0, 0, 0);
            res2 = __builtin_amdgcn_mfma_f64_16x16x4f64(dv[3], acc[3] + acc2[3], res2, 0, 0, 0);
        }
        __syncthreads();               // row I of L has been read for the last time
        if (J < I) { double* X = sm + bcr_dtile(I, J) * BTS;
#pragma unroll
            for (int r = 0; r < 4; ++r) X[(lk + 4 * r) * BP + li] = -(res[r] + res2[r]); }
        __syncthreads();
    }
    double* out = Dinv + (size_t)b * DBB * DBB;
    for (int e = t; e < DBB * DBB; e += 512) { const int r = e & 127, c = e >> 7; out[e] = (r >> 4) >= (c >> 4) ? sm[bcr_dtile(r >> 4, c >> 4) * BTS + (r & 15) * BP + (c & 15)] : 0.0; }
}
struct DenseBwdArgs { const double* S; const double* Dinv; double* x; int* status; int npad, n, NBB; };
// Thread t: rows 32 (t & 3) .. of column t >> 2 of a tile (contiguous in memory).  The tile of the next hop is requested right BEHIND this hop's
// products: a wave's loads return in order, and a poll issued behind a tile request waits for all of it -- placed here the request is long served
// when the workgroup whose turn is next looks for its unknowns (it has been waiting a whole hop), and only workgroups with slack poll behind it.
// The block's inverse stays in registers from the start.
__global__ __launch_bounds__(512) void dense_bwd_fused_kernel(DenseBwdArgs a) {
    __shared__ double xs[2][DBB], us[DBB];
    const int t = threadIdx.x, c = t >> 2, q = t & 3, npad = a.npad, n = a.n;
    const int j = a.NBB - 1 - (int)blockIdx.x, gc = DBB * j + c;
    const bool colok = gc < npad;
    // (rows / columns beyond npad -- the half-empty last block of an odd number of 64-blocks -- are not multiplied at all: tile_ok)
    auto tile_ok = [&](int s) { return colok && DBB * s + 32 * q < npad; };
    // (macros, not lambdas over the arrays: an array whose address is taken lives in scratch memory)
#define DBW_LOAD(T, P) do { _Pragma("unroll") for (int i = 0; i < 32; i += 2) { const bdouble2_t w = *reinterpret_cast<const bdouble2_t*>((P) + i); T[i] = w[0]; T[i + 1] = w[1]; } } while (0)
#define DBW_TILE(T, s) do { const double* P_ = tile_ok(s) ? a.S + (size_t)(DBB * (s) + 32 * q) + (size_t)npad * gc : a.S; DBW_LOAD(T, P_); } while (0)
    // sum_i T[i] v[i], v in LDS: the 32 words first (eight 32-byte reads in flight), then four independent chains
#define DBW_DOT(OUT, T, V) do { double v_[32]; _Pragma("unroll") for (int i = 0; i < 32; i += 4) { const bdouble4_t w = *reinterpret_cast<const bdouble4_t*>((V) + i); v_[i] = w[0]; v_[i + 1] = w[1]; v_[i + 2] = w[2]; v_[i + 3] = w[3]; } \
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0; _Pragma("unroll") for (int i = 0; i < 32; i += 4) { s0 = fma(T[i], v_[i], s0); s1 = fma(T[i + 1], v_[i + 1], s1); s2 = fma(T[i + 2], v_[i + 2], s2); s3 = fma(T[i + 3], v_[i + 3], s3); } \
        OUT = (s0 + s1) + (s2 + s3); } while (0)
    double T[32], INV[32];
    if (j + 1 < a.NBB) DBW_TILE(T, a.NBB - 1);
    { const double* Pi = a.Dinv + (size_t)j * DBB * DBB + (size_t)DBB * c + 32 * q; DBW_LOAD(INV, Pi); }     // rows 32 q .. of column c of inv(L_jj)
    const double z = (q == 0 && gc < n) ? a.S[(size_t)n + (size_t)npad * gc] : 0.0;       // z = D^-1 L^-1 s: row n of the factor
    double acc = 0.0;
#pragma unroll 1
    for (int s = a.NBB - 1; s > j; --s) {
        if (t < DBB) {
            const int g = DBB * s + t; double v = 0.0;
            if (g < n) {   // (bounded: half a second on the constant 100 MHz clock, then the solve is flagged with BCR_STATUS_HANDOFF_TIMEOUT instead of hanging the queue)
                const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
                for (;;) { v = bcr_xload(a.x + g); if ((unsigned long long)__double_as_longlong(v) != BCR_X_SENTINEL) break;
                    __builtin_amdgcn_s_sleep(1);
                    if (__builtin_amdgcn_s_memrealtime() - t0 > BCR_HANDOFF_TICKS) { atomicCAS(a.status, 0, BCR_STATUS_HANDOFF_TIMEOUT); break; } }
            }
            xs[s & 1][t] = v;
        }
        bcr_lds_barrier();             // (xs is double buffered: one barrier per hop)
        if (tile_ok(s)) { double d; DBW_DOT(d, T, xs[s & 1] + 32 * q); acc += d; }
        if (s - 1 > j) DBW_TILE(T, s - 1);
    }
    acc += __shfl_xor(acc, 1, 64); acc += __shfl_xor(acc, 2, 64);
    if (q == 0) us[c] = gc < n ? z - acc : 0.0;
    bcr_lds_barrier();
    double xv; DBW_DOT(xv, INV, us + 32 * q);
    xv += __shfl_xor(xv, 1, 64); xv += __shfl_xor(xv, 2, 64);
    if (q == 0 && gc < n) bcr_xstore(a.x + gc, xv);
#undef DBW_DOT
#undef DBW_TILE
#undef DBW_LOAD
}
// Dinv: ceil(n / 128) slots of 128 x 128 doubles
void launch_dense_bwd_fused(hipStream_t st, const double* S, const double* LiD, double* Dinv, int npad, int n, double* x, int* status) {
    const int NBB = (n + DBB - 1) / DBB; if (NBB <= 0) return;
    static bool attr = false; constexpr int lds = (int)(sizeof(double) * 36 * BTS);
    if (!attr) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&dense_dinv_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, lds); attr = true; }
    hipLaunchKernelGGL(dense_dinv_kernel<0>, dim3((unsigned)NBB), dim3(512), lds, st, S, LiD, Dinv, x, npad, n, 0, (const double*)nullptr);
    DenseBwdArgs a{S, Dinv, x, status, npad, n, NBB};
    hipLaunchKernelGGL(dense_bwd_fused_kernel, dim3((unsigned)NBB), dim3(512), 0, st, a);
}

// tile-sparse reduced system (nlls_tsp.hip): the panels of all pivot tiles of a level; the inverses of all factored diagonal tiles
void launch_tsp_panel(hipStream_t st, double* S, double* W, double* LiD, double* Dfac, const TspPanelJob* jobs, int njobs, int* status, int dch, const double* diag0, double relfloor, unsigned* mask) {
    if (njobs <= 0) return;
    static bool attr = false; constexpr int lds1 = (int)dense_panel_lds<8, 1>(), lds2 = (int)dense_panel_lds<8, 2>();
    if (!attr) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&dense_panel_kernel<8, 1, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds1);
                 (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&dense_panel_kernel<8, 2, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds2);
                 (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&dense_panel_kernel<8, 1, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds1);
                 (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&dense_panel_kernel<8, 2, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds2); attr = true; }
    TspPanelArgs a{S, W, LiD, Dfac, jobs, status, diag0, relfloor, mask};
    if (relfloor > 0.0) {
        if (dch == 2) hipLaunchKernelGGL((dense_panel_kernel<8, 2, true, true>), dim3((unsigned)njobs), dim3(BCR_T), (size_t)lds2, st, a);
        else hipLaunchKernelGGL((dense_panel_kernel<8, 1, true, true>), dim3((unsigned)njobs), dim3(BCR_T), (size_t)lds1, st, a);
    } else if (dch == 2) hipLaunchKernelGGL((dense_panel_kernel<8, 2, true>), dim3((unsigned)njobs), dim3(BCR_T), (size_t)lds2, st, a);
    else hipLaunchKernelGGL((dense_panel_kernel<8, 1, true>), dim3((unsigned)njobs), dim3(BCR_T), (size_t)lds1, st, a);
}
void launch_tsp_dinv(hipStream_t st, const double* LiD, const double* Dfac, double* Dinv, const int32_t* list, int nlist, int nt) {
    if (nlist <= 0) return;
    static bool attr = false; constexpr int lds = (int)(sizeof(double) * 36 * BTS);
    if (!attr) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&dense_dinv_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, lds); attr = true; }
    hipLaunchKernelGGL(dense_dinv_kernel<2>, dim3((unsigned)nlist), dim3(512), lds, st, (const double*)nullptr, LiD, Dinv, (double*)nullptr, DBB * nt, 0, 0, Dfac, list);
}

// ---------------------------------------------------------------------------------------------------
// host
// ---------------------------------------------------------------------------------------------------
bool BcrSolver::supports(int64_t n_band, int bw, int nbd) {
    return bw >= 1 && (bw + 15) / 16 <= BCR_MAXNT && nbd <= 15 && n_band >= 1 && n_band < (int64_t)1 << 30;
}

int BcrSolver::build(int64_t n_band_, int bw_, int nbd_, int H_, std::string* err, int nt) {
    release();
    n_band = (int)n_band_; bw = bw_; nbd = nbd_; H = H_;
    NT = nt > 0 ? nt : std::max(1, (bw + 15) / 16); const int b = 16 * NT; N = (n_band + b - 1) / b;
    const int ND = NT * (NT + 1) / 2, NO = NT * (NT - 1) / 2, RXT = 2 * NT + 1;
    size_t off = 0;
    auto take = [&](size_t doubles) { const size_t o = off; off += (doubles + 31) & ~(size_t)31; return o; };
    geom = BcrGeom{};
    geom.oD = take((size_t)N * ND * 256); geom.oA = take((size_t)N * NT * NT * 256); geom.oBR = take((size_t)N * NT * 256);
    const size_t nslot = (size_t)(N + 1) / 2;                  // blocks a level eliminates at most
    geom.oWx = take(nslot * RXT * NT * 256); geom.oLx = take(nslot * RXT * NT * 256); geom.oMx = take((size_t)N * RXT * NT * 256);
    geom.oLd = take((size_t)N * std::max(NO, 1) * 256); geom.oLi = take((size_t)N * NT * 256);
    geom.oMd = take((size_t)N * std::max(NO, 1) * 256); geom.ocp = take((size_t)(N + 1) * 256); geom.oxb = take(32); geom.odg = take((size_t)N * 16 * NT);
    if (off >= ((size_t)1 << 32)) { if (err) *err = "block cyclic reduction workspace exceeds 32-bit tile offsets"; return NLLS_ERR_UNSUPPORTED; }
    geom.NT = NT; geom.N = N; geom.nbd = nbd; geom.n_band = n_band; geom.bw = bw; geom.H = H;
    std::vector<BcrElim> elims; std::vector<BcrUpd> upds;
    std::vector<int> active(N); for (int k = 0; k < N; ++k) active[k] = k;
    std::vector<int> slot_of(N, 0);                            // position of a block in the launch of the level that eliminates it
    auto wx = [&](int src, int P) { return (uint32_t)(geom.oWx + ((size_t)slot_of[src] * RXT + P) * NT * 256); };
    auto lx = [&](int src, int P) { return (uint32_t)(geom.oLx + ((size_t)slot_of[src] * RXT + P) * NT * 256); };
    // the factor of an eliminated block, pre-multiplied for the backward pass (one wavefront per tile)
    auto premul_jobs = [&](const BcrElim& el) {
        auto one = [&](size_t dst, size_t a, int J) { BcrUpd u{}; u.dst = (uint32_t)dst; u.mode = 3; u.nc = 1; u.a[0] = (uint32_t)a; u.b[0] = (uint32_t)(geom.oLi + ((size_t)el.i * NT + J) * 256); upds.push_back(u); };
        for (int Rg = 0; Rg < RXT; ++Rg) { if (Rg < NT ? el.l < 0 : (Rg < 2 * NT && el.r < 0)) continue;
            for (int J = 0; J < NT; ++J) one(geom.oMx + (((size_t)el.i * RXT + Rg) * NT + J) * 256, (size_t)lx(el.i, Rg) + (size_t)J * 256, J); }
        for (int I = 1; I < NT; ++I) for (int J = 0; J < I; ++J) one(geom.oMd + ((size_t)el.i * NO + I * (I - 1) / 2 + J) * 256, geom.oLd + ((size_t)el.i * NO + I * (I - 1) / 2 + J) * 256, J);
    };
    auto corner_job = [&](int i) { BcrUpd u{}; u.dst = (uint32_t)(geom.ocp + (size_t)(1 + i) * 256); u.mode = 2; u.nc = 1; u.a[0] = wx(i, 2 * NT); u.b[0] = lx(i, 2 * NT); upds.push_back(u); };
    while (active.size() > 1) {
        // the larger independent set of the chain goes: positions 0, 2, 4, ... of an odd-length chain (m -> (m - 1) / 2), else 1, 3, ...
        const size_t m = active.size(), first = (m & 1) ? 0 : 1;
        BcrLevel lv; lv.elim_off = elims.size(); lv.upd_off = upds.size();
        lv.o = active[0]; lv.s = m > 1 ? active[1] - active[0] : 1; lv.m = (int)m; lv.first = (int)first;
        for (size_t idx = first; idx < m; idx += 2) {
            slot_of[active[idx]] = (int)(elims.size() - lv.elim_off);
            elims.push_back(BcrElim{active[idx], idx >= 1 ? active[idx - 1] : -1, idx + 1 < m ? active[idx + 1] : -1, 0});
        }
        for (size_t idx = 1 - first; idx < m; idx += 2) {        // the survivors: both chain neighbours (where they exist) are eliminated now
            const int j = active[idx];
            int src[2], so[2], ns = 0;
            if (idx >= 1) { src[ns] = active[idx - 1]; so[ns] = NT; ++ns; }        // j is the right neighbour of the block eliminated on its left
            if (idx + 1 < m) { src[ns] = active[idx + 1]; so[ns] = 0; ++ns; }      // ... and the left neighbour of the one on its right
            const bool lvl0 = levels.empty();                  // the first level: its RMW jobs can take their old values from the band storage (no conversion launch)
            for (int I = 0; I < NT; ++I) for (int K = 0; K <= I; ++K) {
                BcrUpd u{}; u.dst = (uint32_t)(geom.oD + ((size_t)j * ND + I * (I + 1) / 2 + K) * 256); u.mode = 0; u.nc = ns; u.pad = lvl0 ? (uint32_t)(1 + j) : 0u;
                for (int s = 0; s < ns; ++s) { u.a[s] = wx(src[s], so[s] + I); u.b[s] = lx(src[s], so[s] + K); }
                upds.push_back(u);
            }
            for (int K = 0; K < NT; ++K) {
                BcrUpd u{}; u.dst = (uint32_t)(geom.oBR + ((size_t)j * NT + K) * 256); u.mode = 0; u.nc = ns; u.pad = lvl0 ? (uint32_t)(-(1 + j)) : 0u;
                for (int s = 0; s < ns; ++s) { u.a[s] = wx(src[s], 2 * NT); u.b[s] = lx(src[s], so[s] + K); }
                upds.push_back(u);
            }
        }
        for (size_t e = lv.elim_off; e < elims.size(); ++e) {
            const BcrElim& el = elims[e];
            if (el.l >= 0 && el.r >= 0) for (int P = 0; P < NT; ++P) for (int Q = 0; Q < NT; ++Q) {     // the new coupling (rows: block r, columns: block l)
                BcrUpd u{}; u.dst = (uint32_t)(geom.oA + ((size_t)el.r * NT * NT + P * NT + Q) * 256); u.mode = 1; u.nc = 1;
                u.a[0] = wx(el.i, NT + P); u.b[0] = lx(el.i, Q); upds.push_back(u);
            }
            if (nbd > 0) corner_job(el.i);
            premul_jobs(el);
        }
        lv.nelim = (int)(elims.size() - lv.elim_off); lv.nupd = (int)(upds.size() - lv.upd_off);
        levels.push_back(lv);
        std::vector<int> next; for (size_t idx = 1 - first; idx < m; idx += 2) next.push_back(active[idx]);
        active.swap(next);
    }
    {   // the root block
        BcrLevel lv; lv.elim_off = elims.size(); lv.upd_off = upds.size();
        lv.o = active[0]; lv.s = 1; lv.m = 1; lv.first = 0;
        slot_of[active[0]] = 0;
        elims.push_back(BcrElim{active[0], -1, -1, 0});
        if (nbd > 0) corner_job(active[0]);
        premul_jobs(elims.back());
        lv.nelim = 1; lv.nupd = (int)(upds.size() - lv.upd_off);
        levels.push_back(lv);
    }
    if (hipSuccess != ws.alloc(off) || hipSuccess != d_elim.upload(elims) || hipSuccess != d_upd.upload(upds)) { if (err) *err = "block cyclic reduction workspace alloc"; return NLLS_ERR_HIP; }
    if (hipSuccess != hipMemset(ws.p, 0, off * sizeof(double))) { if (err) *err = "workspace memset"; return NLLS_ERR_HIP; }
    { const char* e = getenv("NLLS_BCR_LEVEL_BACKWARD"); fused_backward = !(e && e[0] == '1'); }      // A/B switch: one backward launch per level, as in round 2
    // (the dispatch order of workgroups is not a contract: the fused pass is taken only while ALL its workgroups are resident at once -- 23 KB of LDS
    //  and 5 wavefronts each, six per CU -- so that nothing waits on a workgroup that has not started)
    if (N > 4 * 256) fused_backward = false;
    geom.ws = ws.p;
    panel_lds = sizeof(double) * ((size_t)(ND + BCR_CH * NT) * BTS + 2 * (size_t)(NT + BCR_CH) * 16 * BP + 64 + 2 * 16 * BP + BTS + 128);
    back_lds = sizeof(double) * ((size_t)RXT * 16 + NT * 64 + 2 * NT * 16 + (size_t)NO * 256 + 256 + 16);
    { const char* e = getenv("NLLS_BCR_CHROWS_SLOTS"); chrows_slots = e ? atoi(e) : 256; }
    launches = 1; for (auto& lv : levels) launches += 1 + (lv.nupd > 0) + (fused_backward ? 0 : 1);
    launches += fused_backward ? 1 : 0;
    {   // matrix-core instructions per solve (2048 flop each): panel kernel per workgroup + update kernel per job
        mfma_issued = 0;
        for (const BcrLevel& lv : levels) for (size_t e = lv.elim_off; e < lv.elim_off + (size_t)lv.nelim; ++e) {
            const BcrElim& el = elims[e];
            const int chr = bcr_level_chrows(NT, lv.nelim, chrows_slots);      // (the X rows per workgroup of this level's launch: fewer rows, more redundant factorisations)
            for (int ch = 0; ch < bcr_nchunks(NT, chr); ++ch) {
                int RX = 0; for (int s2 = 0; s2 < chr; ++s2) { const int Rg = chr * ch + s2; if (Rg <= 2 * NT && (Rg < NT ? el.l >= 0 : (Rg < 2 * NT ? el.r >= 0 : true))) ++RX; }
                if (!RX) continue;
                int64_t m = 0;
                for (int J = 0; J < NT; ++J) {
                    m += 30;                                                   // diagonal tile: 15 pivots x (tile + inverse)
                    if (J + 1 < NT) m += 8;                                    // W_1' and the next diagonal tile's update
                    m += 4 * ((J + 1 < NT ? NT - J - 2 : 0) + RX);             // panel tiles
                    // tile-updates: six helper waves, batches of three jobs -- a batch issues its twelve MFMAs whether or not all three jobs exist
                    // (the absent ones work on a spare tile: no branches), so the count is per BATCH (checked against SQ_INSTS_VALU_MFMA_F64: tools/pmc_mfma.sh)
                    const int mm = NT - 1 - J;
                    if (mm > 0) { const int ntot = mm * (mm + 1) / 2 - 1 + RX * mm; for (int w0 = 0; w0 < 6 && w0 < ntot; ++w0) m += 12 * ((ntot - w0 + 17) / 18); }
                }
                mfma_issued += m;
            }
        }
        for (const BcrUpd& u : upds) mfma_issued += u.mode == 3 ? 4 : 4 * (int64_t)NT * u.nc;
    }
    ready = true;
    return NLLS_OK;
}

template <int NT>
static void bcr_launch_level(const BcrSolver& S, hipStream_t st, const BcrLevel& lv, int* status, double relfloor, double* xr) {
    // X rows per workgroup: as few as still fit ONE round of the chip (every workgroup factors D_i beside its rows; the fewer rows, the less
    // helper work stands beside the pivot chain that sets the pace): 3 when the level is wide, 1 at the narrow levels near the root
    const int chrows = bcr_level_chrows(NT, lv.nelim, S.chrows_slots);
    BcrPanelArgs pa{S.geom, BcrChain{lv.o, lv.s, lv.m, lv.first}, status, relfloor, chrows, S.fused_backward ? xr : nullptr};
    if (relfloor > 0.0) hipLaunchKernelGGL(bcr_panel_kernel<true>, dim3((unsigned)(bcr_nchunks(NT, chrows) * lv.nelim)), dim3(BCR_T), S.panel_lds, st, pa);
    else hipLaunchKernelGGL(bcr_panel_kernel<false>, dim3((unsigned)(bcr_nchunks(NT, chrows) * lv.nelim)), dim3(BCR_T), S.panel_lds, st, pa);
    if (lv.nupd > 0) {
        hipLaunchKernelGGL((bcr_update_kernel<NT>), dim3((unsigned)((lv.nupd + 3) / 4)), dim3(256), 0, st, S.geom.ws, S.d_upd.p + lv.upd_off, lv.nupd);
    }
}
template <int NT>
static void bcr_launch_back(const BcrSolver& S, hipStream_t st, const BcrLevel& lv, double* xr, int root, int* status) {
    BcrBackArgs ba{S.geom, BcrChain{lv.o, lv.s, lv.m, lv.first}, xr, root, status, nullptr};
    hipLaunchKernelGGL((bcr_backward_kernel<NT, false>), dim3((unsigned)lv.nelim), dim3(64 * NT), S.back_lds, st, ba);
}
template <int NT>
static void bcr_launch_back_all(const BcrSolver& S, hipStream_t st, double* xr, int* status) {
    BcrBackArgs ba{S.geom, BcrChain{0, 1, 0, 0}, xr, 0, status, S.d_elim.p};
    hipLaunchKernelGGL((bcr_backward_kernel<NT, true>), dim3((unsigned)S.N), dim3(64 * NT), S.back_lds, st, ba);
}

int BcrSolver::enqueue(hipStream_t st, const double* Sb, double* xr, int* status, double pivot_floor) const {
    const int ND = NT * (NT + 1) / 2, per = ND + NT * NT + NT;
    // The tiles: assembled in place (Sb == nullptr: schur_gather_kernel) or converted from the band storage.  (Rounds 3-4 could skip the conversion for damped solves -- the first
    // level reading the band storage itself, NLLS_BCR_FOLD_CONVERT: measured equal, 312.5 against 313.0 us per trial; out of the library since round 5, last in the tree at 6e015b8.)
    if (Sb) hipLaunchKernelGGL(bcr_convert_kernel, dim3((unsigned)(N * per + 1)), dim3(256), 0, st, geom, Sb);
#define BCR_NT_SWITCH(CALL) switch (NT) { case 1: CALL(1); break; case 2: CALL(2); break; case 3: CALL(3); break; case 4: CALL(4); break; default: CALL(5); break; }
#define BCR_FWD(n) bcr_launch_level<n>(*this, st, lv, status, pivot_floor, xr)
    for (const BcrLevel& lv : levels) BCR_NT_SWITCH(BCR_FWD)
#define BCR_BWD(n) bcr_launch_back<n>(*this, st, levels[li], xr, li + 1 == levels.size() ? 1 : 0, status)
#define BCR_BWD_ALL(n) bcr_launch_back_all<n>(*this, st, xr, status)
    if (fused_backward) { BCR_NT_SWITCH(BCR_BWD_ALL) }
    else for (size_t li = levels.size(); li-- > 0;) BCR_NT_SWITCH(BCR_BWD)
#undef BCR_BWD_ALL
#undef BCR_FWD
#undef BCR_BWD
#undef BCR_NT_SWITCH
    return hipGetLastError() == hipSuccess ? NLLS_OK : NLLS_ERR_HIP;
}

}  // namespace nlls
