// nlls_chain.hip -- the banded reduced system as a CHAIN of dependent pivots (gfx950): round 1's solvers of the bordered band, kept as the fallback for bands wider
// than block cyclic reduction takes (more than 80 columns: nlls_bcr.hip) and behind NLLS_FLAG_NO_BCR / NLLS_FLAG_NO_TWIST (`bench.py --solver chain`).
// Replaces  solve!(linsystem, options)  src/linearsolver.jl:28-32  for the reduced system, as nlls_bcr.hip does; moved out of nlls_solve.hip in round 5 (no change).
#include <cstdlib>
#include <utility>

#include "nlls_wave.hpp"

namespace nlls {

typedef double double4_t __attribute__((ext_vector_type(4)));
static int herr(nlls_ctx* c, hipError_t e, const char* what) { c->err = std::string(what) + ": " + hipGetErrorString(e); return NLLS_ERR_HIP; }
#define HIPCHK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return herr(c, e_, #expr); } while (0)

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


// the solve of a band in band storage (c->S: [banded part | border rows | rhs row] per column, then the border corner); the solution lands in c->s_ptr()
int enqueue_chain_solve(nlls_ctx* c, int n_band, int bw, int nbd, int H) {
    BandArgs a{}; a.Sb = c->S.p; a.Lb = c->Lwork.p; a.xr = c->s_ptr(); a.n_band = n_band; a.bw = bw; a.nbd = nbd; a.H = H; a.CH = c->band_CH; a.status = c->d_status.p;
    a.PFC = (bw + 1 + a.CH - 1) / a.CH + 1; a.RC = (a.PFC + 1) * a.CH; a.NSC = (bw + 1 + c->band_SEG - 1) / c->band_SEG;
    const int nbr = nbd + 1;
    const size_t lds = sizeof(double) * ((size_t)a.RC * H + 2 * (size_t)(2 * a.NSC * c->band_SEG + 2 * c->band_SEG) + (size_t)(bw + 2) * nbr + (size_t)nbr * nbr + nbr + 8);
    const int NBW = (bw + 15) / 16;                // tile rows below the diagonal tile that a block column reaches
    const size_t blk_lds = sizeof(double) * ((size_t)(NBW + 2) * (NBW + 2) * 272 + 272 + 2 * (size_t)(NBW + 2) * 16 * 17 + 64 + 32 * 17 + 272 + 8);
    if (c->band_blocked && NBW <= 5 && (NBW + 2) * 16 <= 128 && H <= 96 && blk_lds <= 160 * 1024) {
        const int nJb = (n_band + 15) / 16, fsz = blk_fsize(NBW, nbd);
        // twisted (two-sided) factorisation: two workgroups, one from each end of the band, meet at a separator of
        // ws columns, bw <= ws <= 16 NBW, so that the sides do not touch each other
        const int kk = (n_band - bw) / 16, ws = n_band - 16 * kk;
        const bool twisted = c->band_twisted && nbd == 0 && ws >= bw && ws <= 16 * NBW && kk >= 4 * (NBW + 2);
        const int JA = twisted ? (kk + 1) / 2 : nJb, JB = twisted ? kk / 2 : 0, cA = 16 * JA;
        double* corner = c->Lwork.p + (size_t)nJb * fsz + 128;
        double* sepA = corner + 2 * nbr * nbr; double* sepB = sepA + (size_t)(16 * NBW) * (16 * NBW) + 16 * NBW;
        BlkArgs2 bkl{};
        for (int sd = 0; sd < (twisted ? 2 : 1); ++sd) {
            BlkArgs& q = bkl.c[sd]; q.Sb = c->S.p; q.Lb = c->Lwork.p + (size_t)(sd ? JA : 0) * fsz; q.corner_out = corner + sd * nbr * nbr;
            q.sep_out = twisted ? (sd ? sepB : sepA) : nullptr; q.n_band = n_band; q.bw = bw; q.nbd = nbd; q.H = H; q.NBW = NBW; q.rev = sd; q.nJs = sd ? JB : JA; q.timing = 1; q.status = c->d_status.p;
        }
        hipLaunchKernelGGL(band_blocked_factor_kernel, dim3(twisted ? 2 : 1), dim3(BLK_T), blk_lds, c->stream, bkl);
        if (twisted) {
            // the separator: a dense ws x ws system in band layout (bandwidth ws - 1), same kernels, one workgroup
            double* Ssep = sepB + (size_t)(16 * NBW) * (16 * NBW) + 16 * NBW; double* Lsep = Ssep + (size_t)ws * (ws + 1) + 8;
            const int nJs2 = (ws + 15) / 16, NBWs = (ws - 1 + 15) / 16;
            hipLaunchKernelGGL(band_sep_combine_kernel, dim3(8), dim3(256), 0, c->stream, (const double*)c->S.p, (const double*)sepA, (const double*)sepB, cA, ws, 16 * NBW, bw, nbd, H, Ssep);
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
            BwdArgs& q = bw2.c[sd]; q.Lt = c->Lwork.p + (size_t)(sd ? JA : 0) * fsz; q.corner_in = corner; q.xr = c->s_ptr(); q.n_band = n_band; q.nbd = nbd; q.NBW = NBW;
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
    HIPCHK(hipGetLastError());
    return NLLS_OK;
}
}  // namespace nlls
