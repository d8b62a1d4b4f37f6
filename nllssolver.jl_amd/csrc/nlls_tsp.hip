// nlls_tsp.hip -- the reduced system as a TILE-SPARSE symmetric matrix, factored level by level of a nested-dissection elimination tree (gfx950).
//
// Same contract as every other reduced solver (src/linearsolver.jl:28-32, src/iterators.jl:149-153): x solves S x = s.  The reference's sparse LDL'
// analyses the pattern once with a fill-reducing ordering (src/linearsystem.jl:52,68) and takes any sparsity; of the device solvers, block cyclic
// reduction (nlls_bcr.hip) needs a narrow band and the dense LDL' (nlls_solve.hip) costs n^3 / 3 -- or, restricted to a wide band, still one dependent
// panel step per 128 columns.  A camera graph that is a 2-D grid, a loop, a tree of sub-maps is neither narrow nor dense.
//
// Here the reduced blocks are ordered by nested dissection (nlls_nd.cpp) and packed into TILES of 128 unknowns; S is stored as the lower tiles of the
// filled tile pattern (128 x 128 column-major each) + one right-hand-side strip per tile column.  The elimination tree over the tiles is cut into LEVELS
// (tiles of one level do not touch each other), and each level is three launches over ALL its tiles:
//   panel    (dense_panel_kernel<8, 1, TSP>, nlls_bcr.hip)  every pivot tile of the level: LDL' of the diagonal tile on the matrix cores, the tiles
//            below it (and the right-hand-side strip) as 16-row chunks, one workgroup each: L in place, W = L Delta beside it;
//   update   (tsp_update_kernel)  every tile (i, j) that receives a Schur complement from the level:  C_ij -= sum_k W_ik L_jk'  -- one workgroup per
//            TARGET tile walks all its contributions, so no two workgroups write the same tile and the result is bit-reproducible given S;
//            the strips take  z_j -= sum_k L_jk w_k  in the same launch (the forward substitution rides along, as in the dense solver);
//   backward (tsp_backward_kernel, levels in reverse)  x_k = inv(L_kk)' (z_k - sum_i L_ik' x_i)  with the explicit inverses of the unit-lower
//            diagonal tiles (dense_dinv_kernel<2>: one launch for all tiles).
// The depth of the tree, not the number of tile columns, is the length of the dependent chain.
#include <algorithm>
#include <cstdlib>
#include <map>

#include "nlls_tsp.hpp"

namespace nlls {

typedef double tdouble4_t __attribute__((ext_vector_type(4)));
typedef double tdouble2_t __attribute__((ext_vector_type(2)));

// padding -> identity, reduced right-hand side -> row 0 of the strips (tile order)
__global__ __launch_bounds__(256) void tsp_begin_kernel(double* __restrict__ S, const double* __restrict__ s, const int32_t* __restrict__ ipos, const int64_t* __restrict__ padpos,
                                                        int64_t npadpos, int64_t npos, int64_t strip0, double* __restrict__ acc, double* __restrict__ diag0, const int32_t* __restrict__ tmap, int nt,
                                                        unsigned* __restrict__ mask, int64_t nslots) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < nslots) mask[i] = 0u;
    if (i < npos) { const int32_t src = ipos[i]; S[strip0 + (i >> 7) * TSP_STRIP + 16 * (i & 127)] = src >= 0 ? s[src] : 0.0; acc[i] = 0.0;
        // the original diagonal, in tile order (pivot floor of undamped solves; the padding's 1.0 is written by another thread of this launch: say so here)
        const int k = (int)(i >> 7), r = (int)(i & 127); diag0[i] = src >= 0 ? fabs(S[(size_t)tmap[(size_t)k * nt + k] * TSP_TE + r + (size_t)TSP_TR * r]) : 1.0; }
    else if (i - npos < npadpos) S[padpos[i - npos]] = 1.0;
}
__global__ __launch_bounds__(256) void tsp_scatter_kernel(double* __restrict__ s, const double* __restrict__ xt, const int32_t* __restrict__ tpos, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) s[i] = xt[tpos[i]];
}

// One TARGET per workgroup.  kind 0: a 128 x 128 tile, syrk_update128_kernel's tiling (nlls_solve.hip): eight wavefronts, 64 x 32 each = 4 x 2 accumulator
// tiles of v_mfma_f64_16x16x4_f64, operands through LDS in chunks of 16 columns, double buffered, products formed transposed so that the read-modify-write
// of C walks down columns; the K loop runs over all contributions (128 columns each).  kind 1: a right-hand-side strip, matrix-vector products.
constexpr int TU_KC = 16, TU_LD = 144;
// a right-hand-side strip:  z_j[c] -= sum over contributions, m:  L_jk[c][m] w_k[m]      (w_k: row 0 of the W strip of pivot k)
__device__ __forceinline__ void tsp_rhs_job(double* __restrict__ S, const double* __restrict__ W, const TspUpdJob& jb, const TspCon* __restrict__ cons, double* red) {
    const int t = threadIdx.x, c = t & 127, mq = t >> 7; double acc = 0.0;
    for (int q = 0; q < jb.ncon; ++q) { const TspCon cn = cons[jb.con0 + q];
        const double* __restrict__ L = S + cn.loff + c; const double* __restrict__ w = W + cn.woff;
#pragma unroll 8
        for (int m = mq; m < TSP_TR; m += 4) acc = fma(L[(size_t)TSP_TR * m], w[16 * m], acc); }
    red[t] = acc;
    __syncthreads();
    if (t < 128) S[jb.coff + 16 * t] -= (red[t] + red[t + 128]) + (red[t + 256] + red[t + 384]);
}
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) void tsp_update_kernel(double* __restrict__ S, const double* __restrict__ W, const TspUpdJob* __restrict__ jobs,
                                                                                                    const TspCon* __restrict__ cons, const unsigned* __restrict__ mask) {
    __shared__ double As[2][TU_KC * TU_LD], Bs[2][TU_KC * TU_LD];
    const TspUpdJob jb = jobs[blockIdx.x];
    const int t = threadIdx.x;
    if (jb.kind == 1) { tsp_rhs_job(S, W, jb, cons, As[0]); return; }
    const int w = __builtin_amdgcn_readfirstlane(t >> 6), lane = t & 63, li = lane & 15, lk = lane >> 4;
    // Wavefront w owns the 16 x 16 blocks (row chunk 2 a + (w & 1), column chunk (w >> 1) + 4 b), a < 4, b < 2 -- INTERLEAVED over the tile, not one 64 x 32
    // corner: the chunks of a tile that hold anything are a contiguous range more often than not, and the skipped products should thin every wavefront alike.
    const int rw = w & 1, cw = w >> 1;
    tdouble4_t acc[4][2];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b2 = 0; b2 < 2; ++b2) acc[a][b2] = tdouble4_t{0, 0, 0, 0};
    const int cr = t & 127, kq = t >> 7;                        // copy roles: row cr of both operands, every fourth column of a chunk
    constexpr int NCP = TU_KC / 4, CPT = TSP_TR / TU_KC;        // columns per thread and chunk; chunks per contribution
    double ra[2][NCP], rb[2][NCP];                             // operand chunks are requested two ahead
    auto gload = [&](int chunk, int set) {
        const TspCon cn = cons[jb.con0 + chunk / CPT]; const int col0 = (chunk % CPT) * TU_KC;
        const double* Ga = W + cn.woff + cr + (size_t)TSP_TR * col0; const double* Gb = S + cn.loff + cr + (size_t)TSP_TR * col0;
#pragma unroll
        for (int i = 0; i < NCP; ++i) { ra[set][i] = Ga[(size_t)TSP_TR * (kq + 4 * i)]; rb[set][i] = Gb[(size_t)TSP_TR * (kq + 4 * i)]; }
    };
    auto lstore = [&](int buf, int set) {
#pragma unroll
        for (int i = 0; i < NCP; ++i) { As[buf][(kq + 4 * i) * TU_LD + cr] = ra[set][i]; Bs[buf][(kq + 4 * i) * TU_LD + cr] = rb[set][i]; }
    };
    // the 16 x 16 products of this wavefront that a contribution needs: bit 2 a + b set = row chunk of W_ik AND column chunk of L_jk hold something, and (a tile
    // on the diagonal) the block is not above it  (wave-uniform: scalar loads of the two tiles' chunk masks)
    unsigned shape = 0;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b2 = 0; b2 < 2; ++b2) if (!jb.diag || 2 * a + rw >= cw + 4 * b2) shape |= 1u << (2 * a + b2);
    auto need_of = [&](int chunk) -> unsigned {
        if (!mask) return shape;
        const TspCon cn = cons[jb.con0 + chunk / CPT];
        const unsigned mw = __builtin_amdgcn_readfirstlane(mask[cn.woff >> 14]), ml = __builtin_amdgcn_readfirstlane(mask[cn.loff >> 14]);
        unsigned nd = 0;
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b2 = 0; b2 < 2; ++b2) if (((mw >> (2 * a + rw)) & 1u) && ((ml >> (cw + 4 * b2)) & 1u)) nd |= 1u << (2 * a + b2);
        return nd & shape;
    };
    auto products = [&](int buf, unsigned need) {
        if (need == 0) return;
#pragma unroll
        for (int kk = 0; kk < TU_KC; kk += 4) {
            double av[4], bv[2];
#pragma unroll
            for (int a = 0; a < 4; ++a) av[a] = As[buf][(kk + lk) * TU_LD + 16 * (2 * a + rw) + li];
#pragma unroll
            for (int b2 = 0; b2 < 2; ++b2) bv[b2] = Bs[buf][(kk + lk) * TU_LD + 16 * (cw + 4 * b2) + li];
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b2 = 0; b2 < 2; ++b2) if (need & (1u << (2 * a + b2))) acc[a][b2] = __builtin_amdgcn_mfma_f64_16x16x4f64(bv[b2], av[a], acc[a][b2], 0, 0, 0);
        }
    };
    const int NCH = jb.ncon * CPT;                              // (even: CPT is)
    unsigned touched = 0;
    gload(0, 0); gload(1, 1); lstore(0, 0);
    __syncthreads();
#pragma unroll 1
    for (int ch = 0; ch < NCH; ch += 2) {
        const unsigned need = need_of(ch);                     // (chunks ch and ch + 1 belong to the same contribution)
        touched |= need;
        if (ch + 2 < NCH) gload(ch + 2, 0);
        products(0, need);
        lstore(1, 1);
        __syncthreads();
        if (ch + 3 < NCH) gload(ch + 3, 1);
        products(1, need);
        if (ch + 2 < NCH) lstore(0, 0);
        __syncthreads();
    }
    // C/D layout of the TRANSPOSED product: row = lane & 15, column = (lane >> 4) + 4 r of the 16 x 16 block; blocks nothing was added to are left alone
#pragma unroll
    for (int b2 = 0; b2 < 2; ++b2)
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            if (!(touched & (1u << (2 * a + b2)))) continue;
            double* Cg = S + jb.coff + 16 * (2 * a + rw) + li + (size_t)TSP_TR * (16 * (cw + 4 * b2) + lk);
            if (jb.kind == 2) {                                 // the target's contributions are split over several workgroups
#pragma unroll
                for (int r = 0; r < 4; ++r) atomicAdd(&Cg[(size_t)TSP_TR * 4 * r], -acc[a][b2][r]);
            } else {
                double cold[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) cold[r] = Cg[(size_t)TSP_TR * 4 * r];
#pragma unroll
                for (int r = 0; r < 4; ++r) Cg[(size_t)TSP_TR * 4 * r] = cold[r] - acc[a][b2][r];
            }
        }
}

// The same update with a QUARTER of a target tile per workgroup (64 x 64; wavefront w: the 16 x 32 block at rows 16 (w & 3), columns 32 (w >> 2)): for the levels
// near the root, whose few target tiles would leave most of the chip idle behind 14 us products.  Workgroup b: quarter b & 3 of tile job b >> 2 (the upper right
// quarter of a diagonal tile is not needed); the strip jobs follow the 4 ntile quarters.
__global__ __launch_bounds__(512) void tsp_update_quad_kernel(double* __restrict__ S, const double* __restrict__ W, const TspUpdJob* __restrict__ jobs, const TspCon* __restrict__ cons, int ntile,
                                                              const unsigned* __restrict__ mask) {
    __shared__ double As[2][TU_KC * TU_LD], Bs[2][TU_KC * TU_LD];
    const int t = threadIdx.x;
    if ((int)blockIdx.x >= 4 * ntile) { tsp_rhs_job(S, W, jobs[ntile + ((int)blockIdx.x - 4 * ntile)], cons, As[0]); return; }
    const TspUpdJob jb = jobs[blockIdx.x >> 2];
    const int qi = (blockIdx.x >> 1) & 1, qj = blockIdx.x & 1;
    if (jb.diag && qi == 0 && qj == 1) return;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6), lane = t & 63, li = lane & 15, lk = lane >> 4;
    const int r0w = (w & 3) * 16, c0w = (w >> 2) * 32;
    tdouble4_t acc[2] = {tdouble4_t{0, 0, 0, 0}, tdouble4_t{0, 0, 0, 0}};
    const int cr = t & 63, op = (t >> 6) & 1, kq = t >> 7;      // copy roles: row cr of operand op (0: W tile, 1: L tile), every fourth column of a chunk
    constexpr int NCP = TU_KC / 4, CPT = TSP_TR / TU_KC;
    double rr[2][NCP];
    auto gload = [&](int chunk, int set) {
        const TspCon cn = cons[jb.con0 + chunk / CPT]; const int col0 = (chunk % CPT) * TU_KC;
        const double* G = (op ? S + cn.loff + 64 * qj : W + cn.woff + 64 * qi) + cr + (size_t)TSP_TR * col0;
#pragma unroll
        for (int i = 0; i < NCP; ++i) rr[set][i] = G[(size_t)TSP_TR * (kq + 4 * i)];
    };
    auto lstore = [&](int buf, int set) {
        double* D = op ? Bs[buf] : As[buf];
#pragma unroll
        for (int i = 0; i < NCP; ++i) D[(kq + 4 * i) * TU_LD + cr] = rr[set][i];
    };
    auto need_of = [&](int chunk) -> unsigned {
        if (!mask) return 3u;
        const TspCon cn = cons[jb.con0 + chunk / CPT];
        const unsigned mw = __builtin_amdgcn_readfirstlane(mask[cn.woff >> 14]) >> (4 * qi + (r0w >> 4)), ml = __builtin_amdgcn_readfirstlane(mask[cn.loff >> 14]) >> (4 * qj + (c0w >> 4));
        return (mw & 1u) ? (ml & 3u) : 0u;
    };
    auto products = [&](int buf, unsigned need) {
        if (need == 0) return;
#pragma unroll
        for (int kk = 0; kk < TU_KC; kk += 4) {
            const double av = As[buf][(kk + lk) * TU_LD + r0w + li];
            const double b0 = Bs[buf][(kk + lk) * TU_LD + c0w + li], b1 = Bs[buf][(kk + lk) * TU_LD + c0w + 16 + li];
            if (need & 1u) acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(b0, av, acc[0], 0, 0, 0);
            if (need & 2u) acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(b1, av, acc[1], 0, 0, 0);
        }
    };
    const int NCH = jb.ncon * CPT;
    gload(0, 0); gload(1, 1); lstore(0, 0);
    __syncthreads();
#pragma unroll 1
    for (int ch = 0; ch < NCH; ch += 2) {
        const unsigned need = need_of(ch);
        if (ch + 2 < NCH) gload(ch + 2, 0);
        products(0, need);
        lstore(1, 1);
        __syncthreads();
        if (ch + 3 < NCH) gload(ch + 3, 1);
        products(1, need);
        if (ch + 2 < NCH) lstore(0, 0);
        __syncthreads();
    }
    double* Cg = S + jb.coff + 64 * qi + r0w + (size_t)TSP_TR * (64 * qj + c0w);
#pragma unroll
    for (int b2 = 0; b2 < 2; ++b2) {
        if (jb.kind == 2) {
#pragma unroll
            for (int r = 0; r < 4; ++r) atomicAdd(&Cg[(size_t)li + (size_t)TSP_TR * (16 * b2 + lk + 4 * r)], -acc[b2][r]);
        } else {
            double cold[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) cold[r] = Cg[(size_t)li + (size_t)TSP_TR * (16 * b2 + lk + 4 * r)];
#pragma unroll
            for (int r = 0; r < 4; ++r) Cg[(size_t)li + (size_t)TSP_TR * (16 * b2 + lk + 4 * r)] = cold[r] - acc[b2][r];
        }
    }
}

// A tile below a factored pivot tile in ONE matrix product (levels with many tiles: dense_trsm128_kernel's scheme, nlls_solve.hip): with X = inv(L_kk) of the
// unit-lower pivot tile,  W = S_ik X'  and  L = W / Delta;  L in place, W beside it.  Dfac: the factored pivot tiles (Delta on the diagonal).
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) void tsp_trsm_kernel(double* __restrict__ S, double* __restrict__ W, const double* __restrict__ Dinv, const double* __restrict__ Dfac,
                                                                                                  const TspTrsmJob* __restrict__ jobs, unsigned* __restrict__ mask) {
    __shared__ double As[2][TU_KC * TU_LD], Bs[2][TU_KC * TU_LD];
    __shared__ double rd[TSP_TR];
    const TspTrsmJob jb = jobs[blockIdx.x];
    const int t = threadIdx.x, w = __builtin_amdgcn_readfirstlane(t >> 6), lane = t & 63, li = lane & 15, lk = lane >> 4;
    // wavefront w: the 64 rows r0w .. and the column chunks cw and 7 - cw (16 columns each): X = inv(L_kk) is LOWER triangular, so the columns of output chunk c
    // take K only up to 16 (c + 1) -- chunk c costs c + 1 K-steps, and the pair (cw, 7 - cw) costs nine for every wavefront (sixteen without the skip)
    const int r0w = (w & 1) * 64, cw = w >> 1, cc0 = cw, cc1 = 7 - cw;
    const double* __restrict__ Xinv = Dinv + (size_t)jb.k * TSP_TE;
    if (t < TSP_TR) rd[t] = 1.0 / Dfac[(size_t)jb.k * TSP_TE + (size_t)t * (TSP_TR + 1)];
    tdouble4_t acc[4][2];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b2 = 0; b2 < 2; ++b2) acc[a][b2] = tdouble4_t{0, 0, 0, 0};
    const int cr = t & 127, kq = t >> 7;
    constexpr int NCP = TU_KC / 4, NCH = TSP_TR / TU_KC;
    double ra[2][NCP], rb[2][NCP];
    auto gload = [&](int chunk, int set) {
        const int col0 = chunk * TU_KC;
        const double* Ga = S + jb.xoff + cr + (size_t)TSP_TR * col0;              // S_ik(row, column k)
        const double* Gb = Xinv + cr + (size_t)TSP_TR * col0;                     // X(j = cr, k)
#pragma unroll
        for (int i = 0; i < NCP; ++i) { ra[set][i] = Ga[(size_t)TSP_TR * (kq + 4 * i)]; rb[set][i] = Gb[(size_t)TSP_TR * (kq + 4 * i)]; }
    };
    auto lstore = [&](int buf, int set) {
#pragma unroll
        for (int i = 0; i < NCP; ++i) { As[buf][(kq + 4 * i) * TU_LD + cr] = ra[set][i]; Bs[buf][(kq + 4 * i) * TU_LD + cr] = rb[set][i]; }
    };
    auto products = [&](int buf, int ch) {                     // K chunk ch (16 columns of X): X(j, k) = 0 for k > j
        const bool n0 = cc0 >= ch, n1 = cc1 >= ch;
        if (!n0 && !n1) return;
#pragma unroll
        for (int kk = 0; kk < TU_KC; kk += 4) {
            double av[4], bv[2];
#pragma unroll
            for (int a = 0; a < 4; ++a) av[a] = As[buf][(kk + lk) * TU_LD + r0w + 16 * a + li];
            bv[0] = Bs[buf][(kk + lk) * TU_LD + 16 * cc0 + li]; bv[1] = Bs[buf][(kk + lk) * TU_LD + 16 * cc1 + li];
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                if (n0) acc[a][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(bv[0], av[a], acc[a][0], 0, 0, 0);
                if (n1) acc[a][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(bv[1], av[a], acc[a][1], 0, 0, 0);
            }
        }
    };
    gload(0, 0); gload(1, 1); lstore(0, 0);
    __syncthreads();
    static_assert(NCH % 2 == 0, "unrolled by two");
#pragma unroll 1
    for (int ch = 0; ch < NCH; ch += 2) {
        if (ch + 2 < NCH) gload(ch + 2, 0);
        products(0, ch);
        lstore(1, 1);
        __syncthreads();
        if (ch + 3 < NCH) gload(ch + 3, 1);
        products(1, ch + 1);
        if (ch + 2 < NCH) lstore(0, 0);
        __syncthreads();
    }
    if (mask) {          // the 16-row chunks of the tile that hold anything (see dense_panel_kernel<TSP>)
#pragma unroll
        for (int a = 0; a < 4; ++a) { bool nz = false;
#pragma unroll
            for (int b2 = 0; b2 < 2; ++b2)
#pragma unroll
                for (int r = 0; r < 4; ++r) nz |= acc[a][b2][r] != 0.0;
            if (__any(nz) && lane == 0) atomicOr(&mask[jb.xoff >> 14], 1u << ((r0w >> 4) + a)); }
    }
#pragma unroll
    for (int b2 = 0; b2 < 2; ++b2)
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = r0w + 16 * a + li, j = 16 * (b2 ? cc1 : cc0) + lk + 4 * r; const double wv = acc[a][b2][r];
                W[jb.xoff + i + (size_t)TSP_TR * j] = wv; S[jb.xoff + i + (size_t)TSP_TR * j] = wv * rd[j];
            }
}

// Backward substitution, one launch per level from the root down, PUSH style: the unknowns of a pivot tile i are  x_i = inv(L_ii)' (z_i - acc_i),  where acc_i has
// collected  L_ai' x_a  from every tile a above it; each tile (i, k) of tile ROW i is a workgroup of i's level that forms x_i itself (one matrix-vector product with
// the explicit inverse: cheaper than a hand-over) and adds  L_ik' x_i  to acc_k (128 atomic adds); one more workgroup per pivot tile stores x_i.  Every workgroup
// reads two tiles, whatever the number of tiles in a row or column: a level costs one launch of a few microseconds.  (One workgroup per pivot tile walking its whole
// COLUMN was measured first: 6 - 45 us per level, a chain of dependent 128 KB reads on one CU.)
// Thread t = rows 32 (t & 3) .. of column t >> 2 of a tile (contiguous in memory).
__global__ __launch_bounds__(512) void tsp_backward_kernel(const double* __restrict__ S, const double* __restrict__ Dinv, double* __restrict__ xt, double* __restrict__ acc, const TspBwdJob* __restrict__ jobs,
                                                           int64_t strip0) {
    __shared__ __attribute__((aligned(32))) double us[TSP_TR], xs[TSP_TR];
    const TspBwdJob jb = jobs[blockIdx.x];
    const int t = threadIdx.x, c = t >> 2, q = t & 3;
    double INV[32], T[32];
    { const double* P = Dinv + (size_t)jb.i * TSP_TE + (size_t)TSP_TR * c + 32 * q;
#pragma unroll
      for (int i = 0; i < 32; i += 2) { const tdouble2_t v = *reinterpret_cast<const tdouble2_t*>(P + i); INV[i] = v[0]; INV[i + 1] = v[1]; } }
    if (jb.k >= 0) { const double* P = S + jb.loff + (size_t)TSP_TR * c + 32 * q;
#pragma unroll
      for (int i = 0; i < 32; i += 2) { const tdouble2_t v = *reinterpret_cast<const tdouble2_t*>(P + i); T[i] = v[0]; T[i + 1] = v[1]; } }
    if (t < TSP_TR) us[t] = S[strip0 + (int64_t)jb.i * TSP_STRIP + 16 * t] - acc[(size_t)TSP_TR * jb.i + t];
    __syncthreads();
    {
        const double* uv = us + 32 * q; double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll
        for (int i = 0; i < 32; i += 4) { s0 = fma(INV[i], uv[i], s0); s1 = fma(INV[i + 1], uv[i + 1], s1); s2 = fma(INV[i + 2], uv[i + 2], s2); s3 = fma(INV[i + 3], uv[i + 3], s3); }
        double xv = (s0 + s1) + (s2 + s3);
        xv += __shfl_xor(xv, 1, 64); xv += __shfl_xor(xv, 2, 64);
        if (q == 0) { xs[c] = xv; if (jb.k < 0) xt[(size_t)TSP_TR * jb.i + c] = xv; }
    }
    if (jb.k < 0) return;
    __syncthreads();
    {
        const double* xv = xs + 32 * q; double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll
        for (int i = 0; i < 32; i += 4) { s0 = fma(T[i], xv[i], s0); s1 = fma(T[i + 1], xv[i + 1], s1); s2 = fma(T[i + 2], xv[i + 2], s2); s3 = fma(T[i + 3], xv[i + 3], s3); }
        double y = (s0 + s1) + (s2 + s3);
        y += __shfl_xor(y, 1, 64); y += __shfl_xor(y, 2, 64);
        if (q == 0) atomicAdd(&acc[(size_t)TSP_TR * jb.k + c], y);
    }
}

// ---------------------------------------------------------------------------------------------------
// host
// ---------------------------------------------------------------------------------------------------
int TspSolver::build(const TspSym& sym, const std::vector<int32_t>& node_red_off, const std::vector<int32_t>& dof, int n_red, std::string* err,
                     const std::vector<std::vector<int32_t>>* adj, int nborder) {
    release();
    n = n_red; nt = sym.nt;
    if (nt <= 0 || nt > 4096) { if (err) *err = "tile-sparse solver: tile count out of range"; return NLLS_ERR_UNSUPPORTED; }
    const size_t nnode = sym.tile_of.size();
    std::vector<int32_t> map((size_t)n + (size_t)nt * nt, -1), ipos((size_t)nt * TSP_TR, -1);
    for (size_t v = 0; v < nnode; ++v) for (int d = 0; d < dof[v]; ++d) {
        const int32_t p = sym.tile_of[v] * TSP_TR + sym.row_in_tile[v] + d, r = node_red_off[v] + d;
        if (r < 0 || r >= n || p < 0 || p >= nt * TSP_TR || map[r] >= 0 || ipos[p] >= 0) { if (err) *err = "tile-sparse solver: inconsistent tiling"; return NLLS_ERR_INVALID_ARG; }
        map[r] = p; ipos[p] = r; }
    for (int r = 0; r < n; ++r) if (map[r] < 0) { if (err) *err = "tile-sparse solver: a reduced unknown without a tile"; return NLLS_ERR_INVALID_ARG; }
    int32_t* tmap = map.data() + n;
    nslots = 0;
    // tiles (i, k), i > k, that hold a coupling of S as assembled (before fill): first in the slot order
    std::vector<std::vector<int32_t>> asm_col((size_t)nt);
    if (adj) {
        const size_t ninner = adj->size();
        auto mark = [&](int32_t a, int32_t b) { if (a != b) asm_col[(size_t)std::min(a, b)].push_back(std::max(a, b)); };
        for (size_t v = 0; v < ninner; ++v) for (int32_t u : (*adj)[v]) if ((size_t)u > v) mark(sym.tile_of[v], sym.tile_of[(size_t)u]);
        for (size_t bnode = ninner; bnode < ninner + (size_t)nborder && bnode < nnode; ++bnode) for (int t = 0; t < nt; ++t) mark(sym.tile_of[bnode], t);   // (a border node is coupled to everything)
        for (auto& v : asm_col) { std::sort(v.begin(), v.end()); v.erase(std::unique(v.begin(), v.end()), v.end()); }
    }
    auto is_asm = [&](int32_t i, int k) { return !adj || std::binary_search(asm_col[(size_t)k].begin(), asm_col[(size_t)k].end(), i); };
    for (int k = 0; k < nt; ++k) { tmap[(size_t)k * nt + k] = (int32_t)nslots++; for (int32_t i : sym.cstruct[k]) if (is_asm(i, k)) tmap[(size_t)i * nt + k] = (int32_t)nslots++; }
    nslots_assembled = nslots;
    for (int k = 0; k < nt; ++k) for (int32_t i : sym.cstruct[k]) if (!is_asm(i, k)) tmap[(size_t)i * nt + k] = (int32_t)nslots++;
    if (adj) for (int k = 0; k < nt; ++k) for (int32_t i : asm_col[(size_t)k]) if (tmap[(size_t)i * nt + k] < 0) { if (err) *err = "tile-sparse solver: an assembled tile outside the symbolic pattern"; return NLLS_ERR_INVALID_ARG; }
    if (nslots >= ((int64_t)1 << 31) / 4) { if (err) *err = "tile-sparse solver: too many tiles"; return NLLS_ERR_UNSUPPORTED; }
    auto slot = [&](int i, int k) { return (int64_t)tmap[(size_t)i * nt + k] * TSP_TE; };
    const int64_t strip0 = nslots * TSP_TE;
    std::vector<int64_t> padpos;
    for (int k = 0; k < nt; ++k) for (int r = sym.fill[k]; r < TSP_TR; ++r) padpos.push_back(slot(k, k) + r + (int64_t)TSP_TR * r);
    npad_entries = (int64_t)padpos.size();
    std::vector<TspPanelJob> pj; std::vector<TspUpdJob> uj; std::vector<TspCon> uc; std::vector<TspBwdJob> bj; std::vector<TspTrsmJob> tj;
    std::vector<int32_t> plist, rest;
    levels.assign(sym.nlevels, TspLevel{});
    std::vector<std::vector<int32_t>> by_level(sym.nlevels), rowlist(nt);
    for (int k = 0; k < nt; ++k) { by_level[sym.level[k]].push_back(k); for (int32_t i : sym.cstruct[k]) rowlist[i].push_back(k); }
    products = 0;
    // (A/B switches, read at every build: a context's upload decides, not the process)
    const int force_scheme = [] { const char* e = getenv("NLLS_TSP_SCHEME"); return e ? atoi(e) : 0; }();      // 1 / 2 / 3 for every level
    const int slots = [] { const char* e = getenv("NLLS_TSP_SLOTS"); return e ? atoi(e) : 256; }();             // workgroups of one round of the chip (a panel workgroup fills a CU's LDS)
    quad_max = [] { const char* e = getenv("NLLS_TSP_QUAD_MAX"); return e ? atoi(e) : 160; }();                 // target tiles of a level up to which a workgroup takes a quarter tile
    for (int lv = 0; lv < sym.nlevels; ++lv) {
        TspLevel& L = levels[lv]; L.panel0 = pj.size(); L.upd0 = uj.size(); L.bwd0 = bj.size(); L.trsm0 = tj.size();
        std::map<std::pair<int32_t, int32_t>, std::vector<TspCon>> tgt; std::map<int32_t, std::vector<TspCon>> rhs;
        int64_t below = 0; for (int32_t k : by_level[lv]) below += (int64_t)sym.cstruct[k].size();
        // panel scheme: one launch while its workgroups (one or two 16-row chunks each, the pivot chain repeated in every one) fit one round of the chip
        const int64_t np = (int64_t)by_level[lv].size();
        L.scheme = (np + 8 * below <= slots) ? 1 : ((np + 4 * below <= slots) ? 2 : 3);
        if (force_scheme >= 1 && force_scheme <= 3) L.scheme = force_scheme;
        const int chunk = L.scheme == 2 ? 2 : 1;
        for (int32_t k : by_level[lv]) {
            const auto& cs = sym.cstruct[k];
            pj.push_back(TspPanelJob{slot(k, k), strip0 + (int64_t)k * TSP_STRIP, k, 16, 1, 1});        // the strip's chunk: the lead (exports the factored diagonal tile)
            if (L.scheme == 3) { plist.push_back(k); for (int32_t i : cs) tj.push_back(TspTrsmJob{slot(i, k), k, 0}); }
            else { rest.push_back(k); for (int32_t i : cs) for (int q = 0; q < TSP_TR / 16; q += chunk) pj.push_back(TspPanelJob{slot(k, k), slot(i, k) + 16 * q, k, TSP_TR, chunk, 0}); }
            for (size_t a = 0; a < cs.size(); ++a) { for (size_t b = 0; b <= a; ++b) tgt[{cs[a], cs[b]}].push_back(TspCon{slot(cs[a], k), slot(cs[b], k)});
                rhs[cs[a]].push_back(TspCon{strip0 + (int64_t)k * TSP_STRIP, slot(cs[a], k)}); }
            bj.push_back(TspBwdJob{k, -1, 0});                                                        // stores x_k
            for (int32_t d : rowlist[k]) bj.push_back(TspBwdJob{k, d, slot(k, d)});                   // tile (k, d) of row k: pushes L_kd' x_k to acc_d
        }
        // A target's contributions are walked by ONE workgroup (no atomics, a fixed order) unless that chain would outlast the level's share of the chip: a 128^3 product
        // keeps a CU busy ~14 us, so with P products in the level a workgroup should not hold more than ~P / 256 of them; longer lists are cut and their pieces add
        // atomically (the elimination's flush into S is atomic as well: x is reproducible to rounding, not to the bit)
        int64_t P = 0; for (auto& kv : tgt) P += (int64_t)kv.second.size();
        const int cap_env = [] { const char* e = getenv("NLLS_TSP_CAP"); return e ? atoi(e) : 0; }();
        const int64_t cap = cap_env > 0 ? cap_env : std::max<int64_t>(2, (P + 255) / 256);
        std::vector<std::pair<int64_t, std::pair<int32_t, int32_t>>> order;
        for (auto& kv : tgt) order.push_back({-(int64_t)kv.second.size(), kv.first});
        std::sort(order.begin(), order.end());                 // heaviest targets first (a launch ends with its longest workgroup)
        for (auto& o : order) { auto& cl = tgt[o.second];
            if (tmap[(size_t)o.second.first * nt + o.second.second] < 0) { if (err) *err = "tile-sparse solver: fill outside the symbolic pattern"; return NLLS_ERR_INVALID_ARG; }
            const int64_t nc = (int64_t)cl.size(), pieces = (nc + cap - 1) / cap;
            for (int64_t pc = 0; pc < pieces; ++pc) { const int64_t c0 = pc * nc / pieces, c1 = (pc + 1) * nc / pieces;
                uj.push_back(TspUpdJob{slot(o.second.first, o.second.second), (int32_t)(uc.size() + c0), (int32_t)(c1 - c0), o.second.first == o.second.second ? 1 : 0, pieces > 1 ? 2 : 0}); }
            uc.insert(uc.end(), cl.begin(), cl.end()); products += nc; }
        products += below;                                     // (the tiles below the pivot tiles: one product each with inv(L_kk)')
        L.nupd_tile = (int)(uj.size() - L.upd0);
        for (auto& kv : rhs) { uj.push_back(TspUpdJob{strip0 + (int64_t)kv.first * TSP_STRIP, (int32_t)uc.size(), (int32_t)kv.second.size(), 0, 1}); uc.insert(uc.end(), kv.second.begin(), kv.second.end()); }
        L.npanel = (int)(pj.size() - L.panel0); L.nupd = (int)(uj.size() - L.upd0); L.nbwd = (int)(bj.size() - L.bwd0); L.ntrsm = (int)(tj.size() - L.trsm0); L.npiv = (int)np;
    }
    rest0 = plist.size(); nrest = (int)rest.size(); plist.insert(plist.end(), rest.begin(), rest.end());
    if (uc.empty()) uc.push_back(TspCon{0, 0}); if (uj.empty()) uj.push_back(TspUpdJob{}); if (padpos.empty()) padpos.push_back(0); if (tj.empty()) tj.push_back(TspTrsmJob{});
    oW = 0; oLiD = s_elems(); oDfac = oLiD + (size_t)nt * 8 * 256; oDinv = oDfac + (size_t)nt * TSP_TE; oxt = oDinv + (size_t)nt * TSP_TE; oacc = oxt + (size_t)nt * TSP_TR; odg = oacc + (size_t)nt * TSP_TR; omask = odg + (size_t)nt * TSP_TR;
    chunk_masks = getenv("NLLS_TSP_NO_MASKS") == nullptr;
    if (hipSuccess != d_map.upload(map) || hipSuccess != d_ipos.upload(ipos) || hipSuccess != d_panel.upload(pj) || hipSuccess != d_upd.upload(uj) || hipSuccess != d_con.upload(uc) ||
        hipSuccess != d_bwd.upload(bj) || hipSuccess != d_trsm.upload(tj) || hipSuccess != d_plist.upload(plist) || hipSuccess != d_padpos.upload(padpos) ||
        hipSuccess != ws.alloc(omask + (size_t)(nslots + 1) / 2 + 64)) {
        release(); if (err) *err = "tile-sparse solver: device allocation"; return NLLS_ERR_HIP; }
    launches = 3 + (nrest > 0 ? 1 : 0); for (auto& L : levels) launches += (L.scheme == 3 ? 3 : 1) + (L.nupd > 0 ? 1 : 0) + 1;
    ready = true;
    return NLLS_OK;
}

int TspSolver::enqueue(hipStream_t st, double* S, double* s, int* status, double pivot_floor) const {
    if (!ready) return NLLS_ERR_NOT_READY;
    const int64_t strip0 = nslots * TSP_TE, npos = (int64_t)nt * TSP_TR;
    double* W = ws.p + oW; double* LiD = ws.p + oLiD; double* Dfac = ws.p + oDfac; double* Dinv = ws.p + oDinv; double* xt = ws.p + oxt; double* acc = ws.p + oacc; double* diag0 = ws.p + odg; unsigned* mask = reinterpret_cast<unsigned*>(ws.p + omask); unsigned* umask = chunk_masks ? mask : nullptr;
    hipLaunchKernelGGL(tsp_begin_kernel, dim3((unsigned)((std::max<int64_t>(npos + npad_entries, nslots) + 255) / 256)), dim3(256), 0, st, S, (const double*)s, (const int32_t*)d_ipos.p, (const int64_t*)d_padpos.p, npad_entries, npos, strip0, acc, diag0, (const int32_t*)(d_map.p + n), nt, mask, nslots);
    size_t pl = 0;
    for (const TspLevel& L : levels) {
        launch_tsp_panel(st, S, W, LiD, Dfac, d_panel.p + L.panel0, L.npanel, status, L.scheme == 2 ? 2 : 1, diag0, pivot_floor, umask);
        if (L.scheme == 3) {
            launch_tsp_dinv(st, LiD, Dfac, Dinv, d_plist.p + pl, L.npiv, nt); pl += (size_t)L.npiv;
            if (L.ntrsm > 0) hipLaunchKernelGGL(tsp_trsm_kernel, dim3((unsigned)L.ntrsm), dim3(512), 0, st, S, W, (const double*)Dinv, (const double*)Dfac, (const TspTrsmJob*)(d_trsm.p + L.trsm0), umask);
        }
        if (L.nupd > 0 && L.nupd_tile <= quad_max) hipLaunchKernelGGL(tsp_update_quad_kernel, dim3((unsigned)(4 * L.nupd_tile + (L.nupd - L.nupd_tile))), dim3(512), 0, st, S, (const double*)W, (const TspUpdJob*)(d_upd.p + L.upd0), (const TspCon*)d_con.p, L.nupd_tile, (const unsigned*)umask);
        else if (L.nupd > 0) hipLaunchKernelGGL(tsp_update_kernel, dim3((unsigned)L.nupd), dim3(512), 0, st, S, (const double*)W, (const TspUpdJob*)(d_upd.p + L.upd0), (const TspCon*)d_con.p, (const unsigned*)umask);
    }
    launch_tsp_dinv(st, LiD, Dfac, Dinv, d_plist.p + rest0, nrest, nt);
    for (int lv = (int)levels.size() - 1; lv >= 0; --lv) { const TspLevel& L = levels[lv];
        hipLaunchKernelGGL(tsp_backward_kernel, dim3((unsigned)L.nbwd), dim3(512), 0, st, (const double*)S, (const double*)Dinv, xt, acc, (const TspBwdJob*)(d_bwd.p + L.bwd0), strip0); }
    hipLaunchKernelGGL(tsp_scatter_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, s, (const double*)xt, (const int32_t*)d_map.p, n);
    return hipPeekAtLastError() == hipSuccess ? NLLS_OK : NLLS_ERR_HIP;       // (peek: the caller reports the error text)
}

}  // namespace nlls
