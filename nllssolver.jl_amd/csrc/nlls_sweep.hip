// nlls_sweep.hip -- the residual+Jacobian sweep and the block-sparse J'J / J'r accumulation (gfx950).
//
// Replaces, per outer iteration:
//   zero!(linsystem); costgradhess!(linsystem, vars, costs)   src/optimize.jl:118,167-170
//     -> per block src/cost.jl:29-52, src/residual.jl:57-111, src/autodiff.jl:81-93
//     -> updatesymlinearsystem!                                src/linearsystem.jl:132-175
//   cost(vars, costs)                                          src/cost.jl:10-13
//   update!(to, from, linsystem)                               src/linearsystem.jl:206-213
//
// Accumulate design ("owner image"): A.data is block-ROW-major (src/BlockSparseMatrix.jl:37-44), so all
// blocks a variable's row owns -- (row, col<row) and the diagonal -- are one contiguous run of A.data,
// and consecutive rows are back to back.  Every (cost, slot) incidence is an *entry* in a list sorted by
// the block row of that slot's variable.  A workgroup takes a tile of consecutive rows, keeps their
// A.data / b segment as an LDS image, lets one lane per entry evaluate the block (dual numbers in
// registers) and add its contributions into the image with LDS atomics, then streams the image to HBM
// with plain coalesced stores: each byte of A.data is written exactly once, no zero! pass, no HBM
// atomics.  Rows with many entries (cameras) get a workgroup each, accumulate the diagonal block + b in
// registers and reduce with wavefront shuffles.  Rows that several lists share, or that are split over
// workgroups, fall back to flushing the (already LDS-reduced) image with HBM atomics.
#include <cstring>
#include <utility>
#include <hip/hip_ext.h>

#include "nlls_wave.hpp"
#include "nlls_mf.hpp"

namespace nlls {

// ================================================================================================
// accumulate: light tiles (many rows per workgroup, one entry per lane)
// ================================================================================================
// workgroup barrier that only waits for this wavefront's LDS / scalar traffic: __syncthreads() would also drain
// vmcnt, i.e. stall on the global loads and stores that are deliberately left in flight across it
NLLS_DEV void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// arguments of one entry list's accumulate pass (light or heavy tiles of it)
struct GhArgs {
    const double* vars; const double* edata; const uint32_t* evoff; const uint32_t* edest; const RowInfo* rows; const Tile* tiles;
    const uint32_t* hvoff; uint32_t own_flags; int compact;      // compact heavy list (EntryList::compact)
    RobustSpec rk; int unique_dest; uint32_t ntiles; double* A; double* b; double* partials;
};
template <int KIND, int SLOT>
__device__ __forceinline__ void gh_light_body(const GhArgs& g, uint32_t tile, double* img_raw) {
    using R = Res<KIND>; using I = ResInfo<KIND>;
    constexpr int DS = I::dof(SLOT);
    constexpr int NSYM = DS * (DS + 1) / 2, NACC = NSYM + DS;   // lower triangle of the diagonal block + b, per accumulator copy
    __shared__ double red[TPB / 64];
    const double* __restrict__ vars = g.vars; const double* __restrict__ edata = g.edata; const uint32_t* __restrict__ evoff = g.evoff;
    const uint32_t* __restrict__ edest = g.edest; const RowInfo* __restrict__ rows = g.rows; const RobustSpec rk = g.rk; const int unique_dest = g.unique_dest;
    double* __restrict__ A = g.A; double* __restrict__ b = g.b; double* __restrict__ partials = g.partials;
    const Tile t = g.tiles[tile];
    // the image is shifted by one double when its A.data segment starts on an odd index, so that 16-byte aligned
    // LDS reads pair up with 16-byte aligned HBM stores at the flush
    const uint32_t odd = (uint32_t)(t.data_off & 1);
    double* img = img_raw + odd;
    const uint32_t imglen = t.data_len + t.b_len;
    double* acc = img + imglen;                                 // [nrows][ACC_COPIES][NACC]
    const uint32_t acclen = t.nrows * ACC_COPIES * NACC;
    // this lane's entry (a light tile holds at most TPB entries) and the row record its first fold item needs: issued
    // before the LDS set-up so that their HBM latency overlaps it
    const uint32_t e = t.e0 + threadIdx.x; const bool ok = e < t.e1; const uint32_t ee = ok ? e : t.e0;
    double d[R::NDATA]; uint32_t vo[R::NDEPS], ds[R::NDEPS];
#pragma unroll
    for (int q = 0; q < R::NDATA; ++q) d[q] = edata[(size_t)ee * R::NDATA + q];
#pragma unroll
    for (int q = 0; q < R::NDEPS; ++q) { vo[q] = evoff[(size_t)ee * R::NDEPS + q]; ds[q] = edest[(size_t)ee * R::NDEPS + q]; }
    const uint32_t nfold = t.nrows * NACC;
    const RowInfo ri0 = rows[t.row0 + (threadIdx.x < nfold ? threadIdx.x / NACC : 0)];
    // exclusive rows whose off-diagonal blocks each have exactly one writer are fully overwritten: no zero fill
    if (!(t.flags & TILE_NOZERO)) for (uint32_t i = threadIdx.x; i < imglen; i += TPB) img[i] = 0.0;
    for (uint32_t i = threadIdx.x; i < acclen; i += TPB) acc[i] = 0.0;
    lds_barrier();
    double mycost = 0;
    if (ok) {
        const uint32_t own = ds[SLOT];
        BlockGH<KIND> B; B.compute(vars, vo, d, rk, (own & OWN_KERNEL_FREE) != 0);
        if (own & OWN_COST_OWNER) mycost = B.cost;
        // diagonal block (lower triangle; mirrored at the flush) and b: LDS atomics into one of ACC_COPIES accumulators of
        // the row, chosen by the entry's rank in its row, so that neighbouring lanes of a row hit different addresses
        double* ar = acc + ((size_t)(own & OWN_ROW_MASK) * ACC_COPIES + ((own >> OWN_COPY_SHIFT) & (ACC_COPIES - 1))) * NACC;
        {
            int q = 0;
#pragma unroll
            for (int j = 0; j < DS; ++j)
#pragma unroll
                for (int i = j; i < DS; ++i) atomicAdd(&ar[q++], h_elem<KIND, SLOT, SLOT>(B, i, j));
#pragma unroll
            for (int i = 0; i < DS; ++i) atomicAdd(&ar[NSYM + i], g_elem<KIND, SLOT>(B, i));
        }
        // off-diagonal blocks this row owns: block(A, row(SLOT), row(T)) += H[SLOT range, T range]  (linearsystem.jl:148-149)
        static_for<R::NDEPS>([&](auto Tc) {
            constexpr int T = decltype(Tc)::value;
            if constexpr (T != SLOT) {
                constexpr int DT = I::dof(T);
                if (ds[T] != DEST_NONE) {
                    if (unique_dest) {
#pragma unroll
                        for (int j = 0; j < DT; ++j)
#pragma unroll
                            for (int i = 0; i < DS; ++i) img[ds[T] + i + DS * j] = h_elem<KIND, SLOT, T>(B, i, j);
                    } else {
#pragma unroll
                        for (int j = 0; j < DT; ++j)
#pragma unroll
                            for (int i = 0; i < DS; ++i) atomicAdd(&img[ds[T] + i + DS * j], h_elem<KIND, SLOT, T>(B, i, j));
                    }
                }
            }
        });
    }
    // deterministic cost sum: fixed DPP tree per wavefront, then four partials in order
    mycost = wave_sum_dpp63(mycost);
    if ((threadIdx.x & 63) == 63) red[threadIdx.x >> 6] = mycost;
    lds_barrier();                     // completes the image, the accumulators and red[]
    if (threadIdx.x == 0) { double tc = 0;
#pragma unroll
        for (int k = 0; k < TPB / 64; ++k) tc += red[k];
        partials[tile] = tc; }
    // fold the accumulator copies into the image: full diagonal block (both triangles, linearsystem.jl:140) and b
    for (uint32_t w = threadIdx.x; w < nfold; w += TPB) {
        const uint32_t r = w / NACC, q = w - r * NACC;
        const double* ar = acc + (size_t)r * ACC_COPIES * NACC + q;
        double v = 0;
#pragma unroll
        for (int k = 0; k < (int)ACC_COPIES; ++k) v += ar[k * NACC];
        const RowInfo ri = w < TPB ? ri0 : rows[t.row0 + r];
        if ((int)q >= NSYM) img[ri.b_off + (q - NSYM)] = v;
        else { int qq = q, j = 0; while (qq >= DS - j) { qq -= DS - j; ++j; } const int i = j + qq;
               img[ri.diag_off + i + DS * j] = v; if (i != j) img[ri.diag_off + j + DS * i] = v; }
    }
    lds_barrier();
    if (t.flags & TILE_PARTIAL) {
        for (uint32_t i = threadIdx.x; i < t.data_len; i += TPB) { double v = img[i]; if (nonzero_bits(v)) atomicAdd(&A[t.data_off + i], v); }
        for (uint32_t i = threadIdx.x; i < t.b_len; i += TPB) { double v = img[t.data_len + i]; if (nonzero_bits(v)) atomicAdd(&b[t.b_off + i], v); }
    } else {
        // 16 bytes per lane: LDS index `odd + 2k` and HBM index `data_off + odd + 2k` are both even
        double* dst = A + t.data_off;
        const uint32_t npair = (t.data_len - odd) >> 1;
        const double2* src2 = reinterpret_cast<const double2*>(img + odd);
        double2* dst2 = reinterpret_cast<double2*>(dst + odd);
        for (uint32_t k = threadIdx.x; k < npair; k += TPB) dst2[k] = src2[k];
        if (threadIdx.x == 0 && odd) dst[0] = img[0];
        if (threadIdx.x == 1 && odd + 2 * npair < t.data_len) dst[t.data_len - 1] = img[t.data_len - 1];
        for (uint32_t i = threadIdx.x; i < t.b_len; i += TPB) b[t.b_off + i] = img[t.data_len + i];
    }
}

template <int KIND, int SLOT>
__global__ __launch_bounds__(TPB) void gh_light_kernel(GhArgs g) {
    extern __shared__ __attribute__((aligned(16))) double dyn_lds[];
    gh_light_body<KIND, SLOT>(g, blockIdx.x, dyn_lds);
}

// ================================================================================================
// accumulate: heavy tiles (one row, or a slice of one, per workgroup)
// ================================================================================================
constexpr int HTPB = 128;  // two wavefronts per heavy tile, each walking every other 64-entry slice of the row through its own pipeline
constexpr int HROWS = TPB / HTPB;   // heavy tiles per workgroup
constexpr int HCHUNK = 14;          // accumulators reduced per pass through the transposed LDS image
NLLS_HD size_t gh_heavy_lds(uint32_t heavy_img) { return (size_t)HROWS * heavy_img + (size_t)HCHUNK * (TPB + 1); }   // doubles
template <int KIND, int SLOT, int DEPTH>   // DEPTH: stages of the register pipeline (3 alone; 2 when fused: measured equal there, and 3 would spill)
__device__ __forceinline__ void gh_heavy_body(const GhArgs& g, uint32_t wg, uint32_t heavy_img, double* dyn) {
    using R = Res<KIND>; using I = ResInfo<KIND>;
    constexpr int DS = I::dof(SLOT);
    constexpr int NTRI = DS * (DS + 1) / 2, NACC = NTRI + DS + 1;   // lower triangle + b + cost
    const double* __restrict__ vars = g.vars; const double* __restrict__ edata = g.edata; const uint32_t* __restrict__ evoff = g.evoff;
    const uint32_t* __restrict__ edest = g.edest; const RobustSpec rk = g.rk;
    double* __restrict__ A = g.A; double* __restrict__ b = g.b;
    const int half = threadIdx.x / HTPB, ht = threadIdx.x % HTPB;
    const uint32_t tile = wg * HROWS + half; const bool live = tile < g.ntiles;
    double* img = dyn + (size_t)half * heavy_img;
    double (*hred)[TPB + 1] = reinterpret_cast<double (*)[TPB + 1]>(dyn + (size_t)HROWS * heavy_img);
    Tile t{}; if (live) t = g.tiles[tile];                     // a workgroup's spare half walks an empty tile (same barriers)
    const bool direct = (t.flags & TILE_DIRECT) != 0;
    RowInfo ri{}; if (live) ri = g.rows[t.row0];              // diag_off: offset of the diagonal block inside the row's segment
    for (uint32_t i = ht; i < t.data_len; i += HTPB) img[i] = 0.0;
    __syncthreads();
    double acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = 0.0;
    // The row's entries are walked 128 at a time (one per lane) through a three-deep software pipeline: while entry i
    // is evaluated, the variable gathers of entry i+1 and the entry record of i+2 are in flight.  All workgroups of the
    // launch are resident at once, so this pipeline -- not occupancy -- is what hides the HBM and L2 latency.
    struct Rec { double d[R::NDATA]; uint32_t vo[R::NDEPS], ds[R::NDEPS]; bool ok; };
    const uint32_t* __restrict__ hvoff = g.hvoff; const bool compact = g.compact != 0; const uint32_t own_flags = g.own_flags;
    const uint32_t own_vo = evoff[(size_t)t.e0 * R::NDEPS + SLOT];     // the row's own variable (the same in every entry of the tile)
    auto load_rec = [&](uint32_t e, Rec& r) {
        r.ok = e < t.e1; const uint32_t ee = r.ok ? e : t.e0;
#pragma unroll
        for (int q = 0; q < R::NDATA; ++q) r.d[q] = edata[(size_t)ee * R::NDATA + q];
        if (compact) {                                                   // 4 bytes per other slot; nothing else varies from entry to entry
            int q2 = 0;
#pragma unroll
            for (int q = 0; q < R::NDEPS; ++q) { if (q == SLOT) { r.vo[q] = own_vo; r.ds[q] = own_flags; } else { r.vo[q] = hvoff[(size_t)ee * (R::NDEPS - 1) + q2++]; r.ds[q] = DEST_NONE; } }
        } else {
#pragma unroll
            for (int q = 0; q < R::NDEPS; ++q) { r.vo[q] = evoff[(size_t)ee * R::NDEPS + q]; r.ds[q] = edest[(size_t)ee * R::NDEPS + q]; }
        }
    };
    using St = double[R::NDEPS][MAXST];
    auto stage = [&](uint32_t e2, const Rec& cur, const St& cst, const Rec& nxt, St& nst, Rec& nn) {
        load_rec(e2, nn);                                        // entry record two stages ahead
        BlockGH<KIND>::template load_skip<SLOT>(vars, nxt.vo, nst);   // gathers one stage ahead (the row's own variable is already there)
        if (cur.ok) {
            BlockGH<KIND> B; B.compute_st(cst, cur.d, rk, (cur.ds[SLOT] & OWN_KERNEL_FREE) != 0);
            if (cur.ds[SLOT] & OWN_COST_OWNER) acc[NACC - 1] += B.cost;
            {
                int q = 0;
#pragma unroll
                for (int j = 0; j < DS; ++j)
#pragma unroll
                    for (int i = j; i < DS; ++i) acc[q++] += h_elem<KIND, SLOT, SLOT>(B, i, j);
#pragma unroll
                for (int i = 0; i < DS; ++i) acc[NTRI + i] += g_elem<KIND, SLOT>(B, i);
            }
            static_for<R::NDEPS>([&](auto Tc) {
                constexpr int T = decltype(Tc)::value;
                if constexpr (T != SLOT) {
                    constexpr int DT = I::dof(T);
                    if (cur.ds[T] != DEST_NONE) {
#pragma unroll
                        for (int j = 0; j < DT; ++j)
#pragma unroll
                            for (int i = 0; i < DS; ++i) {
                                const double v = h_elem<KIND, SLOT, T>(B, i, j);
                                if (direct) atomicAdd(&A[(size_t)cur.ds[T] + i + DS * j], v); else atomicAdd(&img[cur.ds[T] + i + DS * j], v);
                            }
                    }
                }
            });
        }
    };
    if constexpr (DEPTH == 3) {
        if (live) {
            Rec r0, r1, r2; St s0, s1, s2;
            load_rec(t.e0 + ht, r0);
            load_rec(t.e0 + HTPB + ht, r1);
            BlockGH<KIND>::load(vars, r0.vo, s0);
            BlockGH<KIND>::template copy_only<SLOT>(s0, s1); BlockGH<KIND>::template copy_only<SLOT>(s0, s2);   // the row's own variable: once per tile
#pragma unroll 1
            for (uint32_t base = t.e0; base < t.e1; base += 3 * HTPB) {   // roles rotate through the three register sets
                stage(base + 2 * HTPB + ht, r0, s0, r1, s1, r2);
                stage(base + 3 * HTPB + ht, r1, s1, r2, s2, r0);
                stage(base + 4 * HTPB + ht, r2, s2, r0, s0, r1);
            }
        }
    } else {
        // two register sets: entry record and gathers of entry i+1 are requested back to back while entry i is evaluated
        auto stage2 = [&](uint32_t e1, const Rec& cur, const St& cst, Rec& nxt, St& nst) {
            load_rec(e1, nxt);
            BlockGH<KIND>::template load_skip<SLOT>(vars, nxt.vo, nst);
            Rec dummy;                                               // stage() wants a record two ahead: none here
            (void)dummy;
            if (cur.ok) {
                BlockGH<KIND> B; B.compute_st(cst, cur.d, rk, (cur.ds[SLOT] & OWN_KERNEL_FREE) != 0);
                if (cur.ds[SLOT] & OWN_COST_OWNER) acc[NACC - 1] += B.cost;
                int q = 0;
#pragma unroll
                for (int j = 0; j < DS; ++j)
#pragma unroll
                    for (int i = j; i < DS; ++i) acc[q++] += h_elem<KIND, SLOT, SLOT>(B, i, j);
#pragma unroll
                for (int i = 0; i < DS; ++i) acc[NTRI + i] += g_elem<KIND, SLOT>(B, i);
                static_for<R::NDEPS>([&](auto Tc) {
                    constexpr int T = decltype(Tc)::value;
                    if constexpr (T != SLOT) {
                        constexpr int DT = I::dof(T);
                        if (cur.ds[T] != DEST_NONE) {
#pragma unroll
                            for (int j = 0; j < DT; ++j)
#pragma unroll
                                for (int i = 0; i < DS; ++i) {
                                    const double v = h_elem<KIND, SLOT, T>(B, i, j);
                                    if (direct) atomicAdd(&A[(size_t)cur.ds[T] + i + DS * j], v); else atomicAdd(&img[cur.ds[T] + i + DS * j], v);
                                }
                        }
                    }
                });
            }
        };
        if (live) {
            Rec r0, r1; St s0, s1;
            load_rec(t.e0 + ht, r0);
            BlockGH<KIND>::load(vars, r0.vo, s0);
            BlockGH<KIND>::template copy_only<SLOT>(s0, s1);
#pragma unroll 1
            for (uint32_t base = t.e0; base < t.e1; base += 2 * HTPB) {
                stage2(base + HTPB + ht, r0, s0, r1, s1);
                stage2(base + 2 * HTPB + ht, r1, s1, r0, s0);
            }
        }
    }
    // fixed-order reduction of the row's diagonal block, b and cost through a transposed LDS image, HCHUNK accumulators per
    // pass: HSUB neighbouring lanes share one accumulator (each sums HTPB/HSUB lane-partials, row pitch TPB+1 doubles),
    // a fixed xor tree joins them, and the first lane of the group owns the destination
    constexpr int HSUB = 8;
    static_assert(HCHUNK * HSUB <= HTPB && (HTPB % (4 * HSUB)) == 0);
#pragma unroll
    for (int c0 = 0; c0 < NACC; c0 += HCHUNK) {
#pragma unroll
        for (int i = 0; i < HCHUNK; ++i) if (c0 + i < NACC) hred[i][threadIdx.x] = acc[c0 + i];
        __syncthreads();
        const int hv = ht / HSUB, hp = ht % HSUB;
        const bool mine = hv < HCHUNK && c0 + hv < NACC;
        double sum = 0;
        if (mine) {
            const double* hr = &hred[hv][half * HTPB + hp * (HTPB / HSUB)];
            double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
#pragma unroll
            for (int k = 0; k < HTPB / HSUB; k += 4) { s0 += hr[k]; s1 += hr[k + 1]; s2 += hr[k + 2]; s3 += hr[k + 3]; }
            sum = (s0 + s1) + (s2 + s3);
        }
#pragma unroll
        for (int m = 1; m < HSUB; m <<= 1) sum += __shfl_xor(sum, m);     // groups are aligned runs of HSUB lanes
        if (mine && hp == 0 && live) {
            const int v = c0 + hv;
            if (v == NACC - 1) g.partials[tile] = sum;
            else if (v >= NTRI) { const int i = v - NTRI; if (t.flags & TILE_PARTIAL) atomicAdd(&b[t.b_off + i], sum); else b[t.b_off + i] = sum; }
            else {
                int q = v, j = 0; while (q >= DS - j) { q -= DS - j; ++j; } const int i = j + q;   // unpack (i >= j)
                if (direct) { double* dg = A + t.data_off + ri.diag_off; atomicAdd(&dg[i + DS * j], sum); if (i != j) atomicAdd(&dg[j + DS * i], sum); }
                else { img[ri.diag_off + i + DS * j] = sum; if (i != j) img[ri.diag_off + j + DS * i] = sum; }   // only registers feed the diagonal block
            }
        }
        __syncthreads();
    }
    if (direct || !live) return;
    if (t.flags & TILE_PARTIAL) {
        for (uint32_t i = ht; i < t.data_len; i += HTPB) { double v = img[i]; if (nonzero_bits(v)) atomicAdd(&A[t.data_off + i], v); }
    } else {
        double* dst = A + t.data_off;
        for (uint32_t i = ht; i < t.data_len; i += HTPB) dst[i] = img[i];
    }
}
template <int KIND, int SLOT>
__global__ __launch_bounds__(TPB) __attribute__((amdgpu_waves_per_eu(2, 2))) void gh_heavy_kernel(GhArgs g, uint32_t heavy_img) {
    extern __shared__ __attribute__((aligned(16))) double dyn_lds[];
    gh_heavy_body<KIND, SLOT, 3>(g, blockIdx.x, heavy_img, dyn_lds);
}
// the heavy tiles of a list with the finishing workgroup of a matrix-free LM trial in front (nlls_mf.hpp): the look-ahead sweep of the trial point's reduced rows carries the
// trial's last reduction -- first in the grid, the host is waiting for it -- instead of standing behind a launch of its own
template <int KIND, int SLOT>
__global__ __launch_bounds__(TPB) __attribute__((amdgpu_waves_per_eu(2, 2))) void gh_heavy_fin_kernel(GhArgs g, uint32_t heavy_img, MfFin fin) {
    extern __shared__ __attribute__((aligned(16))) double dyn_lds[];
    if (blockIdx.x == 0) { __shared__ double red[6][4]; mf_finish_body(fin, red); return; }
    gh_heavy_body<KIND, SLOT, 3>(g, blockIdx.x - 1, heavy_img, dyn_lds);
}
// One launch for a cost group whose entry lists split cleanly into one list of light tiles (bundle adjustment: the point
// rows) and one of heavy tiles (the camera rows): the heavy workgroups come first in the grid and are compute / latency
// bound with almost no HBM traffic, the light ones are bound by the A.data stream -- side by side they overlap instead
// of running back to back.
template <int KIND, int LSLOT, int HSLOT>
__global__ __launch_bounds__(TPB) __attribute__((amdgpu_waves_per_eu(3, 3))) void gh_fused_kernel(GhArgs gl, GhArgs gh, uint32_t heavy_img, uint32_t nhw, uint32_t nhw_pad, unsigned long long* prof) {
    extern __shared__ __attribute__((aligned(16))) double dyn_lds[];
    // in-situ profile (nlls_profile_sweep; prof == nullptr otherwise): first workgroup start / last workgroup end on the 100 MHz
    // constant clock -- the launch's execution span as a kernel trace reports it, taken inside the caller's own loop
    if (prof && threadIdx.x == 0) prof[blockIdx.x] = (unsigned long long)wall_clock64();     // (one slot per workgroup: no contention)
    // the heavy workgroups come first: their lifetime is the longest.  (Spreading them through the grid, or alternating
    // them with light ones at the front, measured slower: 114 and 57 us against 49.)
    // Heavy workgroup w runs on XCD w % 8 (workgroups are dealt round-robin to the eight XCDs, each with its own L2): give every XCD a
    // CONTIGUOUS range of heavy rows -- neighbouring camera rows gather the same points, so the point coordinates then cross from HBM
    // into about one L2 instead of into all eight (2.4 MB x 8 at config 4)
    // (nhw_pad = 8 * ceil(nhw / 8) workgroups stand for the nhw heavy ones: the few beyond the last row have nothing to do)
    if (blockIdx.x < nhw_pad) { const uint32_t hw = (blockIdx.x & 7) * (nhw_pad >> 3) + (blockIdx.x >> 3);
        if (hw < nhw) gh_heavy_body<KIND, HSLOT, 2>(gh, hw, heavy_img, dyn_lds); }
    else gh_light_body<KIND, LSLOT>(gl, blockIdx.x - nhw_pad, dyn_lds);
    if (prof && threadIdx.x == 0) prof[gridDim.x + blockIdx.x] = (unsigned long long)wall_clock64();
}

// ================================================================================================
// accumulate, folded (round 5): every block is evaluated ONCE, by the workgroup of its light row -- and nothing is staged
// ================================================================================================
// One lane per entry of the light list, one tile of consecutive light rows per workgroup, as gh_light_body.  What differs:
//  * the block's share of its HEAVY rows (costgradhess! of the same block, src/residual.jl:91-107: diagonal block, gradient, and the off-diagonal blocks every
//    entry of such a row shares -- BASELINE config 5: camera rows with their (camera, kernel) block, the kernel variable's own row) is summed per heavy row and
//    tile in LDS accumulators ("slots", FoldTile) and leaves the tile as one record per slot in the slab; gh_fold_gather_kernel sums every heavy row's records
//    in a fixed order.  Replaces the heavy passes (gh_heavy_body: one more evaluation of the block per role, 244 registers for the second-order duals).
//  * NO LDS image of the rows' A.data segment: an off-diagonal block with ONE writer (point x camera) goes from the lane's registers straight to HBM -- 16-byte
//    stores, neighbouring lanes neighbouring blocks, the partial lines meet in the L2 -- which costs what staging + flushing it cost (7 against 7.7 us at config 5)
//    and frees 40 of the 56 KB of LDS per workgroup: four workgroups per CU instead of two.  The row's diagonal block, gradient and the blocks its entries SHARE
//    ((point, kernel)) are summed over the row's run of neighbouring lanes in registers (wave_run_sum: same-address LDS atomics serialise lane by lane), added by
//    the run's last lane to one of two accumulators of the row (a row spans at most two wavefronts' lanes: no two lanes of an instruction meet) and stored from there.
// Same A.data, same b, no HBM atomics, every byte written once, bit-reproducible.
constexpr uint32_t FOLD_ROW_COPIES = 2;
struct GhFoldArgs { GhArgs g; const uint32_t* fslot; const FoldTile* ftiles; const uint32_t* rowx; double* slab; FoldHeavy fh[FOLD_MAX_HEAVY]; uint32_t unique_mask, shared_mask; };
template <int KIND, int SLOT>
__global__ __launch_bounds__(TPB) __attribute__((amdgpu_waves_per_eu(4, 4))) void gh_fold_kernel(GhFoldArgs a) {
    using R = Res<KIND>; using I = ResInfo<KIND>;
    extern __shared__ __attribute__((aligned(16))) double dyn_lds[];
    constexpr int DS = I::dof(SLOT);
    constexpr int NSYM = DS * (DS + 1) / 2, NACC = NSYM + DS;
    constexpr int NH = R::NDEPS - 1;
    __shared__ double red[TPB / 64];
    const GhArgs& g = a.g;
    const double* __restrict__ vars = g.vars; const double* __restrict__ edata = g.edata; const uint32_t* __restrict__ evoff = g.evoff;
    const uint32_t* __restrict__ edest = g.edest; const RowInfo* __restrict__ rows = g.rows; const RobustSpec rk = g.rk;
    double* __restrict__ A = g.A; double* __restrict__ b = g.b; double* __restrict__ partials = g.partials;
    const uint32_t tile = blockIdx.x;
    const Tile t = g.tiles[tile]; const FoldTile ft = a.ftiles[tile];
    // doubles per row accumulator: [lower triangle of the diagonal block | gradient | the shared off-diagonal blocks (SLOT, T), T ascending]
    uint32_t nw = NACC;
    static_for<R::NDEPS>([&](auto Tc) { constexpr int T = decltype(Tc)::value; if constexpr (T != SLOT) if (a.shared_mask >> T & 1) nw += DS * I::dof(T); });
    double* acc = dyn_lds;                                      // [nrows][FOLD_ROW_COPIES][nw]
    const uint32_t acclen = t.nrows * FOLD_ROW_COPIES * nw;
    double* facc = acc + acclen;                                // per heavy slot h: [slot][copy][cw]
    uint32_t hoff[NH + 1], roff[NH + 1];                        // first accumulator / first record double of heavy slot h
    hoff[0] = 0; roff[0] = 0;
#pragma unroll
    for (int h = 0; h < NH; ++h) { hoff[h + 1] = hoff[h] + (uint32_t)ft.ns[h] * (uint32_t)a.fh[h].copies * (uint32_t)a.fh[h].cw; roff[h + 1] = roff[h] + (uint32_t)ft.ns[h] * (uint32_t)a.fh[h].cw; }
    const uint32_t e = t.e0 + threadIdx.x; const bool ok = e < t.e1; const uint32_t ee = ok ? e : t.e0;
    double d[R::NDATA]; uint32_t vo[R::NDEPS], ds[R::NDEPS];
#pragma unroll
    for (int q = 0; q < R::NDATA; ++q) d[q] = edata[(size_t)ee * R::NDATA + q];
#pragma unroll
    for (int q = 0; q < R::NDEPS; ++q) { vo[q] = evoff[(size_t)ee * R::NDEPS + q]; ds[q] = edest[(size_t)ee * R::NDEPS + q]; }
    const uint32_t fs = a.fslot[ee];
    for (uint32_t i = threadIdx.x; i < acclen + hoff[NH]; i += TPB) acc[i] = 0.0;
    lds_barrier();
    double mycost = 0;
    {
        // every lane evaluates (a lane without an entry re-evaluates the tile's first one and adds nothing): the segmented sums below are wavefront-wide
        const uint32_t own = ds[SLOT];
        BlockGH<KIND> B; B.compute(vars, vo, d, rk, (own & OWN_KERNEL_FREE) != 0);
        if (ok) mycost = B.cost;                                // the light list holds every block of the group exactly once
        const WaveRuns runs = wave_runs(ok ? 1u + (own & OWN_ROW_MASK) : 0xFFFF0000u + threadIdx.x);
        const bool add = ok && runs.tail;
        double* ar = acc + ((size_t)(own & OWN_ROW_MASK) * FOLD_ROW_COPIES + ((threadIdx.x >> 6) & (FOLD_ROW_COPIES - 1))) * nw;
        {
            int q = 0;
#pragma unroll
            for (int j = 0; j < DS; ++j)
#pragma unroll
                for (int i = j; i < DS; ++i) { const double v = wave_run_sum(h_elem<KIND, SLOT, SLOT>(B, i, j), runs); if (add) atomicAdd(&ar[q], v); ++q; }
#pragma unroll
            for (int i = 0; i < DS; ++i) { const double v = wave_run_sum(g_elem<KIND, SLOT>(B, i), runs); if (add) atomicAdd(&ar[NSYM + i], v); }
        }
        double* ax = ar + NACC;
        static_for<R::NDEPS>([&](auto Tc) {
            constexpr int T = decltype(Tc)::value;
            if constexpr (T != SLOT) {
                constexpr int DT = I::dof(T);
                if (a.unique_mask >> T & 1) {
                    // ONE writer: registers -> HBM, 16 bytes per store (the block is 8-byte aligned: packed pairs)
                    if (ok && ds[T] != DEST_NONE) {
                        struct __attribute__((packed, aligned(8))) D2 { double x, y; };
                        double* dst = A + t.data_off + ds[T];
#pragma unroll
                        for (int q = 0; q + 1 < DS * DT; q += 2) { D2 v; v.x = h_elem<KIND, SLOT, T>(B, q % DS, q / DS); v.y = h_elem<KIND, SLOT, T>(B, (q + 1) % DS, (q + 1) / DS); *reinterpret_cast<D2*>(dst + q) = v; }
                        if constexpr ((DS * DT) & 1) dst[DS * DT - 1] = h_elem<KIND, SLOT, T>(B, DS - 1, DT - 1);
                    }
                } else if (a.shared_mask >> T & 1) {
                    // shared by the row's entries (the same block for all of them: build_fold): summed over the row's run like the diagonal block
                    const bool on = add && ds[T] != DEST_NONE;
#pragma unroll
                    for (int j = 0; j < DT; ++j)
#pragma unroll
                        for (int i = 0; i < DS; ++i) { const double v = wave_run_sum(h_elem<KIND, SLOT, T>(B, i, j), runs); if (on) atomicAdd(&ax[i + DS * j], v); }
                    ax += DS * DT;
                }
                // ... and the block's share of T's heavy row: [lower triangle of (T, T) | gradient | blocks (T, U) the row's entries share]
                constexpr int h = T - (T > SLOT ? 1 : 0);
                const uint32_t w = (fs >> (10 * h)) & 0x3FFu, sl = w & 0x3Fu, cp = w >> 6;
                if (ok && sl != FOLD_SLOT_NONE) {
                    double* fa = facc + hoff[h] + (size_t)(sl * (uint32_t)a.fh[h].copies + cp) * (uint32_t)a.fh[h].cw;
                    int q = 0;
#pragma unroll
                    for (int j = 0; j < DT; ++j)
#pragma unroll
                        for (int i = j; i < DT; ++i) atomicAdd(&fa[q++], h_elem<KIND, T, T>(B, i, j));
#pragma unroll
                    for (int i = 0; i < DT; ++i) atomicAdd(&fa[q++], g_elem<KIND, T>(B, i));
                    const int xm = a.fh[h].xmask; double* fx = fa + q;
                    static_for<R::NDEPS>([&](auto Uc) {
                        constexpr int U = decltype(Uc)::value;
                        if constexpr (U != T) {
                            constexpr int DU = I::dof(U);
                            if (xm >> U & 1) {
#pragma unroll
                                for (int j = 0; j < DU; ++j)
#pragma unroll
                                    for (int i = 0; i < DT; ++i) atomicAdd(&fx[i + DT * j], h_elem<KIND, T, U>(B, i, j));
                                fx += DT * DU;
                            }
                        }
                    });
                }
            }
        });
    }
    mycost = wave_sum_dpp63(mycost);
    if ((threadIdx.x & 63) == 63) red[threadIdx.x >> 6] = mycost;
    lds_barrier();
    if (threadIdx.x == 0) { double tc = 0;
#pragma unroll
        for (int k = 0; k < TPB / 64; ++k) tc += red[k];
        partials[tile] = tc; }
    // the tile's records: per slot the sum of its accumulator copies, in slab order [h][slot][component]
    {
        double* __restrict__ rec = a.slab + ft.slab_off;
        for (uint32_t w = threadIdx.x; w < roff[NH]; w += TPB) {
            uint32_t cw = (uint32_t)a.fh[0].cw, cps = (uint32_t)a.fh[0].copies, r0 = 0, h0 = 0;      // (selects, not indexed arrays: those would live in scratch memory)
#pragma unroll
            for (int k = 1; k < NH; ++k) if (w >= roff[k]) { cw = (uint32_t)a.fh[k].cw; cps = (uint32_t)a.fh[k].copies; r0 = roff[k]; h0 = hoff[k]; }
            const uint32_t r = w - r0, sl = r / cw, q = r - sl * cw;
            const double* fa = facc + h0 + (size_t)sl * cps * cw + q;
            double v = 0; for (uint32_t k = 0; k < cps; ++k) v += fa[k * cw];
            rec[w] = v;
        }
    }
    // the light rows themselves: accumulator copies summed, the diagonal block mirrored (src/linearsystem.jl:140), straight to A.data / b
    for (uint32_t w = threadIdx.x; w < t.nrows * nw; w += TPB) {
        const uint32_t r = w / nw, q = w - r * nw;
        const double* ar = acc + (size_t)r * FOLD_ROW_COPIES * nw + q;
        double v = 0;
#pragma unroll
        for (int k = 0; k < (int)FOLD_ROW_COPIES; ++k) v += ar[k * nw];
        const RowInfo ri = rows[t.row0 + r];
        double* row = A + t.data_off;
        if ((int)q < NSYM) { int qq = q, j = 0; while (qq >= DS - j) { qq -= DS - j; ++j; } const int i = j + qq;
            row[ri.diag_off + i + DS * j] = v; if (i != j) row[ri.diag_off + j + DS * i] = v; }
        else if ((int)q < NACC) b[t.b_off + (ri.b_off - t.data_len) + (q - NSYM)] = v;
        else { uint32_t rr = q - NACC;
            static_for<R::NDEPS>([&](auto Tc) { constexpr int T = decltype(Tc)::value;
                if constexpr (T != SLOT) if (a.shared_mask >> T & 1) { constexpr uint32_t sz = DS * I::dof(T);
                    if (rr < sz) { const uint32_t xo = a.rowx[(size_t)(t.row0 + r) * 4 + T]; if (xo != DEST_NONE) row[xo + rr] = v; }
                    rr -= sz; } }); }       // (rr wraps below zero behind its block: every later test fails)
    }
}
// One workgroup per heavy row: sums the row's records (FoldRow::cbeg .. cend, tile order) component by component -- GTPB / cwp groups of threads walk
// every (GTPB / cwp)-th record, a fixed tree joins the groups -- and writes the row: the diagonal block mirrored (src/linearsystem.jl:140), its part of b,
// the shared off-diagonal blocks.  The row's whole segment of A.data is written here, once.
constexpr int GTPB = 1024, GCONS = 4096;
__global__ __launch_bounds__(GTPB) void gh_fold_gather_kernel(const FoldRow* __restrict__ frows, const uint32_t* __restrict__ cons, const double* __restrict__ slab,
                                                              GhFoldArgs a, double* __restrict__ A, double* __restrict__ b) {
    __shared__ uint32_t scon[GCONS];
    __shared__ double part[GTPB];
    const FoldRow fr = frows[blockIdx.x]; const FoldHeavy F = a.fh[fr.h];
    const uint32_t cw = (uint32_t)F.cw; uint32_t cwp = 1; while (cwp < cw) cwp <<= 1;
    const uint32_t ng = GTPB / cwp, gi = threadIdx.x / cwp, q = threadIdx.x % cwp;
    double sum = 0;
    for (uint32_t c0 = fr.cbeg; c0 < fr.cend; c0 += GCONS) {
        const uint32_t nc = min((uint32_t)GCONS, fr.cend - c0);
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < nc; i += GTPB) scon[i] = cons[c0 + i];
        __syncthreads();
        if (q < cw) {
            // eight independent loads in flight per thread (a record past the end reads the first one and counts zero): the long rows -- the adaptive kernel's: one
            // record per tile -- are a few round trips instead of one per record
            for (uint32_t k = gi; k < nc; k += 8 * ng) {
                double v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) { const uint32_t kk = k + u * ng; v[u] = slab[scon[kk < nc ? kk : 0] + q]; }
#pragma unroll
                for (int u = 0; u < 8; ++u) sum += (k + u * ng < nc) ? v[u] : 0.0;
            }
        }
    }
    part[threadIdx.x] = sum;
    __syncthreads();
    if (threadIdx.x >= cw) return;
    double v = 0; for (uint32_t k = 0; k < ng; ++k) v += part[k * cwp + threadIdx.x];
    const int ds = F.ds, nsym = F.nsym; const int qq = (int)threadIdx.x;
    double* row = A + fr.data_off;
    if (qq < nsym) { int r = qq, j = 0; while (r >= ds - j) { r -= ds - j; ++j; } const int i = j + r;
        row[fr.diag_off + i + ds * j] = v; if (i != j) row[fr.diag_off + j + ds * i] = v; }
    else if (qq < nsym + ds) b[fr.b_off + (qq - nsym)] = v;
    else { int r = qq - nsym - ds;
        for (int t = 0; t < 4; ++t) if (F.xmask >> t & 1) { const int sz = ds * F.xdof[t]; if (r < sz) { row[fr.xoff[t] + r] = v; break; } r -= sz; } }
}

// The small dense system (nlls_ctx::tiny_dense): A and b = the sum of the sweep workgroups' images in launch order, the lower triangle mirrored
// (symmetrifyfull, BlockDenseMatrix.jl:24-34), and the cost partials' sum -- what two zero fills, the atomics' flush, the mirror launch and the reduction did.
// (The gather as the tail of the last accumulate launch -- the workgroup that draws the last ticket sums the images -- was measured and is SLOWER than this launch:
//  32.5k against 37.7k LM iterations/s at BASELINE config 2; the device-scope fences it needs write the L2 back.  Same for the trial's cost sweep and its reduction.)
__global__ __launch_bounds__(TPB) void dense_tiny_gather_kernel(const double* __restrict__ slab, int nimg, int n, double* __restrict__ A, double* __restrict__ b,
                                                                const double* __restrict__ partials, int64_t npart, double* __restrict__ scalars) {
    __shared__ double red[TPB / 64];
    const int e = blockIdx.x * TPB + threadIdx.x;
    if (e < n * n + n) dense_tiny_gather_elem(slab, nimg, n, A, b, e);
    if (blockIdx.x == 0 && scalars) reduce_partials_body(partials, npart, scalars, red);
}
// ================================================================================================
// accumulate: dense linear system (MultiVariateLSdense, src/linearsystem.jl:73-87; BlockDenseMatrix.jl)
// ================================================================================================
template <int KIND>
__global__ __launch_bounds__(TPB) void gh_dense_kernel(const double* __restrict__ vars, const double* __restrict__ edata,
                                                       const uint32_t* __restrict__ evoff, const uint32_t* __restrict__ ebrow,
                                                       int64_t n, RobustSpec rk, int ndof, int use_lds,
                                                       double* __restrict__ A, double* __restrict__ b, double* __restrict__ partials, double* __restrict__ slab, DenseFin fin) {
    using R = Res<KIND>; using I = ResInfo<KIND>;
    extern __shared__ __attribute__((aligned(16))) double img[];
    __shared__ double red[TPB / 64];
    // (fin: workgroup 0 ends the LM trial in front of this look-ahead sweep -- first in the grid, the host is waiting for it)
    int bid = (int)blockIdx.x, nwg = (int)gridDim.x;
    if (fin.cpart) { if (bid == 0) { dense_fin_body(fin, red); return; } --bid; --nwg; }
    const int imglen = use_lds ? ndof * ndof + ndof : 0;
    for (int i = threadIdx.x; i < imglen; i += TPB) img[i] = 0.0;
    __syncthreads();
    double* HA = use_lds ? img : A;
    double* Hb = use_lds ? img + ndof * ndof : b;
    double mycost = 0;
    // (the loop is uniform per wavefront -- lanes past the end carry zeros -- so that the sums below may run across the whole wave)
    for (int64_t base = (int64_t)bid * TPB; base < n; base += (int64_t)nwg * TPB) {
        const int64_t e = base + threadIdx.x; const bool valid = e < n;
        const uint64_t vmask = __ballot(valid);
        if (vmask == 0) continue;
        double d[R::NDATA]; uint32_t vo[R::NDEPS], br[R::NDEPS];
#pragma unroll
        for (int q = 0; q < R::NDEPS; ++q) br[q] = DEST_NONE;
        BlockGH<KIND> B;
        if (valid) {
#pragma unroll
            for (int q = 0; q < R::NDATA; ++q) d[q] = edata[(size_t)e * R::NDATA + q];
#pragma unroll
            for (int q = 0; q < R::NDEPS; ++q) { vo[q] = evoff[(size_t)e * R::NDEPS + q]; br[q] = ebrow[(size_t)e * R::NDEPS + q]; }
            bool kfree = false; if constexpr (R::ADAPT) kfree = br[0] != DEST_NONE;
            B.compute(vars, vo, d, rk, kfree);
            mycost += B.cost;
        }
        // Every block of the wavefront on the SAME variables (a curve fit: all of them, always): 64 lanes' atomics on one LDS address serialise at the read-modify-write
        // latency -- sum across the wave on the VALU instead (DPP, the total in lane 63) and add once.
        bool uni = use_lds != 0; uint32_t ubr[R::NDEPS];
        { const int src = __ffsll((unsigned long long)vmask) - 1;
#pragma unroll
          for (int q = 0; q < R::NDEPS; ++q) { ubr[q] = (uint32_t)__shfl((int)br[q], src, 64); uni = uni && __all(!valid || br[q] == ubr[q]); } }
        if (uni) {
            const bool l63 = (threadIdx.x & 63) == 63;
            static_for<R::NDEPS>([&](auto Sc) {
                constexpr int S = decltype(Sc)::value; constexpr int DS = I::dof(S);
                if (ubr[S] != DEST_NONE) {
#pragma unroll
                    for (int i = 0; i < DS; ++i) { const double v = wave_sum_dpp63(valid ? g_elem<KIND, S>(B, i) : 0.0); if (l63) atomicAdd(&Hb[ubr[S] + i], v); }
#pragma unroll
                    for (int j = 0; j < DS; ++j)
#pragma unroll
                        for (int i = 0; i < DS; ++i) { const double v = wave_sum_dpp63(valid ? h_elem<KIND, S, S>(B, i, j) : 0.0); if (l63) atomicAdd(&HA[(ubr[S] + i) + (size_t)ndof * (ubr[S] + j)], v); }
                    static_for<S>([&](auto Tc) {
                        constexpr int T = decltype(Tc)::value; constexpr int DT = I::dof(T);
                        if (ubr[T] != DEST_NONE) {
#pragma unroll
                            for (int j = 0; j < DT; ++j)
#pragma unroll
                                for (int i = 0; i < DS; ++i) {
                                    const double v = wave_sum_dpp63(valid ? h_elem<KIND, S, T>(B, i, j) : 0.0);
                                    if (l63) { if (ubr[S] >= ubr[T]) atomicAdd(&HA[(ubr[S] + i) + (size_t)ndof * (ubr[T] + j)], v);
                                               else atomicAdd(&HA[(ubr[T] + j) + (size_t)ndof * (ubr[S] + i)], v); }
                                }
                        }
                    });
                }
            });
            continue;
        }
        if (!valid) continue;
        static_for<R::NDEPS>([&](auto Sc) {
            constexpr int S = decltype(Sc)::value; constexpr int DS = I::dof(S);
            if (br[S] != DEST_NONE) {
#pragma unroll
                for (int i = 0; i < DS; ++i) atomicAdd(&Hb[br[S] + i], g_elem<KIND, S>(B, i));
#pragma unroll
                for (int j = 0; j < DS; ++j)
#pragma unroll
                    for (int i = 0; i < DS; ++i) atomicAdd(&HA[(br[S] + i) + (size_t)ndof * (br[S] + j)], h_elem<KIND, S, S>(B, i, j));
                static_for<S>([&](auto Tc) {   // j < i in slot order; stored in the block-lower triangle (linearsystem.jl:148-152)
                    constexpr int T = decltype(Tc)::value; constexpr int DT = I::dof(T);
                    if (br[T] != DEST_NONE) {
#pragma unroll
                        for (int j = 0; j < DT; ++j)
#pragma unroll
                            for (int i = 0; i < DS; ++i) {
                                const double v = h_elem<KIND, S, T>(B, i, j);
                                if (br[S] >= br[T]) atomicAdd(&HA[(br[S] + i) + (size_t)ndof * (br[T] + j)], v);
                                else atomicAdd(&HA[(br[T] + j) + (size_t)ndof * (br[S] + i)], v);
                            }
                    }
                });
            }
        });
    }
    double tc = block_sum(mycost, red);
    if (threadIdx.x == 0) partials[bid] = tc;
    __syncthreads();
    if (slab) {   // the small dense system: this workgroup's image as it is; dense_tiny_gather_kernel sums the images in launch order
        for (int i = threadIdx.x; i < imglen; i += TPB) slab[(size_t)bid * imglen + i] = img[i];
    } else if (use_lds) {
        for (int i = threadIdx.x; i < ndof * ndof; i += TPB) { double v = img[i]; if (nonzero_bits(v)) atomicAdd(&A[i], v); }
        for (int i = threadIdx.x; i < ndof; i += TPB) { double v = img[ndof * ndof + i]; if (nonzero_bits(v)) atomicAdd(&b[i], v); }
    }
}
// gethessian(::MultiVariateLSdense) = symmetrifyfull: mirror the lower triangle  BlockDenseMatrix.jl:24-34
__global__ void symmetrize_dense_kernel(double* __restrict__ A, int n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)n * n) return;
    const int r = (int)(i % n), c = (int)(i / n);
    if (r > c) A[c + (size_t)n * r] = A[i];
}

__global__ void zero_ranges_kernel(double* __restrict__ A, const int64_t* __restrict__ off, const uint32_t* __restrict__ len,
                                   double* __restrict__ b, const uint32_t* __restrict__ boff, const uint32_t* __restrict__ blen) {
    const int64_t o = off[blockIdx.x]; const uint32_t l = len[blockIdx.x];
    for (uint32_t i = threadIdx.x; i < l; i += blockDim.x) A[o + i] = 0.0;
    const uint32_t bo = boff[blockIdx.x], bl = blen[blockIdx.x];
    for (uint32_t i = threadIdx.x; i < bl; i += blockDim.x) b[bo + i] = 0.0;
}

// stage-0 reduce buffer of the sharded sweep: [cost | reduced rows of A.data | reduced part of b]
__global__ void pack_reduce0_kernel(const double* __restrict__ A, const double* __restrict__ b, const int64_t* __restrict__ off, const uint32_t* __restrict__ len,
                                    const uint32_t* __restrict__ dst, const uint32_t* __restrict__ which, const double* __restrict__ scalars, double* __restrict__ buf) {
    const double* src = (which[blockIdx.x] ? b : A) + off[blockIdx.x]; double* d = buf + dst[blockIdx.x];
    for (uint32_t i = threadIdx.x; i < len[blockIdx.x]; i += blockDim.x) d[i] = src[i];
    if (blockIdx.x == 0 && threadIdx.x == 0) buf[0] = scalars[0];
}
__global__ void unpack_reduce0_kernel(double* __restrict__ A, double* __restrict__ b, const int64_t* __restrict__ off, const uint32_t* __restrict__ len,
                                      const uint32_t* __restrict__ dst, const uint32_t* __restrict__ which, double* __restrict__ scalars, const double* __restrict__ buf) {
    double* d = (which[blockIdx.x] ? b : A) + off[blockIdx.x]; const double* src = buf + dst[blockIdx.x];
    for (uint32_t i = threadIdx.x; i < len[blockIdx.x]; i += blockDim.x) d[i] = src[i];
    if (scalars && blockIdx.x == 0 && threadIdx.x == 0) scalars[0] = buf[0];
}

// ================================================================================================
// host-side enqueue
// ================================================================================================
static int herr(nlls_ctx* c, hipError_t e, const char* what) { c->err = std::string(what) + ": " + hipGetErrorString(e); return NLLS_ERR_HIP; }
#define HIPCHK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return herr(c, e_, #expr); } while (0)

template <int KIND>
static GhArgs gh_args(nlls_ctx* c, const Group& G, const EntryList& E, const double* vars, bool heavy, double* partials) {
    GhArgs g{}; g.vars = vars; g.edata = E.data.p; g.evoff = E.voff.p; g.edest = E.dest.p; g.rows = E.rows.p; g.tiles = heavy ? E.heavy.p : E.light.p;
    g.hvoff = E.hvoff.p; g.own_flags = E.own_flags; g.compact = (heavy && E.compact) ? 1 : 0;
    g.rk = G.rk; g.unique_dest = E.unique_dest ? 1 : 0; g.ntiles = (uint32_t)(heavy ? E.nheavy : E.nlight); g.A = c->A.p; g.b = c->b.p; g.partials = partials;
    return g;
}
template <int KIND, int SLOT>
static void launch_gh_slot(nlls_ctx* c, const Group& G, const double* vars, int64_t& pbase) {
    if constexpr (SLOT < Res<KIND>::NDEPS) {
        const EntryList& E = G.lists[SLOT];
        if (E.nlight > 0) {
            hipLaunchKernelGGL((gh_light_kernel<KIND, SLOT>), dim3((unsigned)E.nlight), dim3(TPB), (E.light_lds + 2) * sizeof(double), c->stream,
                               gh_args<KIND>(c, G, E, vars, false, c->partials.p + pbase));
            pbase += E.nlight;
        }
        if (E.nheavy > 0 && c->mf_fin_pending && c->nzero == 0) {      // (a matrix-free trial's finishing workgroup rides in front: nlls_lm_trial deferred it)
            c->mf_fin_pending = false;
            hipLaunchKernelGGL((gh_heavy_fin_kernel<KIND, SLOT>), dim3(1 + (unsigned)((E.nheavy + HROWS - 1) / HROWS)), dim3(TPB), gh_heavy_lds(E.heavy_lds) * sizeof(double), c->stream,
                               gh_args<KIND>(c, G, E, vars, true, c->partials.p + pbase), E.heavy_lds, mf_fin_args(c));
            pbase += E.nheavy;
        } else if (E.nheavy > 0) {
            hipLaunchKernelGGL((gh_heavy_kernel<KIND, SLOT>), dim3((unsigned)((E.nheavy + HROWS - 1) / HROWS)), dim3(TPB), gh_heavy_lds(E.heavy_lds) * sizeof(double), c->stream,
                               gh_args<KIND>(c, G, E, vars, true, c->partials.p + pbase), E.heavy_lds);
            pbase += E.nheavy;
        }
    }
}
// two-slot kinds whose lists split into {light only, heavy only}: one fused launch
template <int KIND>
static bool launch_gh_fused(nlls_ctx* c, const Group& G, const double* vars, int64_t& pbase) {
    if constexpr (Res<KIND>::NDEPS == 2) {
        const EntryList& E0 = G.lists[0]; const EntryList& E1 = G.lists[1];
        int ls = -1;
        if (E0.nlight > 0 && E0.nheavy == 0 && E1.nheavy > 0 && E1.nlight == 0) ls = 0;
        else if (E1.nlight > 0 && E1.nheavy == 0 && E0.nheavy > 0 && E0.nlight == 0) ls = 1;
        if (ls < 0) return false;
        const EntryList& EL = ls == 0 ? E0 : E1; const EntryList& EH = ls == 0 ? E1 : E0;
        const size_t lds = std::max<size_t>(EL.light_lds + 2, gh_heavy_lds(EH.heavy_lds)) * sizeof(double);
        if (lds > ((size_t)EL.light_lds + 2) * sizeof(double) + 4096) return false;    // the heavy role must not cost the light one occupancy
        const unsigned nhw = (unsigned)((EH.nheavy + HROWS - 1) / HROWS), nhw_pad = 8 * ((nhw + 7) / 8);
        const GhArgs gl = gh_args<KIND>(c, G, EL, vars, false, c->partials.p + pbase), gh = gh_args<KIND>(c, G, EH, vars, true, c->partials.p + pbase + EL.nlight);
        unsigned long long* prof = nullptr;
        if (c->prof_sweep && c->prof_clk.p) {                   // (profiling only: two 8-byte fills in front of the launch)
            const unsigned nwg = nhw_pad + (unsigned)EL.nlight;
            if (nwg <= PROF_MAXWG) { const size_t slot = (size_t)(c->prof_kcount % PROF_SLOTS); prof = c->prof_clk.p + slot * 2 * PROF_MAXWG; c->prof_nwg[slot] = nwg; ++c->prof_kcount; }
        }
        // (profiling: the launch carries its own start / stop events -- the dispatch's begin / end timestamps, what a kernel trace reports for it)
        hipEvent_t e0 = c->prof_e0, e1 = c->prof_e1; c->prof_e0 = c->prof_e1 = nullptr; if (e0) c->prof_taken = true;
        if (ls == 0) hipExtLaunchKernelGGL((gh_fused_kernel<KIND, 0, 1>), dim3(nhw_pad + (unsigned)EL.nlight), dim3(TPB), lds, c->stream, e0, e1, 0, gl, gh, EH.heavy_lds, nhw, nhw_pad, prof);
        else         hipExtLaunchKernelGGL((gh_fused_kernel<KIND, 1, 0>), dim3(nhw_pad + (unsigned)EL.nlight), dim3(TPB), lds, c->stream, e0, e1, 0, gl, gh, EH.heavy_lds, nhw, nhw_pad, prof);
        pbase += EL.nlight + EH.nheavy;
        return true;
    }
    return false;
}
// the folded sweep of a group (Group::fold, built at upload): the light tiles' launch, then the heavy rows' gather
template <int KIND, int LS>
static void launch_gh_fold_as(nlls_ctx* c, const Group& G, const double* vars, int64_t& pbase) {
    if constexpr (LS < Res<KIND>::NDEPS && Res<KIND>::NDEPS >= 2 && Res<KIND>::NDEPS <= FOLD_MAX_HEAVY + 1) {
        const EntryList& EL = G.lists[LS];
        GhFoldArgs a{}; a.g = gh_args<KIND>(c, G, EL, vars, false, c->partials.p + pbase); a.fslot = EL.fslot.p; a.ftiles = EL.ftiles.p; a.rowx = EL.frowx.p; a.slab = G.fslab.p;
        a.unique_mask = G.fold_unique; a.shared_mask = G.fold_shared;
        for (int h = 0; h < FOLD_MAX_HEAVY; ++h) a.fh[h] = G.fh[h];
        hipEvent_t e0 = c->prof_e0, e1 = c->prof_e1; c->prof_e0 = c->prof_e1 = nullptr; if (e0) c->prof_taken = true;     // (profiling: begin of the first dispatch .. end of the second)
        hipExtLaunchKernelGGL((gh_fold_kernel<KIND, LS>), dim3((unsigned)EL.nlight), dim3(TPB), ((size_t)G.fold_lds + 2) * sizeof(double), c->stream, e0, G.nfrows > 0 ? (hipEvent_t) nullptr : e1, 0, a);
        if (G.nfrows > 0) hipExtLaunchKernelGGL(gh_fold_gather_kernel, dim3((unsigned)G.nfrows), dim3(GTPB), 0, c->stream, (hipEvent_t) nullptr, e1, 0, G.frows.p, G.fcons.p, G.fslab.p, a, c->A.p, c->b.p);
        pbase += EL.nlight;
    }
}
template <int KIND>
static bool launch_gh_fold(nlls_ctx* c, const Group& G, const double* vars, int64_t& pbase) {
    if (!G.fold) return false;
    switch (G.fold_ls) {
    case 0: launch_gh_fold_as<KIND, 0>(c, G, vars, pbase); break;
    case 1: launch_gh_fold_as<KIND, 1>(c, G, vars, pbase); break;
    case 2: launch_gh_fold_as<KIND, 2>(c, G, vars, pbase); break;
    case 3: launch_gh_fold_as<KIND, 3>(c, G, vars, pbase); break;
    default: return false;
    }
    return true;
}
// the reduced slot's pass alone (the gradient sweep of the matrix-free LM trial, nlls_ctx::grad_level 1: the eliminated rows of A.data are never formed)
template <int KIND>
static void launch_gh_reduced(nlls_ctx* c, const Group& G, const double* vars, int64_t& pbase) {
    if constexpr (Res<KIND>::NDEPS == 2) { if (c->mf_ps == 0) launch_gh_slot<KIND, 1>(c, G, vars, pbase); else launch_gh_slot<KIND, 0>(c, G, vars, pbase); }
}
template <int KIND>
static int launch_gh(nlls_ctx* c, const Group& G, const double* vars, int64_t& pbase, int mode = 0) {
    if (mode == 1) { launch_gh_reduced<KIND>(c, G, vars, pbase); return NLLS_OK; }
    if (c->info.is_sparse) {
        // (three-slot kinds whose lists do not qualify for the folded sweep take one launch per role: rounds 2-4's one-launch form of that, gh_fused3_kernel -- every block
        //  evaluated once per ROLE, 244 registers -- went with the fold: 104 against 55 us at BASELINE config 5; last in the tree at commit 6e015b8)
        if (!launch_gh_fold<KIND>(c, G, vars, pbase) && !launch_gh_fused<KIND>(c, G, vars, pbase)) {
            [&]<int... S>(std::integer_sequence<int, S...>) { (launch_gh_slot<KIND, S>(c, G, vars, pbase), ...); }(std::make_integer_sequence<int, Res<KIND>::NDEPS>{});   // (one pass per slot: up to MAX_SLOTS)
        }
    } else if (G.dense.n > 0) {
        const int ndof = (int)c->info.ndof; const int use_lds = ndof <= 64;
        int grid = (int)std::min<int64_t>((G.dense.n + TPB - 1) / TPB, c->tiny_dense ? TINY_DENSE_MAX_WGS : 1024);
        double* slab = nullptr; DenseFin fin{};
        if (c->tiny_dense) { slab = c->dense_slab.p + (size_t)c->dense_slab_used * (size_t)(ndof * ndof + ndof); c->dense_slab_used += grid; }
        if (c->dense_fin_pending) { fin = c->dense_fin; c->dense_fin_pending = false; }
        hipLaunchKernelGGL(gh_dense_kernel<KIND>, dim3(grid + (fin.cpart ? 1 : 0)), dim3(TPB), use_lds ? (size_t)(ndof * ndof + ndof) * sizeof(double) : 0, c->stream,
                           vars, G.dense.data.p, G.dense.voff.p, G.dense.brow.p, G.dense.n, G.rk, ndof, use_lds, c->A.p, c->b.p, c->partials.p + pbase, slab, fin);
        pbase += grid;
    }
    return enqueue_fixedcost(c, G, vars, pbase);
}

int enqueue_sweep_gradhess(nlls_ctx* c, bool want_cost, int which, int mode) {
    if (mode == 1 && (!c->mf_ok || want_cost)) mode = 0;
    c->grad_level = mode == 1 ? 1 : 2; if (mode == 1) c->mf_reduced_sweeps++; else c->full_sweeps++;
    c->tE_valid = false; c->step_cached = false;  // A and b change: what the last solve kept of them is stale
    c->grad_phys = c->vars_slot[which];           // the variable set (physical slot) A and b are the linearisation of
    const double* vars = vars_ptr(c, which); int64_t pbase = 0;
    c->dense_slab_used = 0;
    if (c->tiny_dense) {
        // (nothing to zero: the gathering launch writes every element of A and b)
    } else if (!c->info.is_sparse) {
        HIPCHK(hipMemsetAsync(c->A.p, 0, sizeof(double) * std::max<int64_t>(c->info.nnz_data, 1), c->stream));
        HIPCHK(hipMemsetAsync(c->b.p, 0, sizeof(double) * std::max<int64_t>(c->info.ndof, 1), c->stream));
    } else if (c->nzero > 0 && c->heavy_rows_zeroed) {
        c->heavy_rows_zeroed = false;          // (the finishing launch of the trial in front of this look-ahead sweep has done it)
    } else if (c->nzero > 0) {
        hipLaunchKernelGGL(zero_ranges_kernel, dim3((unsigned)c->nzero), dim3(64), 0, c->stream, c->A.p, c->d_zero_off.p, c->d_zero_len.p, c->b.p, c->d_zero_b_off.p, c->d_zero_b_len.p);
    }
    const bool prof = c->prof_sweep && !c->prof_ev.empty() && mode == 0;      // (the in-situ profile is of the full accumulate launch)
    const size_t pslot = prof ? (size_t)(c->prof_count % (int64_t)(c->prof_ev.size() / 2)) : 0;
    // profiling: a problem of ONE cost group whose sweep is one fused (or folded) launch hands the event pair to that launch (hipExtLaunchKernelGGL: the
    // dispatch's own begin / end timestamps); anything else is bracketed by recorded events, which also hold the dispatch latency in front
    c->prof_e0 = c->prof_e1 = nullptr; c->prof_taken = false;
    if (prof && c->groups.size() == 1 && c->info.is_sparse && mode == 0) { c->prof_e0 = c->prof_ev[2 * pslot]; c->prof_e1 = c->prof_ev[2 * pslot + 1]; }
    else if (prof) (void)hipEventRecord(c->prof_ev[2 * pslot], c->stream);
    for (const Group& G : c->groups) {
        if (is_dyn_kind(G.res_kind)) { enqueue_dyn_gradhess(c, G, vars, pbase); enqueue_fixedcost(c, G, vars, pbase); continue; }   // (into the dense system, or the variable's diagonal block of a block-sparse one)
        switch (G.res_kind) {
#define X(K) case K: launch_gh<K>(c, G, vars, pbase, mode); break;
            NLLS_FOR_EACH_RES(X)
#undef X
        }
    }
    if (prof) {
        if (c->prof_e0) { (void)hipEventRecord(c->prof_e0, c->stream); c->prof_e0 = nullptr; c->prof_taken = false; c->prof_e1 = nullptr; }   // (no launch took the pair: bracket what ran -- late, but both events exist)
        if (!c->prof_taken) (void)hipEventRecord(c->prof_ev[2 * pslot + 1], c->stream);
        ++c->prof_count; }
    if (c->tiny_dense) {
        const int n = (int)c->info.ndof;
        if (c->dense_slab_used > c->dense_slab_wgs) { c->err = "dense slab overrun"; return NLLS_ERR_HIP; }
        // (summing the images in the next trial's own launch instead -- one launch fewer -- was measured: 39.9k against 46.5k LM iterations/s at BASELINE config 2; the host's
        //  turn-around between two trials hides behind this launch)
        hipLaunchKernelGGL(dense_tiny_gather_kernel, dim3((unsigned)((n * n + n + TPB - 1) / TPB)), dim3(TPB), 0, c->stream, c->dense_slab.p, (int)c->dense_slab_used, n, c->A.p, c->b.p,
                           c->partials.p, pbase, want_cost ? c->scalars.p : (double*)nullptr);
        HIPCHK(hipGetLastError());
        return NLLS_OK;
    }
    if (!c->info.is_sparse && c->info.ndof > 0) {
        const int64_t n2 = c->info.ndof * c->info.ndof;
        hipLaunchKernelGGL(symmetrize_dense_kernel, dim3((unsigned)((n2 + 255) / 256)), dim3(256), 0, c->stream, c->A.p, (int)c->info.ndof);
    }
    if (want_cost) return enqueue_reduce_partials(c, pbase);
    HIPCHK(hipGetLastError());
    return NLLS_OK;
}

int enqueue_pack_reduce0(nlls_ctx* c) {
    if (c->nred_ranges > 0)
        hipLaunchKernelGGL(pack_reduce0_kernel, dim3((unsigned)c->nred_ranges), dim3(64), 0, c->stream, c->A.p, c->b.p, c->d_red_off.p, c->d_red_len.p,
                           c->d_red_dst.p, c->d_red_which.p, c->scalars.p, c->redbuf.p);
    HIPCHK(hipGetLastError());
    return NLLS_OK;
}
int enqueue_unpack_reduce0(nlls_ctx* c, bool with_cost) {
    if (c->nred_ranges > 0)
        hipLaunchKernelGGL(unpack_reduce0_kernel, dim3((unsigned)c->nred_ranges), dim3(64), 0, c->stream, c->A.p, c->b.p, c->d_red_off.p, c->d_red_len.p,
                           c->d_red_dst.p, c->d_red_which.p, with_cost ? c->scalars.p : (double*)nullptr, c->redbuf.p);
    HIPCHK(hipGetLastError());
    return NLLS_OK;
}
}  // namespace nlls
