// nlls_mf.hip -- the matrix-free Levenberg-Marquardt trial (gfx950): Schur elimination and back-substitution that evaluate the cost blocks themselves.
//
// The reference accumulates every block into the BlockSparseMatrix (updatesymlinearsystem!, src/linearsystem.jl:132-175) and then factors the full matrix
// (src/linearsolver.jl:28-32).  With the eliminated blocks' rows materialised an LM iteration of BASELINE config 4 wrote 151 MB of point rows (accumulate sweep), read
// them back (elimination) and read the E blocks a third time (back-substitution) -- for 34 MB of inputs.  Here the point rows are never formed:
//   mf_elim_kernel     one workgroup per supernode (run of eliminated blocks with identical neighbour columns).  A wavefront takes a BATCH of members, one lane per cost
//                      block: residual + Jacobian (the arithmetic of computerescostgradhess, src/residual.jl:57-111, through BlockGH), the block's E part (3 x 6 at bundle
//                      adjustment) into the wavefront's LDS slab, C_v and b_v summed over the member's lanes, (C_v + lambda I)^-1 by the member's first lanes, then per member
//                      the rank-DV update of the supernode's share of S on the matrix cores, operands read from the slab in the layout v_mfma_f64_16x16x4_f64 wants.
//                      Leaves: [S | s] (atomics, as the materialised kernel), (C_v + lambda I)^-1 and b_v per member.
//   mf_backsub_kernel  the same evaluation, E_v x_R per member by a sum over its lanes, x_v = -(C_v + lambda I)^-1 (b_v + E_v x_R), the retraction, and the member rows' share
//                      of the step's quadratic form x'Hx (src/iterators.jl:163) -- nothing of A.data is read.
// The reduced rows (camera diagonal blocks, their part of b) stay materialised: the gradient sweep between two iterations is the reduced slot's pass alone
// (enqueue_sweep_gradhess, mode 1), and the reduced-reduced blocks enter S as before (schur_prep_roles).
// Built with the flags of nlls_sweep.hip (structural zeros of the dual numbers fold away) AND those of nlls_solve.hip (matrix-core accumulators in VGPRs).
#include <algorithm>
#include <utility>

#include "nlls_wave.hpp"
#include "nlls_slayout.hpp"

namespace nlls {

typedef double double4_t __attribute__((ext_vector_type(4)));
constexpr int MF_NW = 4;            // wavefronts per workgroup of the back-substitution
constexpr int MF_ENW = 2;           // wavefronts per supernode of the elimination, each taking every other batch of members (four: two workgroups per CU, and a workgroup's atomic flush -- its slot
                                    // held until the memory side has taken 1891 atomics -- left the CU half idle: 132 us; two: four workgroups per CU, every large supernode of BASELINE config 4 resident at once)
constexpr int MF_BMAX = 8;          // members per batch at most (one lane per cost block: 64 / blocks per member, capped)
constexpr int MF_TRMAX = 5;         // tile rows of [E | b]: nd + 1 <= 80
constexpr int MF_SLOTS = 40;        // members one wavefront handles at most (128 members per supernode)
// one supernode of the matrix-free trial, in launch order (nlls_ctx::d_mf_desc): everything a workgroup needs to start on it comes with one uniform load
// (struct MfDesc: nlls_ctx.hpp -- v0, nmem, nd, rc_off, eb0, obs0, B = members per batch)

// a wavefront's own LDS traffic: writes of some lanes, then reads by others.  The LDS pipe serves one wavefront's instructions in order; the compiler must not
// move them across this point, and the counter wait covers the returned data
NLLS_DEV void wave_lds_sync() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier(); }
NLLS_DEV double mf_rcp(double d) { const double r = __builtin_amdgcn_rcp(d); const double e = fma(-d, r, 1.0); return fma(r, fma(e, e, e), r); }   // v_rcp_f64 (2^-24) + one cubic step: 1.1e-16

struct MfArgs {
    const double* vars; const double* odata; const uint32_t* ovoff;      // the blocks in elimination order (Group::mf_data / mf_voff)
    RobustSpec rk;
    const MfDesc* desc; const uint32_t* rcflat;
    double* Cinv; double* b; double* slab;                               // per member: (C_v + lambda I)^-1 and b_v (b's eliminated part); per supernode: its share of [S | s]
    double lambda; int* status;
    uint32_t wsz, ecap;                                                  // doubles of LDS per wavefront, of which the E slab
    uint32_t nbig, ntiny;                                                // supernodes of several batches (one workgroup each) come first, then those of ONE batch (one wavefront each)
    int dbg;
};
// per-wavefront LDS: [E slab: B x DP x LDC | red: 64 x NRED | sums: BMAX x NRED | cinv: BMAX x DP^2]; at least the supernode's share in slab layout (nlls_ctx::mf_wsz)
NLLS_HD uint32_t mf_wave_lds(uint32_t ecap, int dp) { const uint32_t nred = (uint32_t)(dp * (dp + 1) / 2 + dp); return (ecap + 64 * nred + MF_BMAX * (nred + (uint32_t)(dp * dp)) + 1) & ~1u; }

// All batches first, first + stride, ... of one supernode, by ONE wavefront: evaluation, slab, (C + lambda I)^-1, and the members' rank-DP updates summed into acc.
template <int KIND, int PS, int TRK>
__device__ __forceinline__ void mf_wave_batches(const MfArgs& a, const MfDesc& d, double* __restrict__ Ew, int first, int stride, double4_t (&acc)[TRK * (TRK + 1) / 2]) {
    using R = Res<KIND>; using I = ResInfo<KIND>;
    constexpr int CS = 1 - PS, DP = I::dof(PS), DC = I::dof(CS), NSYM = DP * (DP + 1) / 2, NRED = NSYM + DP, LDC = 16 * TRK;
    const int lane = threadIdx.x & 63, li = lane & 15, lk = lane >> 4;
    const int nd = (int)d.nd, nmem = (int)d.nmem, ncb = nd / DC, B = (int)d.B;
    double* const red = Ew + a.ecap; double* const sums = red + 64 * NRED; double* const cinvw = sums + MF_BMAX * NRED;
    const int ml = lane / ncb, j = lane - ml * ncb;       // this lane's block inside a batch: member ml of the batch, column block j of [E]
    const bool lane_in = ml < B;
    const double* __restrict__ vars = a.vars; const double* __restrict__ odata = a.odata; const uint32_t* __restrict__ ovoff = a.ovoff;
    const RobustSpec rk = a.rk; const double lambda = a.lambda; const int dbg = a.dbg;
    const uint32_t obs0 = d.obs0, v0 = d.v0, eb0 = d.eb0;
    const bool kslot = lk < DP; const int kk = kslot ? lk : 0;
    using St = double[2][MAXST];
    struct Rec { double dd[R::NDATA]; uint32_t vo[2]; };
    auto load_rec = [&](int mb, Rec& r) {                             // unconditional, clamped (a predicated load becomes copy + vmcnt(0))
        const bool on = lane_in && mb + ml < nmem;
        const size_t e = (size_t)obs0 + (on ? (size_t)(mb + ml) * ncb + j : 0);
#pragma unroll
        for (int q = 0; q < R::NDATA; ++q) r.dd[q] = odata[e * R::NDATA + q];
        r.vo[0] = ovoff[e * 2]; r.vo[1] = ovoff[e * 2 + 1];
    };
    Rec r0, r1; St s0, s1;
    load_rec(first * B, r0);
    BlockGH<KIND>::load(vars, r0.vo, s0);
#pragma unroll 1
    for (int mb = first * B; mb < nmem; mb += stride * B) {
        const int nlive = min(B, nmem - mb);
        const bool active = lane_in && ml < nlive;
        load_rec(mb + stride * B, r1);                                // the next batch's records: in flight through this batch's evaluation
        {
            BlockGH<KIND> G; G.compute_st(s0, r0.dd, rk, false);
            if (active && !(dbg & 4)) {
                double* er = Ew + (size_t)(ml * DP) * LDC + DC * j;
#pragma unroll
                for (int k = 0; k < DP; ++k)
#pragma unroll
                    for (int c2 = 0; c2 < DC; ++c2) er[k * LDC + c2] = h_elem<KIND, PS, CS>(G, k, c2);
                double* rr = red + lane * NRED; int q = 0;
#pragma unroll
                for (int c2 = 0; c2 < DP; ++c2)
#pragma unroll
                    for (int r2 = c2; r2 < DP; ++r2) rr[q++] = h_elem<KIND, PS, PS>(G, r2, c2);
#pragma unroll
                for (int r2 = 0; r2 < DP; ++r2) rr[q++] = g_elem<KIND, PS>(G, r2);
            }
        }
        BlockGH<KIND>::load(vars, r1.vo, s1);                         // ... and its variables: in flight through the matrix-core phase
        wave_lds_sync();
        // C_v (lower triangle) and b_v: component q of member m2 summed over the member's lanes
        for (int idx = lane; idx < nlive * NRED; idx += 64) {
            const int m2 = idx / NRED, q = idx - m2 * NRED; const double* rr = red + (size_t)(m2 * ncb) * NRED + q;
            double sum = 0.0; for (int t = 0; t < ncb; ++t) sum += rr[t * NRED];
            sums[idx] = sum;
        }
        wave_lds_sync();
        // (C_v + lambda I)^-1 by LDL' (the arithmetic of schur_cinv_kernel; the pivots' reciprocals by v_rcp_f64 + one cubic step), one lane per member
        if (lane < nlive && !(dbg & 2)) {
            const double* sm = sums + lane * NRED;
            double C[DP * DP], id[DP];
            { int q = 0;
#pragma unroll
              for (int c2 = 0; c2 < DP; ++c2)
#pragma unroll
                  for (int r2 = c2; r2 < DP; ++r2) C[r2 + DP * c2] = sm[q++]; }
            bool bad = false;
#pragma unroll
            for (int c2 = 0; c2 < DP; ++c2) {
                double dd = C[c2 + DP * c2] + lambda;
#pragma unroll
                for (int k = 0; k < c2; ++k) dd -= C[c2 + DP * k] * C[c2 + DP * k] * C[k + DP * k];
                if (!nonzero_bits(dd) || is_nan_bits(dd)) { bad = true; dd = 1.0; }
                C[c2 + DP * c2] = dd; id[c2] = mf_rcp(dd);
#pragma unroll
                for (int r2 = c2 + 1; r2 < DP; ++r2) { double t = C[r2 + DP * c2];
#pragma unroll
                    for (int k = 0; k < c2; ++k) t -= C[r2 + DP * k] * C[c2 + DP * k] * C[k + DP * k];
                    C[r2 + DP * c2] = t * id[c2]; }
            }
            if (bad) atomicCAS(a.status, 0, 1);
            const size_t vi = (size_t)(v0 + mb + lane);
#pragma unroll
            for (int c2 = 0; c2 < DP; ++c2) {
                double y[DP];
#pragma unroll
                for (int r2 = 0; r2 < DP; ++r2) { double t = (r2 == c2) ? 1.0 : 0.0;
#pragma unroll
                    for (int k = 0; k < r2; ++k) t -= C[r2 + DP * k] * y[k]; y[r2] = t; }
#pragma unroll
                for (int r2 = 0; r2 < DP; ++r2) y[r2] *= id[r2];
#pragma unroll
                for (int r2 = DP - 1; r2 >= 0; --r2) { double t = y[r2];
#pragma unroll
                    for (int k = r2 + 1; k < DP; ++k) t -= C[k + DP * r2] * y[k]; y[r2] = t; }
#pragma unroll
                for (int r2 = 0; r2 < DP; ++r2) { cinvw[lane * (DP * DP) + r2 + DP * c2] = y[r2]; a.Cinv[vi * (DP * DP) + r2 + DP * c2] = y[r2]; }
            }
#pragma unroll
            for (int r2 = 0; r2 < DP; ++r2) { const double bv = sm[NSYM + r2]; a.b[eb0 + (size_t)(mb + lane) * DP + r2] = bv; Ew[(size_t)(lane * DP + r2) * LDC + nd] = bv; }   // the right-hand side rides as column nd
        }
        wave_lds_sync();
        // per member: S_supernode += E' (C + lambda I)^-1 [E | b] on the matrix cores -- tile (Rr, Cc) is one instruction whose A operand is lane (i, k) <- e_{16 Rr + i}[k]
        // and whose B operand is lane (j, k) <- y_{16 Cc + j}[k], y = (C + lambda I)^-1 e; both read from the slab (row k of E is contiguous: sixteen lanes, sixteen doubles).
#pragma unroll 1
        for (int m2 = 0; m2 < ((dbg & 1) ? 0 : nlive); ++m2) {
            const double* em = Ew + (size_t)(m2 * DP) * LDC + li;
            double cr[DP];
#pragma unroll
            for (int q = 0; q < DP; ++q) cr[q] = cinvw[m2 * (DP * DP) + kk + DP * q];
            double aop[TRK], bop[TRK];
#pragma unroll
            for (int r2 = 0; r2 < TRK; ++r2) {
                double e[DP];
#pragma unroll
                for (int q = 0; q < DP; ++q) e[q] = em[q * LDC + 16 * r2];
                double av = 0.0, y = 0.0;
#pragma unroll
                for (int q = 0; q < DP; ++q) { if (q == kk) av = e[q]; y = fma(cr[q], e[q], y); }
                aop[r2] = kslot ? av : 0.0; bop[r2] = kslot ? y : 0.0;
            }
#pragma unroll
            for (int Rr = 0; Rr < TRK; ++Rr)
#pragma unroll
                for (int Cc = 0; Cc <= Rr; ++Cc) acc[Rr * (Rr + 1) / 2 + Cc] = __builtin_amdgcn_mfma_f64_16x16x4f64(aop[Rr], bop[Cc], acc[Rr * (Rr + 1) / 2 + Cc], 0, 0, 0);
        }
        wave_lds_sync();
        r0 = r1;
#pragma unroll
        for (int q = 0; q < MAXST; ++q) { s0[0][q] = s1[0][q]; s0[1][q] = s1[1][q]; }
    }
}

// a supernode of several batches: one workgroup, the four wavefronts' tiles meet in LDS and leave with atomics whose lanes cover consecutive addresses of one column of S
template <int KIND, int PS, class LAY>
__device__ __forceinline__ void mf_elim_big(const MfArgs& a, uint32_t bidx, double* lds) {
    using I = ResInfo<KIND>;
    constexpr int DP = I::dof(PS), DC = I::dof(1 - PS);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), li = lane & 15, lk = lane >> 4;
    const MfDesc d = a.desc[bidx];                     // uniform: scalar loads
    const int nd = (int)d.nd, nmem = (int)d.nmem, B = (int)d.B, ncb = nd / DC;
    const int TR = (nd + 1 + 15) >> 4;
    double* const Ew = lds + (size_t)wave * a.wsz;
    for (int i = lane; i < B * DP * 16 * TR; i += 64) Ew[i] = 0.0;      // (the padding columns behind nd stay zero for the whole launch)
    wave_lds_sync();
    const int nbatch = (nmem + B - 1) / B, nact = min(nbatch, MF_ENW);   // wavefronts that have a batch at all
    auto members = [&](auto TRc) {
        constexpr int TRK = decltype(TRc)::value;
        double4_t acc[TRK * (TRK + 1) / 2];
#pragma unroll
        for (int t = 0; t < TRK * (TRK + 1) / 2; ++t) acc[t] = double4_t{0.0, 0.0, 0.0, 0.0};
        mf_wave_batches<KIND, PS, TRK>(a, d, Ew, wave, MF_ENW, acc);
        if (a.dbg & 16) return;
        // The wavefronts' tiles meet in LDS in the layout of the supernode's SLAB (build_schur: one column-major block per pair of neighbour blocks a >= b in list order -- of a
        // diagonal pair the lower triangle --, then the right-hand side) in the place of the wavefronts' regions, and leave with plain coalesced stores: schur_gather_kernel
        // sums the supernodes' shares of every block pair of S in a fixed order, straight into the block cyclic reduction's tiles.  No atomics on HBM: the reduced system
        // -- and with it the step -- is bit-reproducible (the atomic flush of rounds 1-5 took the memory side 45 us for 4.1 M atomics here).
        // Register v of lane (li, lk) = entry (row lk + 4 v, column li) of its tile; wavefront 0's registers cover every entry exactly once: it STORES, the others add.
        __syncthreads();
        double* const img = lds; const int npair = ncb * (ncb + 1) / 2; double* const irhs = img + npair * (DC * DC);
        auto each = [&](auto&& f) {
#pragma unroll
            for (int Rr = 0; Rr < TRK; ++Rr)
#pragma unroll
                for (int Cc = 0; Cc <= Rr; ++Cc)
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        const int pp = 16 * Rr + lk + 4 * v, q = 16 * Cc + li; const double val = acc[Rr * (Rr + 1) / 2 + Cc][v];
                        if (pp < nd && q <= pp) { const int ba = pp / DC, bb = q / DC; f(&img[(ba * (ba + 1) / 2 + bb) * (DC * DC) + (pp - ba * DC) + DC * (q - bb * DC)], val); }
                        else if (pp == nd && q < nd) f(&irhs[q], val);
                    }
        };
        if (wave == 0) each([](double* p, double v) { *p = v; });
        __syncthreads();
        if (wave > 0 && wave < nact && !(a.dbg & 32)) each([](double* p, double v) { atomicAdd(p, v); });
        if (nact > 1) __syncthreads();
        if (a.dbg & 64) return;
        { double* __restrict__ out = a.slab + d.slab; const int len = npair * (DC * DC) + nd;
          for (int i = tid; i < len; i += 64 * MF_ENW) out[i] = img[i]; }
    };
    if (TR == 4) members(std::integral_constant<int, 4>{});
    else if (TR == 5) members(std::integral_constant<int, 5>{});
    else if (TR == 3) members(std::integral_constant<int, 3>{});
    else if (TR == 2) members(std::integral_constant<int, 2>{});
    else members(std::integral_constant<int, 1>{});
}
// a supernode of ONE batch (bundle adjustment: the one or two points at every step of the visibility window): one WAVEFRONT, no workgroup barrier, its tiles leave straight
// from the registers (the workgroup form spent its time in the barriers, the zero fill and the merge of three empty wavefronts: 34 of 127 us at BASELINE config 4)
template <int KIND, int PS, class LAY>
__device__ __forceinline__ void mf_elim_tiny(const MfArgs& a, uint32_t sidx, double* lds) {
    using I = ResInfo<KIND>;
    constexpr int DP = I::dof(PS), DC = I::dof(1 - PS);
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), li = lane & 15, lk = lane >> 4;
    const MfDesc d = a.desc[sidx];
    const int nd = (int)d.nd, B = (int)d.B, ncb = nd / DC;
    const int TR = (nd + 1 + 15) >> 4;
    double* const Ew = lds + (size_t)wave * a.wsz;
    for (int i = lane; i < B * DP * 16 * TR; i += 64) Ew[i] = 0.0;
    wave_lds_sync();
    auto members = [&](auto TRc) {
        constexpr int TRK = decltype(TRc)::value;
        double4_t acc[TRK * (TRK + 1) / 2];
#pragma unroll
        for (int t = 0; t < TRK * (TRK + 1) / 2; ++t) acc[t] = double4_t{0.0, 0.0, 0.0, 0.0};
        mf_wave_batches<KIND, PS, TRK>(a, d, Ew, 0, 1, acc);
        if (a.dbg & (16 | 64)) return;
        // the wavefront's own region takes its tiles in slab layout (mf_elim_big), then they leave with coalesced stores
        double* const img = Ew; const int npair = ncb * (ncb + 1) / 2; double* const irhs = img + npair * (DC * DC);
#pragma unroll
        for (int Rr = 0; Rr < TRK; ++Rr)
#pragma unroll
            for (int Cc = 0; Cc <= Rr; ++Cc)
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int pp = 16 * Rr + lk + 4 * v, q = 16 * Cc + li; const double val = acc[Rr * (Rr + 1) / 2 + Cc][v];
                    if (pp < nd && q <= pp) { const int ba = pp / DC, bb = q / DC; img[(ba * (ba + 1) / 2 + bb) * (DC * DC) + (pp - ba * DC) + DC * (q - bb * DC)] = val; }
                    else if (pp == nd && q < nd) irhs[q] = val;
                }
        wave_lds_sync();
        { double* __restrict__ out = a.slab + d.slab; const int len = npair * (DC * DC) + nd;
          for (int i = lane; i < len; i += 64) out[i] = img[i]; }
    };
    if (TR == 5) members(std::integral_constant<int, 5>{});
    else if (TR == 4) members(std::integral_constant<int, 4>{});
    else if (TR == 3) members(std::integral_constant<int, 3>{});
    else if (TR == 2) members(std::integral_constant<int, 2>{});
    else members(std::integral_constant<int, 1>{});
}

template <int KIND, int PS, class LAY>
__global__ __launch_bounds__(64 * MF_ENW) __attribute__((amdgpu_waves_per_eu(2, 2))) void mf_elim_kernel(MfArgs a) {
    extern __shared__ __attribute__((aligned(16))) double mf_lds[];
    if (blockIdx.x < a.nbig) { mf_elim_big<KIND, PS, LAY>(a, blockIdx.x, mf_lds); return; }
    const uint32_t t = (blockIdx.x - a.nbig) * MF_ENW + (threadIdx.x >> 6); if (t < a.ntiny) mf_elim_tiny<KIND, PS, LAY>(a, a.nbig + t, mf_lds);
}

// ---------------------------------------------------------------------------------------------------------------------------------------------------
// back-substitution: x_v = -(C_v + lambda I)^-1 (b_v - E_v s), s = the reduced system's solution (x_R = -s), E_v s = sum over the member's blocks of H_pc s_c
// ---------------------------------------------------------------------------------------------------------------------------------------------------
struct MfBackArgs {
    const double* vars; const double* odata; const uint32_t* ovoff; RobustSpec rk;
    const MfDesc* desc; const uint32_t* rcflat; const double* Cinv; const double* b; const double* xr; double* x; double* q;
    uint32_t ngroups; const uint32_t* red_boff; int nred, write_red; double* Szero; int64_t nzero; uint32_t nextra; BsfRetract rt;
};
template <int KIND, int PS>
__global__ __launch_bounds__(64 * MF_NW) void mf_backsub_kernel(MfBackArgs a) {
    using R = Res<KIND>; using I = ResInfo<KIND>;
    constexpr int CS = 1 - PS, DP = I::dof(PS), DC = I::dof(CS);
    __shared__ double red[MF_NW][64 * DP], xpw[MF_NW][MF_BMAX * DP], stage[MF_NW][MF_SLOTS][2 * DP]; __shared__ uint32_t stpv[MF_NW][MF_SLOTS]; __shared__ double qred[MF_NW];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (blockIdx.x >= a.ngroups) { backsub_rest_roles((blockIdx.x - a.ngroups) * MF_NW + wave, a.nextra, lane, a.xr, a.x, a.red_boff, a.nred, a.write_red, a.Szero, a.nzero, a.rt); return; }
    const MfDesc d = a.desc[blockIdx.x];
    const int nd = (int)d.nd, nmem = (int)d.nmem, ncb = nd / DC;
    const int B = (int)d.B;
    const int ml = lane / ncb, j = lane - ml * ncb; const bool lane_in = ml < B;
    const double* __restrict__ vars = a.vars; const double* __restrict__ odata = a.odata; const uint32_t* __restrict__ ovoff = a.ovoff; const RobustSpec rk = a.rk;
    const uint32_t obs0 = d.obs0, v0 = d.v0, eb0 = d.eb0;
    // the reduced solution under this lane's column block: the same for every batch
    double sc[DC];
    { const int jj = lane_in ? j : 0;
#pragma unroll
      for (int c2 = 0; c2 < DC; ++c2) sc[c2] = a.xr[a.rcflat[d.rc_off + DC * jj + c2]]; }
    using St = double[2][MAXST];
    struct Rec { double dd[R::NDATA]; uint32_t vo[2]; };
    auto load_rec = [&](int mb, Rec& r) {
        const bool on = lane_in && mb + ml < nmem;
        const size_t e = (size_t)obs0 + (on ? (size_t)(mb + ml) * ncb + j : 0);
#pragma unroll
        for (int q = 0; q < R::NDATA; ++q) r.dd[q] = odata[e * R::NDATA + q];
        r.vo[0] = ovoff[e * 2]; r.vo[1] = ovoff[e * 2 + 1];
    };
    Rec r0, r1; St s0, s1;
    load_rec(wave * B, r0);
    BlockGH<KIND>::load(vars, r0.vo, s0);
    double qacc = 0.0; int slot0 = 0;
#pragma unroll 1
    for (int mb = wave * B; mb < nmem; mb += MF_NW * B, slot0 += B) {
        const int nlive = min(B, nmem - mb);
        const bool active = lane_in && ml < nlive, head = active && j == 0;
        load_rec(mb + MF_NW * B, r1);
        // the member's right-hand side and inverse block, by its first lane: requested now, used behind the sum
        double bv[DP], ci[DP * DP];
        { const size_t m = (size_t)(mb + (head ? ml : 0));
#pragma unroll
          for (int k = 0; k < DP; ++k) bv[k] = a.b[eb0 + m * DP + k];
#pragma unroll
          for (int q = 0; q < DP * DP; ++q) ci[q] = a.Cinv[((size_t)v0 + m) * (DP * DP) + q]; }
        BlockGH<KIND> G; G.compute_st(s0, r0.dd, rk, false);
        double ts[DP];
#pragma unroll
        for (int k = 0; k < DP; ++k) { double t = 0.0;
#pragma unroll
            for (int c2 = 0; c2 < DC; ++c2) t = fma(h_elem<KIND, PS, CS>(G, k, c2), sc[c2], t);
            ts[k] = t; }
        if (active) {
#pragma unroll
            for (int k = 0; k < DP; ++k) red[wave][lane * DP + k] = ts[k]; }
        BlockGH<KIND>::load(vars, r1.vo, s1);
        wave_lds_sync();
        if (head) {
            double accv[DP];
#pragma unroll
            for (int k = 0; k < DP; ++k) accv[k] = 0.0;
            for (int t = 0; t < ncb; ++t)
#pragma unroll
                for (int k = 0; k < DP; ++k) accv[k] += red[wave][(lane + t) * DP + k];
            double xp[DP];
#pragma unroll
            for (int i = 0; i < DP; ++i) { double t = 0.0;
#pragma unroll
                for (int k = 0; k < DP; ++k) t = fma(ci[i + DP * k], bv[k] - accv[k], t);
                xp[i] = -t; }
#pragma unroll
            for (int k = 0; k < DP; ++k) { xpw[wave][ml * DP + k] = xp[k]; stage[wave][slot0 + ml][k] = xp[k]; stage[wave][slot0 + ml][DP + k] = s0[PS][k] + xp[k]; }   // the retraction of a Euclidean block (src/variable.jl:5)
            stpv[wave][slot0 + ml] = r0.vo[PS];
        }
        wave_lds_sync();
        if (active) {
            // the member rows' share of x'Hx:  2 x_v'(E_v x_R) + x_v' C_v x_v  with  E_v x_R = -E_v s  (what quadform_points_body takes from A.data and tE)
            double xp[DP];
#pragma unroll
            for (int k = 0; k < DP; ++k) xp[k] = xpw[wave][ml * DP + k];
            double qv = 0.0;
#pragma unroll
            for (int k = 0; k < DP; ++k) { qv = fma(-2.0 * xp[k], ts[k], qv);
#pragma unroll
                for (int l2 = 0; l2 < DP; ++l2) qv = fma(xp[k] * h_elem<KIND, PS, PS>(G, k, l2), xp[l2], qv); }
            qacc += qv;
        }
        wave_lds_sync();
        r0 = r1;
#pragma unroll
        for (int q = 0; q < MAXST; ++q) { s0[0][q] = s1[0][q]; s0[1][q] = s1[1][q]; }
    }
    // results leave behind the loop: a store between the loads would make every wait a full vmcnt(0) (finding 3)
    {
        const int nb_all = (nmem + B - 1) / B;                        // batches of the supernode; this wavefront took wave, wave + 4, ...
        for (int sl = lane; sl < MF_SLOTS; sl += 64) {
            const int bi = sl / B, mi = sl - bi * B; const int batch = wave + MF_NW * bi; const int m = batch * B + mi;
            if (batch >= nb_all || m >= nmem) continue;
#pragma unroll
            for (int k = 0; k < DP; ++k) a.x[eb0 + (size_t)m * DP + k] = stage[wave][sl][k];
            if (a.rt.on) { const uint32_t pv = stpv[wave][sl];
#pragma unroll
                for (int k = 0; k < DP; ++k) a.rt.vto[pv + k] = stage[wave][sl][DP + k]; }
        }
    }
    qacc = wave_sum_dpp63(qacc);
    if (lane == 63) qred[wave] = qacc;
    __syncthreads();
    if (tid == 0) { double t = 0.0;
#pragma unroll
        for (int w = 0; w < MF_NW; ++w) t += qred[w];
        a.q[blockIdx.x] = t; }
}

// ---------------------------------------------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------------------------------------------
static int herr(nlls_ctx* c, hipError_t e, const char* what) { c->err = std::string(what) + ": " + hipGetErrorString(e); return NLLS_ERR_HIP; }
#define HIPCHK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return herr(c, e_, #expr); } while (0)

template <int KIND, int PS>
static int launch_mf_elim(nlls_ctx* c, const Group& G) {
    if constexpr (Res<KIND>::NDEPS == 2 && Res<KIND>::ADAPT == 0 && !is_cost_kind<KIND> && ResInfo<KIND>::dof(PS < 2 ? PS : 0) <= 3) {
        const unsigned nsn = (unsigned)c->mf_nbig + (unsigned)((c->n_fast_groups - c->mf_nbig + MF_ENW - 1) / MF_ENW);
        MfArgs a{}; a.vars = vars_ptr(c, NLLS_VARS_CURRENT); a.odata = G.mf_data.p; a.ovoff = G.mf_voff.p; a.rk = G.rk; a.desc = c->d_mf_desc.p; a.rcflat = c->d_elim_rc.p; a.nbig = (uint32_t)c->mf_nbig; a.ntiny = (uint32_t)(c->n_fast_groups - c->mf_nbig);
        a.Cinv = c->Cinv.p; a.b = c->b.p; a.slab = c->slab.p; a.lambda = c->lambda; a.status = c->d_status.p; a.wsz = c->mf_wsz; a.ecap = c->mf_ecap; { static const int dbg = [] { const char* e = getenv("NLLS_MF_DBG"); return e ? atoi(e) : 0; }(); a.dbg = dbg; }
        static size_t granted = 0;
        if (c->mf_lds > 64 * 1024 && c->mf_lds > granted) { HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&mf_elim_kernel<KIND, PS, SLayout>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)c->mf_lds)); granted = c->mf_lds; }
        hipLaunchKernelGGL((mf_elim_kernel<KIND, PS, SLayout>), dim3(nsn), dim3(64 * MF_ENW), c->mf_lds, c->stream, a);
        HIPCHK(hipGetLastError());
        return NLLS_OK;
    } else { c->err = "matrix-free trial: kind not eligible"; return NLLS_ERR_UNSUPPORTED; }
}
template <int KIND, int PS>
static int launch_mf_backsub(nlls_ctx* c, const Group& G, const BsfRetract& rt, int write_red, double* zptr, int64_t zcount, unsigned nextra, unsigned nrestwg) {
    if constexpr (Res<KIND>::NDEPS == 2 && Res<KIND>::ADAPT == 0 && !is_cost_kind<KIND> && ResInfo<KIND>::dof(PS < 2 ? PS : 0) <= 3) {
        MfBackArgs a{}; a.vars = vars_ptr(c, NLLS_VARS_CURRENT); a.odata = G.mf_data.p; a.ovoff = G.mf_voff.p; a.rk = G.rk; a.desc = c->d_mf_desc.p; a.rcflat = c->d_elim_rc.p;
        a.Cinv = c->Cinv.p; a.b = c->b.p; a.xr = c->s_ptr(); a.x = c->x.p; a.q = c->mf_q.p; a.ngroups = (uint32_t)c->n_fast_groups; a.red_boff = c->d_red_boff.p; a.nred = (int)c->nred; a.write_red = write_red;
        a.Szero = zptr; a.nzero = zcount; a.nextra = nextra; a.rt = rt;
        const unsigned rest = (nextra + nrestwg + MF_NW - 1) / MF_NW;
        hipLaunchKernelGGL((mf_backsub_kernel<KIND, PS>), dim3((unsigned)c->n_fast_groups + rest), dim3(64 * MF_NW), 0, c->stream, a);
        HIPCHK(hipGetLastError());
        return NLLS_OK;
    } else { c->err = "matrix-free trial: kind not eligible"; return NLLS_ERR_UNSUPPORTED; }
}

// the assembly of the reduced system of a matrix-free trial: the supernodes' launch (their shares into the slabs), then the gather (schur_gather_kernel: the shares of every
// block pair summed in a fixed order + the reduced-reduced blocks + lambda, straight into the block cyclic reduction's tiles)
int enqueue_mf_solve_local(nlls_ctx* c) {
    const int n = (int)c->nred; if (n == 0 || !c->mf_ok) return NLLS_ERR_NOT_READY;
    const Group& G = c->groups[c->mf_group];
    if (!c->status_known_zero) HIPCHK(hipMemsetAsync(c->d_status.p, 0, sizeof(int32_t) * 5, c->stream));
    c->status_known_zero = false;
    int rc = NLLS_ERR_UNSUPPORTED;
    switch (G.res_kind) {
#define X(K) case K: rc = c->mf_ps == 0 ? launch_mf_elim<K, 0>(c, G) : launch_mf_elim<K, 1>(c, G); break;
        NLLS_FOR_EACH_RES(X)
#undef X
    }
    if (rc != NLLS_OK) return rc;
    return enqueue_gather(c);
}
int enqueue_mf_backsub(nlls_ctx* c, const BsfRetract& rt, int write_red, double* zptr, int64_t zcount, unsigned nextra, unsigned nrestwg) {
    const Group& G = c->groups[c->mf_group];
    switch (G.res_kind) {
#define X(K) case K: return c->mf_ps == 0 ? launch_mf_backsub<K, 0>(c, G, rt, write_red, zptr, zcount, nextra, nrestwg) : launch_mf_backsub<K, 1>(c, G, rt, write_red, zptr, zcount, nextra, nrestwg);
        NLLS_FOR_EACH_RES(X)
#undef X
    }
    return NLLS_ERR_UNSUPPORTED;
}
// (nlls_structure.cpp sizes the launches' LDS with this)
uint32_t mf_wave_doubles(uint32_t ecap, int dp) { return mf_wave_lds(ecap, dp); }
int mf_elim_waves() { return MF_ENW; }
int mf_batch_max() { return MF_BMAX; }

}  // namespace nlls
