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
#include "nlls_mf.hpp"

namespace nlls {

struct MfArgs {
    const double* vars; const double* odata; const uint32_t* ovoff;      // the blocks in elimination order (Group::mf_data / mf_voff)
    RobustSpec rk;
    const MfDesc* desc; const uint32_t* rcflat;
    double* Cinv; double* b; double* slab;                               // per member: (C_v + lambda I)^-1 (for the back-substitution) and b_v (b's eliminated part); per supernode: its share of [S | s]
    double lambda; int* status;
    uint32_t wsz, ecap;                                                  // doubles of LDS per wavefront, of which the E slab
    double* stamps;                                                      // nlls_ctx::stamp_ptr (device-timed buckets)
    uint32_t nbig, ntiny;                                                // supernodes of several batches (one workgroup each) come first, then those of ONE batch (one wavefront each)
};
// per-wavefront LDS: [slab: B x DP x LDC rows of L^-1 [E | b], then ONE zero row | red: 64 x NRED | sums: BMAX x NRED | factors: BMAX x (DP + 1) x DP (per member: L below the diagonal, then 1 / D)];
// at least the supernode's share in slab layout (nlls_ctx::mf_wsz).
// The matrix-core loop takes FOUR rows of the batch per instruction, across members; the k-slots behind the batch's last row read the ZERO row behind the members' rows -- like the
// others, no selects on the operands.  (Round 6, late: a zero row PER MEMBER was a quarter of the slab and of the matrix-core time.)
// LDS strides of the slab: a row is 16 TR doubles + 8 (consecutive rows then start 16 banks apart: the lanes of the four k-slots read their A operands in ONE instruction -- with rows a
// multiple of 64 doubles apart all four hit the same banks: 117 instead of 83 us), a member DP rows + 2 (EVEN: with an odd member stride the launch took 115 instead of 82 us; the
// evaluation's stores of one instruction come from up to eight members)
#ifndef MF_ROWPAD
#define MF_ROWPAD 8
#endif
#ifndef MF_MEMPAD
#define MF_MEMPAD 2      /* even: with an ODD member stride the launch took 115 instead of 82 us at BASELINE config 4 */
#endif
NLLS_HD constexpr int mf_row_stride(int tr) { return 16 * tr + MF_ROWPAD; }
NLLS_HD constexpr int mf_member_stride(int dp, int tr) { return dp * mf_row_stride(tr) + MF_MEMPAD; }
NLLS_HD uint32_t mf_wave_lds(uint32_t ecap, int dp) { const uint32_t nred = (uint32_t)(dp * (dp + 1) / 2 + dp); return (ecap + 64 * nred + MF_BMAX * (nred + (uint32_t)((dp + 1) * dp)) + 1) & ~1u; }

// All batches first, first + stride, ... of one supernode, by ONE wavefront: evaluation, slab, (C + lambda I)^-1, and the members' rank-DP updates summed into acc.
template <int KIND, int PS, int TRK>
__device__ __forceinline__ void mf_wave_batches(const MfArgs& a, const MfDesc& d, double* __restrict__ Ew, int first, int stride, double4_t (&acc)[TRK * (TRK + 1) / 2]) {
    using R = Res<KIND>; using I = ResInfo<KIND>;
    constexpr int CS = 1 - PS, DP = I::dof(PS), DC = I::dof(CS), NSYM = DP * (DP + 1) / 2, NRED = NSYM + DP, LDC = mf_row_stride(TRK), MST = mf_member_stride(DP, TRK);
    const int lane = threadIdx.x & 63, li = lane & 15, lk = lane >> 4;
    const int nd = (int)d.nd, nmem = (int)d.nmem, ncb = nd / DC, B = (int)d.B;
    double* const red = Ew + a.ecap; double* const sums = red + 64 * NRED; double* const cinvw = sums + MF_BMAX * NRED;
    const int ml = lane / ncb, j = lane - ml * ncb;       // this lane's block inside a batch: member ml of the batch, column block j of [E]
    const bool lane_in = ml < B;
    const double* __restrict__ vars = a.vars; const double* __restrict__ odata = a.odata; const uint32_t* __restrict__ ovoff = a.ovoff;
    const RobustSpec rk = a.rk; const double lambda = a.lambda;
    const uint32_t obs0 = d.obs0, v0 = d.v0, eb0 = d.eb0;
    const double* const zrow = Ew + (size_t)B * MST + li;                      // the slab's one zero row (rows behind the batch's last in the matrix-core loop)
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
        double ev[DP][DC];                                            // this lane's block of E: in registers until the member's factor L is known (it leaves for the slab as L^-1 e)
        {
            BlockGH<KIND> G; G.compute_st(s0, r0.dd, rk, false);
#pragma unroll
            for (int k = 0; k < DP; ++k)
#pragma unroll
                for (int c2 = 0; c2 < DC; ++c2) ev[k][c2] = h_elem<KIND, PS, CS>(G, k, c2);
            if (active) {
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
            // (four values requested together: one wait per four, not one per value -- a run-time loop of single dependent LDS reads was a chain of ncb round trips)
            double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0; int t = 0;
            for (; t + 4 <= ncb; t += 4) { const double v0 = rr[t * NRED], v1 = rr[(t + 1) * NRED], v2 = rr[(t + 2) * NRED], v3 = rr[(t + 3) * NRED]; s0 += v0; s1 += v1; s2 += v2; s3 += v3; }
            for (; t < ncb; ++t) s0 += rr[t * NRED];
            sums[idx] = (s0 + s1) + (s2 + s3);
        }
        wave_lds_sync();
        // (C_v + lambda I)^-1 by LDL' (the arithmetic of schur_cinv_kernel; the pivots' reciprocals by v_rcp_f64 + one cubic step), one lane per member
        if (lane < nlive) {
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
                for (int r2 = 0; r2 < DP; ++r2) a.Cinv[vi * (DP * DP) + r2 + DP * c2] = y[r2];
            }
            // the member's factor for the wave: unit lower L (row-wise, below the diagonal) and 1 / D -- the matrix-core loop forms E' C^-1 E as (L^-1 E)' D^-1 (L^-1 E): both operands
            // from the SAME rows f = L^-1 e, the B operand one multiplication by 1 / d (with y = C^-1 e as B operand every k-slot read DP rows and did DP FMAs per tile column)
            { double* lw = cinvw + lane * ((DP + 1) * DP); int q = 0;
#pragma unroll
              for (int r2 = 1; r2 < DP; ++r2)
#pragma unroll
                  for (int c2 = 0; c2 < r2; ++c2) lw[q++] = C[r2 + DP * c2];
#pragma unroll
              for (int r2 = 0; r2 < DP; ++r2) lw[NSYM - DP + r2] = id[r2]; }
            double fb[DP];
#pragma unroll
            for (int r2 = 0; r2 < DP; ++r2) { const double bv = sm[NSYM + r2]; a.b[eb0 + (size_t)(mb + lane) * DP + r2] = bv; double t = bv;
#pragma unroll
                for (int k = 0; k < r2; ++k) t -= C[r2 + DP * k] * fb[k];
                fb[r2] = t; Ew[(size_t)lane * MST + r2 * LDC + nd] = t; }   // the right-hand side rides as column nd (forward-substituted like the rows of E)
        }
        wave_lds_sync();
        if (active) {                                                 // f = L^-1 e of this lane's block, into the slab
            const double* lw = cinvw + ml * ((DP + 1) * DP);
            double lv[NSYM - DP > 0 ? NSYM - DP : 1];
#pragma unroll
            for (int q = 0; q < NSYM - DP; ++q) lv[q] = lw[q];
            double* er = Ew + (size_t)ml * MST + DC * j;
#pragma unroll
            for (int c2 = 0; c2 < DC; ++c2) {
                double f[DP];
#pragma unroll
                for (int r2 = 0; r2 < DP; ++r2) { double t = ev[r2][c2];
#pragma unroll
                    for (int k = 0; k < r2; ++k) t -= lv[r2 * (r2 - 1) / 2 + k] * f[k];
                    f[r2] = t; er[r2 * LDC + c2] = t; }
            }
        }
        wave_lds_sync();
        // S_supernode += E' (C + lambda I)^-1 [E | b] on the matrix cores, FOUR ROWS of [E | b] per step: the k-slots of v_mfma_f64_16x16x4_f64 take rows rho = 4 t + lk of the batch's
        // DP nlive rows -- whichever members they belong to (a member's DP = 3 rows used three of the four slots: a quarter of the matrix-core time multiplied zeros) --: tile (Rr, Cc)
        // is one instruction whose A operand is lane (i, k) <- f_rho[16 Rr + i] and whose B operand is lane (j, k) <- f_rho[16 Cc + j] / d_rho, f = L_m^-1 [e | b] of the row's member m
        // (C_m + lambda I = L D L'); read from the slab (a row is contiguous: sixteen lanes, sixteen doubles).  Rows behind the last: the slab's zero row.
        const int nrow = nlive * DP;
#pragma unroll 1
        for (int t0 = 0; t0 < nrow; t0 += 4) {
            const int rho = t0 + lk; const bool on = rho < nrow; const int m2 = on ? rho / DP : 0, q2 = rho - m2 * DP;
            const double* ea = on ? Ew + (size_t)m2 * MST + q2 * LDC + li : zrow;
            const double dinv = cinvw[(on ? m2 * ((DP + 1) * DP) + q2 : 0) + NSYM - DP];
            double aop[TRK], bop[TRK];
#pragma unroll
            for (int r2 = 0; r2 < TRK; ++r2) { aop[r2] = ea[16 * r2]; bop[r2] = aop[r2] * dinv; }
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
    for (int i = lane; i < B * mf_member_stride(DP, TR) + mf_row_stride(TR); i += 64) Ew[i] = 0.0;      // (the padding columns behind nd and the zero row stay zero for the whole launch)
    { double* const cz = Ew + a.ecap + 64 * (DP * (DP + 1) / 2 + DP) + MF_BMAX * (DP * (DP + 1) / 2 + DP); for (int i = lane; i < MF_BMAX * (DP + 1) * DP; i += 64) cz[i] = 0.0; }
    wave_lds_sync();
    const int nbatch = (nmem + B - 1) / B, nact = min(nbatch, MF_ENW);   // wavefronts that have a batch at all
    auto members = [&](auto TRc) {
        constexpr int TRK = decltype(TRc)::value;
        double4_t acc[TRK * (TRK + 1) / 2];
#pragma unroll
        for (int t = 0; t < TRK * (TRK + 1) / 2; ++t) acc[t] = double4_t{0.0, 0.0, 0.0, 0.0};
        mf_wave_batches<KIND, PS, TRK>(a, d, Ew, wave, MF_ENW, acc);
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
        if (wave > 0 && wave < nact) each([](double* p, double v) { atomicAdd(p, v); });
        if (nact > 1) __syncthreads();
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
    for (int i = lane; i < B * mf_member_stride(DP, TR) + mf_row_stride(TR); i += 64) Ew[i] = 0.0;
    { double* const cz = Ew + a.ecap + 64 * (DP * (DP + 1) / 2 + DP) + MF_BMAX * (DP * (DP + 1) / 2 + DP); for (int i = lane; i < MF_BMAX * (DP + 1) * DP; i += 64) cz[i] = 0.0; }
    wave_lds_sync();
    auto members = [&](auto TRc) {
        constexpr int TRK = decltype(TRc)::value;
        double4_t acc[TRK * (TRK + 1) / 2];
#pragma unroll
        for (int t = 0; t < TRK * (TRK + 1) / 2; ++t) acc[t] = double4_t{0.0, 0.0, 0.0, 0.0};
        mf_wave_batches<KIND, PS, TRK>(a, d, Ew, 0, 1, acc);
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
    if (blockIdx.x == 0 && threadIdx.x == 0) time_stamp(a.stamps, 0);
    if (blockIdx.x < a.nbig) { mf_elim_big<KIND, PS, LAY>(a, blockIdx.x, mf_lds); return; }
    const uint32_t t = (blockIdx.x - a.nbig) * MF_ENW + (threadIdx.x >> 6); if (t < a.ntiny) mf_elim_tiny<KIND, PS, LAY>(a, a.nbig + t, mf_lds);
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
        a.stamps = c->stamp_ptr(); a.Cinv = c->Cinv.p; a.b = c->b.p; a.slab = c->slab.p; a.lambda = c->lambda; a.status = c->d_status.p; a.wsz = c->mf_wsz; a.ecap = c->mf_ecap;
        static size_t granted = 0;
        if (c->mf_lds > 64 * 1024 && c->mf_lds > granted) { HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&mf_elim_kernel<KIND, PS, SLayout>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)c->mf_lds)); granted = c->mf_lds; }
        hipLaunchKernelGGL((mf_elim_kernel<KIND, PS, SLayout>), dim3(nsn), dim3(64 * MF_ENW), c->mf_lds, c->stream, a);
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
// (nlls_structure.cpp sizes the launches' LDS with this)
uint32_t mf_wave_doubles(uint32_t ecap, int dp) { return mf_wave_lds(ecap, dp); }
int mf_elim_waves() { return MF_ENW; }
int mf_batch_max() { return MF_BMAX; }
uint32_t mf_slab_doubles(int B, int dp, int tr) { return (uint32_t)(B * mf_member_stride(dp, tr) + mf_row_stride(tr)); }      // the members' rows + the zero row

}  // namespace nlls
