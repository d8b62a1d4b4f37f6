// nlls_mfb.hip -- the matrix-free LM trial, second half (gfx950): back-substitution + retraction + the trial point's cost + the step statistics in ONE launch.
//
//   x_v = -(C_v + lambda I)^-1 (b_v + E_v x_R)                     (the eliminated blocks of  negate!(solve!(linsystem))  src/iterators.jl:152)
//   update!(to, from, x)                                           src/iterators.jl:155, src/linearsystem.jl:206-213
//   cost(to)                                                       src/iterators.jl:157, src/cost.jl:10-13
//   fast_bAb(H, x), dot(g, x), maximum(abs, x), |x|^2              src/iterators.jl:163, src/optimize.jl:149
// One lane per cost block, as in the elimination (nlls_mf.hip): the block is evaluated again instead of reading its 18 doubles of E from A.data; E_v x_R is a sum over the
// member's lanes; the member's first lane forms x_v and the new value of its variable; every lane then has both new variables of its block in registers (the reduced
// variable's: its old value + its share of x_R, the retraction the workgroups behind the supernodes store) and takes the block's cost at the trial point -- the cost sweep
// of the trial (24 bytes per block once more, a launch of its own) is gone, and so is the launch that carried the step statistics: the members' share of x'Hx, g'x, max |x|
// and |x|^2 leave as one row of partials per workgroup, the reduced blocks' share by a few workgroups behind them (from the reduced solution itself: the scatter of this very
// launch is not visible to them), and mf_trial_finish_kernel sums the rows.
// Built WITH -fno-honor-nans / -fno-signed-zeros like the sweep (the dual numbers' structural zeros fold away: 168 registers with one spilled instead of ten -- three wavefronts per
// SIMD).  A NaN in the trial point still ends in a NaN trial cost -- the residual, its square and the kernel are plain IEEE instructions, nothing here multiplies by a literal
// zero -- and every NaN TEST looks at the bits (is_nan_bits), as in nlls_sweep.hip.
#include <algorithm>
#include <utility>

#include "nlls_wave.hpp"
#include "nlls_post.hpp"
#include "nlls_slayout.hpp"
#include "nlls_mf.hpp"

namespace nlls {

struct MfBackArgs {
    const double* vars; const double* odata; const uint32_t* ovoff; RobustSpec rk;
    const MfDesc* desc; const uint32_t* rcflat; const double* Cinv; const double* b; const double* xr; double* x; double* part;
    uint32_t ngroups; const uint32_t* red_boff; int nred, write_red; double* Szero; int64_t nzero; uint32_t nextra, nrest_wg; BsfRetract rt;
    double* stamps;
    const double* A; const SchurCopy* copies; int64_t ncopy; int npq, nps;      // the reduced blocks' share of the statistics: npq workgroups for x_R' B x_R, nps for max / |x|^2 / g'x over the reduced unknowns
};

// a workgroup's row of partials from its lanes' accumulators (sums by a fixed tree; thread 0 writes)
NLLS_DEV void mfb_row(double* __restrict__ row, double q, double cost, double mx, double nan, double ss, double bx, double (*red)[MF_NW]) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    q = wave_sum_dpp63(q); cost = wave_sum_dpp63(cost); ss = wave_sum_dpp63(ss); bx = wave_sum_dpp63(bx); mx = mfb_wave_max(mx); nan = mfb_wave_max(nan);
    __syncthreads();
    if (lane == 63) { red[0][w] = q; red[1][w] = cost; red[2][w] = mx; red[3][w] = nan; red[4][w] = ss; red[5][w] = bx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double t[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) { t[k] = red[k][0];
#pragma unroll
            for (int u = 1; u < MF_NW; ++u) t[k] = (k == 2 || k == 3) ? fmax(t[k], red[k][u]) : t[k] + red[k][u]; }
#pragma unroll
        for (int k = 0; k < 6; ++k) row[k] = t[k];
    }
}

// does the back-substitution take the trial point's cost along?  Only where the reduced variable's retraction is the plain sum it can repeat bit for bit (Euclidean: src/variable.jl:5);
// other kinds (an SO(3) pose) get mf_cost_kernel behind it -- the SAME lanes, batches and sums, so that every cost the matrix-free path reports is one fixed sum of the same per-block values
// (nlls_sweep_cost takes that kernel too: cost(problem) == result.bestcost bit for bit, as in the reference, where both are the same function)
template <int KIND, int PS> constexpr bool mf_fuses_cost = Res<KIND>::SK[1 - PS] == NLLS_VAR_EUCLIDEAN;
template <int KIND, int PS>
__global__ __launch_bounds__(64 * MF_NW) __attribute__((amdgpu_waves_per_eu(3, 3))) void mf_backsub_kernel(MfBackArgs a) {
    using R = Res<KIND>; using I = ResInfo<KIND>;
    constexpr int CS = 1 - PS, DP = I::dof(PS), DC = I::dof(CS);
    __shared__ double red[MF_NW][64 * DP], xpw[MF_NW][MF_BMAX * DP], stage[MF_NW][MF_SLOTS][2 * DP]; __shared__ uint32_t stpv[MF_NW][MF_SLOTS]; __shared__ double rowred[6][MF_NW];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    double* const row = a.part + (size_t)blockIdx.x * MF_PW;
    if (blockIdx.x == 0 && tid == 0) time_stamp(a.stamps, 1);
    if (blockIdx.x >= a.ngroups) {
        const uint32_t w = blockIdx.x - a.ngroups;
        if (w < a.nrest_wg) {                                  // x_R = -s scattered, S / the tiles zero-filled for the next solve, the other variables retracted (64-thread roles)
            backsub_rest_roles(w * MF_NW + wave, a.nextra, lane, a.xr, a.x, a.red_boff, a.nred, a.write_red, a.Szero, a.nzero, a.rt);
            if (tid < 6) row[tid] = 0.0;
            return;
        }
        if (w < a.nrest_wg + (uint32_t)a.npq) {                // x_R' B x_R over the reduced-reduced blocks (their own reduced offsets; the sign of x_R cancels)
            if (tid == 0) { row[1] = row[2] = row[3] = row[4] = row[5] = 0.0; }
            quadform_blocks_body(a.A, a.copies, a.ncopy, a.xr, nullptr, nullptr, (int)(w - a.nrest_wg), a.npq, row);
            return;
        }
        // max |x|, |x|^2, g'x over the reduced unknowns
        const int b2 = (int)(w - a.nrest_wg - (uint32_t)a.npq);
        double mx = 0, nan = 0, ss = 0, bx = 0;
        for (int i = b2 * 64 * MF_NW + tid; i < a.nred; i += a.nps * 64 * MF_NW) { const double xv = a.write_red ? -a.xr[i] : 0.0; if (is_nan_bits(xv)) nan = 1.0; mx = fmax(mx, fabs(xv)); ss += xv * xv; bx += a.b[a.red_boff[i]] * xv; }
        mfb_row(row, 0.0, 0.0, mx, nan, ss, bx, rowred);
        return;
    }
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
    double qacc = 0.0, cacc = 0.0, macc = 0.0, nacc = 0.0, sacc = 0.0, bacc = 0.0; int slot0 = 0;
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
        // what the block contributes: H_pc s_c (its share of E_v s) and its H_pp (lower triangle) -- formed NOW, so that the block's Jacobian is dead across the waits below
        double ts[DP], hpp[DP * (DP + 1) / 2];
        { BlockGH<KIND> G; G.compute_st(s0, r0.dd, rk, false);
#pragma unroll
          for (int k = 0; k < DP; ++k) { double t = 0.0;
#pragma unroll
              for (int c2 = 0; c2 < DC; ++c2) t = fma(h_elem<KIND, PS, CS>(G, k, c2), sc[c2], t);
              ts[k] = t; }
          int q = 0;
#pragma unroll
          for (int c2 = 0; c2 < DP; ++c2)
#pragma unroll
              for (int r2 = c2; r2 < DP; ++r2) hpp[q++] = h_elem<KIND, PS, PS>(G, r2, c2); }
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
            for (int k = 0; k < DP; ++k) { xpw[wave][ml * DP + k] = xp[k]; stage[wave][slot0 + ml][k] = xp[k]; stage[wave][slot0 + ml][DP + k] = s0[PS][k] + xp[k];   // the retraction of a Euclidean block (src/variable.jl:5)
                if (is_nan_bits(xp[k])) nacc = 1.0; macc = fmax(macc, fabs(xp[k])); sacc = fma(xp[k], xp[k], sacc); bacc = fma(bv[k], xp[k], bacc); }
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
            for (int k = 0; k < DP; ++k) qv = fma(-2.0 * xp[k], ts[k], qv);
            { int q = 0;
#pragma unroll
              for (int c2 = 0; c2 < DP; ++c2)
#pragma unroll
                  for (int r2 = c2; r2 < DP; ++r2) { qv = fma((r2 == c2 ? 1.0 : 2.0) * xp[r2] * hpp[q], xp[c2], qv); ++q; } }
            qacc += qv;
            // the block at the TRIAL point: its eliminated variable + x_v, its reduced variable retracted by its share of x_R = -s (update(), src/variable.jl: the arithmetic
            // of the workgroups that store it) -- and the block's cost there (computerescost, src/residual.jl:49-55)
            if constexpr (mf_fuses_cost<KIND, PS>) {
                double stn[2][MAXST];
#pragma unroll
                for (int c2 = 0; c2 < DC; ++c2) stn[CS][c2] = s0[CS][c2] + (a.write_red ? -sc[c2] : 0.0);
#pragma unroll
                for (int k = 0; k < DP; ++k) stn[PS][k] = s0[PS][k] + xp[k];
                cacc += block_cost_st<KIND>(stn, r0.dd, rk);
            }
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
    mfb_row(row, qacc, cacc, macc, nacc, sacc, bacc, rowred);
}

// cost(vars, costs) (src/cost.jl:10-13) in the lanes, batches and sums of the back-substitution: row bid of `part` gets the supernode's cost in column 1 (only == 0: the other
// columns are left as the back-substitution wrote them; else they are zeroed)
struct MfCostArgs { const double* vars; const double* odata; const uint32_t* ovoff; RobustSpec rk; const MfDesc* desc; double* part; int only; };
template <int KIND, int PS>
__global__ __launch_bounds__(64 * MF_NW) void mf_cost_kernel(MfCostArgs a) {
    using R = Res<KIND>; using I = ResInfo<KIND>;
    constexpr int CS = 1 - PS, DC = I::dof(CS);
    __shared__ double rowred[6][MF_NW];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const MfDesc d = a.desc[blockIdx.x];
    const int nd = (int)d.nd, nmem = (int)d.nmem, ncb = nd / DC, B = (int)d.B;
    const int ml = lane / ncb, j = lane - ml * ncb; const bool lane_in = ml < B;
    double cacc = 0.0;
#pragma unroll 1
    for (int mb = wave * B; mb < nmem; mb += MF_NW * B) {
        const bool active = lane_in && mb + ml < nmem;
        const size_t e = (size_t)d.obs0 + (active ? (size_t)(mb + ml) * ncb + j : 0);
        double dd[R::NDATA]; uint32_t vo[2]; double st[2][MAXST];
#pragma unroll
        for (int q = 0; q < R::NDATA; ++q) dd[q] = a.odata[e * R::NDATA + q];
        vo[0] = a.ovoff[e * 2]; vo[1] = a.ovoff[e * 2 + 1];
        BlockGH<KIND>::load(a.vars, vo, st);
        const double cb = block_cost_st<KIND>(st, dd, a.rk);
        if (active) cacc += cb;
    }
    double* const row = a.part + (size_t)blockIdx.x * MF_PW;
    cacc = wave_sum_dpp63(cacc);
    if (lane == 63) rowred[1][wave] = cacc;
    __syncthreads();
    if (tid == 0) { double t = rowred[1][0];
#pragma unroll
        for (int u = 1; u < MF_NW; ++u) t += rowred[1][u];
        row[1] = t; if (a.only) { row[0] = row[2] = row[3] = row[4] = row[5] = 0.0; } }
}
// ... and its sum in the order of mf_finish_body (thread t: rows t, t + 256, ...; the rows behind the supernodes' hold no cost)
__global__ __launch_bounds__(256) void mf_cost_reduce_kernel(const double* __restrict__ part, int nrows, double* __restrict__ out) {
    __shared__ double red[4];
    double cost = 0;
    for (int i = threadIdx.x; i < nrows; i += 256) cost += part[(size_t)i * MF_PW + 1];
    cost = mfb_wave_sum(cost);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = cost;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = red[0] + red[1] + red[2] + red[3];
}

// The end of a matrix-free trial: ONE workgroup sums the rows of partials (fixed order: the totals are bit-reproducible) into the trial's scalars -- out[0] cost, [1] max|x| (NaN if
// any entry is), [2] x'x, [4] x'(H + lambda I)x, [5] g'x, [8] x'Hx, [9] x'x, [10] factorisation status: what trial_finish_kernel leaves -- and publishes them to the pinned host
// mirror with the trial's sequence number; the workgroups behind it zero-fill the rows the look-ahead sweep accumulates into with atomics (ZeroRanges).
struct MfZero { double* A; const int64_t* off; const uint32_t* len; double* b; const uint32_t* boff; const uint32_t* blen; int n; };
__global__ __launch_bounds__(256) void mf_trial_finish_kernel(MfFin fin, MfZero zr) {
    if (blockIdx.x >= 1) {
        const int r = (int)blockIdx.x - 1;
        const int64_t o = zr.off[r]; const uint32_t l = zr.len[r];
        for (uint32_t i = threadIdx.x; i < l; i += 256) zr.A[o + i] = 0.0;
        const uint32_t bo = zr.boff[r], bl = zr.blen[r];
        for (uint32_t i = threadIdx.x; i < bl; i += 256) zr.b[bo + i] = 0.0;
        return;
    }
    __shared__ double red[6][4];
    mf_finish_body(fin, red);
}

// ---------------------------------------------------------------------------------------------------------------------------------------------------
static int herr(nlls_ctx* c, hipError_t e, const char* what) { c->err = std::string(what) + ": " + hipGetErrorString(e); return NLLS_ERR_HIP; }
#define HIPCHK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return herr(c, e_, #expr); } while (0)

template <int KIND, int PS>
static int launch_mf_backsub(nlls_ctx* c, const Group& G, const BsfRetract& rt, int write_red, double* zptr, int64_t zcount, unsigned nextra, unsigned nrestwg) {
    if constexpr (Res<KIND>::NDEPS == 2 && Res<KIND>::ADAPT == 0 && !is_cost_kind<KIND> && ResInfo<KIND>::dof(PS < 2 ? PS : 0) <= 3) {
        MfBackArgs a{}; a.vars = vars_ptr(c, NLLS_VARS_CURRENT); a.odata = G.mf_data.p; a.ovoff = G.mf_voff.p; a.rk = G.rk; a.desc = c->d_mf_desc.p; a.rcflat = c->d_elim_rc.p;
        a.Cinv = c->Cinv.p; a.b = c->b.p; a.xr = c->s_ptr(); a.x = c->x.p; a.part = c->mf_q.p; a.ngroups = (uint32_t)c->n_fast_groups; a.red_boff = c->d_red_boff.p; a.nred = (int)c->nred; a.write_red = write_red;
        a.Szero = zptr; a.nzero = zcount; a.nextra = nextra; a.rt = rt; a.stamps = c->stamp_ptr();
        a.nrest_wg = (nextra + nrestwg + MF_NW - 1) / MF_NW;
        a.A = c->A.p; a.copies = c->d_copy.p; a.ncopy = c->ncopy;
        a.npq = (int)std::max<int64_t>(1, std::min<int64_t>((c->ncopy * QF_COLS + 255) / 256, 64)); a.nps = (int)std::max<int64_t>(1, std::min<int64_t>((c->nred + 255) / 256, 32));
        const unsigned grid = (unsigned)c->n_fast_groups + a.nrest_wg + (unsigned)a.npq + (unsigned)a.nps;
        if ((size_t)grid * MF_PW > c->mf_q.n) { c->err = "matrix-free trial: partials buffer too small"; return NLLS_ERR_HIP; }
        c->mf_rows = (int)grid;
        hipLaunchKernelGGL((mf_backsub_kernel<KIND, PS>), dim3(grid), dim3(64 * MF_NW), 0, c->stream, a);
        if constexpr (!mf_fuses_cost<KIND, PS>) {      // (the trial point is in memory now: its cost by the same lanes and sums, into column 1 of the supernodes' rows)
            MfCostArgs ca{rt.vto, G.mf_data.p, G.mf_voff.p, G.rk, c->d_mf_desc.p, c->mf_q.p, 0};
            hipLaunchKernelGGL((mf_cost_kernel<KIND, PS>), dim3((unsigned)c->n_fast_groups), dim3(64 * MF_NW), 0, c->stream, ca);
        }
        HIPCHK(hipGetLastError());
        return NLLS_OK;
    } else { c->err = "matrix-free trial: kind not eligible"; return NLLS_ERR_UNSUPPORTED; }
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
// the end of the trial whose back-substitution launch has left the rows of partials (nlls_ctx::mf_rows): one finishing workgroup, nothing else -- as a launch of its own, or
// (nlls_ctx::mf_fin_defer) as the first workgroup of the look-ahead sweep's launch behind it (nlls_sweep.hip takes mf_fin_pending along)
MfFin mf_fin_args(nlls_ctx* c) { return MfFin{c->mf_q.p, c->mf_rows, c->lambda, c->scalars.p, c->d_status.p, c->h_scalars_dev, (double)c->trial_seq, c->stamp_ptr()}; }
int enqueue_mf_trial_finish(nlls_ctx* c) {
    ++c->trial_seq;
    if (c->mf_fin_defer) { c->mf_fin_pending = true; return NLLS_OK; }
    return enqueue_mf_trial_finish_now(c);
}
int enqueue_mf_trial_finish_now(nlls_ctx* c) {
    c->mf_fin_pending = false;
    MfZero zr{};
    if (c->tail_zero_for_lookahead && c->nzero > 0) { zr = MfZero{c->A.p, c->d_zero_off.p, c->d_zero_len.p, c->b.p, c->d_zero_b_off.p, c->d_zero_b_len.p, (int)c->nzero}; c->heavy_rows_zeroed = true; }
    hipLaunchKernelGGL(mf_trial_finish_kernel, dim3(1 + (unsigned)zr.n), dim3(256), 0, c->stream, mf_fin_args(c), zr);
    HIPCHK(hipGetLastError());
    return NLLS_OK;
}
// cost(vars[which]) of a problem the matrix-free trial applies to: the trial's own sum (scalars[0])
template <int KIND, int PS>
static int launch_mf_cost(nlls_ctx* c, const Group& G, int which) {
    if constexpr (Res<KIND>::NDEPS == 2 && Res<KIND>::ADAPT == 0 && !is_cost_kind<KIND> && ResInfo<KIND>::dof(PS < 2 ? PS : 0) <= 3) {
        MfCostArgs ca{vars_ptr(c, which), G.mf_data.p, G.mf_voff.p, G.rk, c->d_mf_desc.p, c->mf_q.p, 1};
        hipLaunchKernelGGL((mf_cost_kernel<KIND, PS>), dim3((unsigned)c->n_fast_groups), dim3(64 * MF_NW), 0, c->stream, ca);
        hipLaunchKernelGGL(mf_cost_reduce_kernel, dim3(1), dim3(256), 0, c->stream, c->mf_q.p, (int)c->n_fast_groups, c->scalars.p);
        HIPCHK(hipGetLastError());
        return NLLS_OK;
    } else { c->err = "matrix-free trial: kind not eligible"; return NLLS_ERR_UNSUPPORTED; }
}
int enqueue_mf_sweep_cost(nlls_ctx* c, int which) {
    const Group& G = c->groups[c->mf_group];
    switch (G.res_kind) {
#define X(K) case K: return c->mf_ps == 0 ? launch_mf_cost<K, 0>(c, G, which) : launch_mf_cost<K, 1>(c, G, which);
        NLLS_FOR_EACH_RES(X)
#undef X
    }
    return NLLS_ERR_UNSUPPORTED;
}
size_t mf_part_doubles(int64_t nsupernodes, int64_t nrest_wg_max) { return (size_t)(nsupernodes + nrest_wg_max + 64 + 32 + 8) * MF_PW; }

}  // namespace nlls
