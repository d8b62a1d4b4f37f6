// nlls_cost.hip -- cost sweep, retraction, step statistics and optimizesingles! (gfx950).
//
//   cost(vars, costs)                     src/cost.jl:10-13 -> src/residual.jl:49-55
//   update!(to, from, linsystem)          src/linearsystem.jl:206-213
//   optimizesingles!                      src/optimize.jl:60-76,183-205
//
// Built WITHOUT the -fno-honor-nans / -fno-signed-zeros flags of nlls_sweep.hip: the NaN / Inf behaviour of these kernels is part of
// the reference's semantics (a NaN cost is accepted by '!(cost > bestcost)', the termination flags report non-finite costs and steps).
#include <cstring>
#include <utility>

#include "nlls_wave.hpp"
#include "nlls_post.hpp"

namespace nlls {

// ================================================================================================
// cost sweep   src/cost.jl:10-13 -> src/residual.jl:49-55
// ================================================================================================
// POST: the workgroups in front of the last `cgrid` carry the step-statistics roles of an LM trial (nlls_post.hpp: quadratic form, g'x, max |x| of the step the
// back-substitution in front of this launch has just formed -- independent of the cost blocks, and a launch of their own would cost more than they do)
template <int KIND, bool POST>
__global__ __launch_bounds__(TPB) void cost_kernel(const double* __restrict__ vars, const double* __restrict__ data,
                                                   const uint32_t* __restrict__ voff, const uint32_t* __restrict__ index,
                                                   int64_t n, RobustSpec rk, double* __restrict__ partials, int cgrid, PostSolveArgs post) {
    using R = Res<KIND>;
    // (the roles come FIRST in the grid: their chains of dependent loads are the longer ones, and the cost blocks alone fill every wave slot of the chip --
    //  behind them the roles would only start when the sweep is over)
    int bid = (int)blockIdx.x;
    if constexpr (POST) { if (blockIdx.x == 0 && threadIdx.x == 0) time_stamp(post.stamps, 2);
        const int npost = (int)gridDim.x - cgrid; if (bid < npost) { post_roles_any(post, bid); return; } bid -= npost; }
    __shared__ double red[TPB / 64];
    double acc = 0;
    for (int64_t i = (int64_t)bid * TPB + threadIdx.x; i < n; i += (int64_t)cgrid * TPB) {
        const int64_t k = index ? index[i] : i;
        double d[R::NDATA]; uint32_t vo[R::NDEPS];
#pragma unroll
        for (int q = 0; q < R::NDATA; ++q) d[q] = data[k * R::NDATA + q];
#pragma unroll
        for (int q = 0; q < R::NDEPS; ++q) vo[q] = voff[k * R::NDEPS + q];
        acc += block_cost<KIND>(vars, vo, d, rk);
    }
    double t = block_sum(acc, red);
    if (threadIdx.x == 0) partials[bid] = t;
}

// final deterministic reduction of the per-workgroup partials
__global__ __launch_bounds__(TPB) void reduce_partials_kernel(const double* __restrict__ partials, int64_t n, double* __restrict__ out) {
    __shared__ double red[TPB / 64];
    reduce_partials_body(partials, n, out, red);
}

// ================================================================================================
// dynamic-size residual blocks   computeresjacdynamic, src/autodiff.jl:96-121 (heap-allocated, run-time sizes in the reference)
// The registered residuals have closed-form Jacobians: LinearResidual X'w - y (test/dynamicvars.jl:3-11), J = X'; NormResidual w
// (test/dynamicvars.jl:13-21), J = I.  One workgroup per block; n = the variable's run-time length.  With A != nullptr the block's
// J'J and J'r go into the dense linear system (src/residual.jl:72-74, src/linearsystem.jl:132-175), else only the cost.
// ================================================================================================
// Under a robust kernel rho (round 3): cost = rho(r'r) / 2, g = rho' J'r, H = rho' J'J + 2 rho'' (J'r)(J'r)'   (src/residual.jl:76-101 applies to any
// residual, the dynamic ones included); the non-squared cost kind takes none.
__global__ __launch_bounds__(TPB) void dyn_block_kernel(int kind, int n, int ndata, const double* __restrict__ vars, const double* __restrict__ data,
                                                        const uint32_t* __restrict__ voff, const uint32_t* __restrict__ index, const uint32_t* __restrict__ brow,
                                                        int ndof, double* __restrict__ A, double* __restrict__ b, double* __restrict__ partials, RobustSpec rk,
                                                        const uint32_t* __restrict__ aoff = nullptr) {
    __shared__ double red[TPB / 64]; __shared__ double total;
    const int64_t k = index ? index[blockIdx.x] : blockIdx.x;
    const double* w = vars + voff[k];
    const uint32_t bo = (A && brow) ? brow[blockIdx.x] : DEST_NONE;            // (brow is in launch order: one entry per launched block)
    // where H(i, j) of the block goes: the dense system (lower triangle: mirrored behind the sweep), or -- aoff -- the variable's n x n diagonal block of a
    // block-sparse system, which is stored in full (src/linearsystem.jl:140)
    const bool blk = aoff != nullptr; const size_t abase = blk ? (size_t)aoff[blockIdx.x] : 0;
    auto Hat = [&](int i, int j) -> double* { return blk ? A + abase + i + (size_t)n * j : A + (bo + i) + (size_t)ndof * (bo + j); };
    const bool robust = (rk.kind & 0xF) != NLLS_ROBUST_NONE || (rk.kind & NLLS_ROBUST_SCALED);
    double cost;
    // rho, rho', rho'' at c = r'r (every thread: c is the workgroup's total)
    auto kernel_at = [&](double c, double& rho, double& d1, double& d2) { if (robust) robustifydcost_fixed(rk, c, rho, d1, d2); else { rho = c; d1 = 1.0; d2 = 0.0; } };
    if (kind == NLLS_RES_DYN_LINEAR) {
        const double* dd = data + k * (int64_t)ndata; const double* X = dd + 1;
        double acc = 0;
        for (int i = threadIdx.x; i < n; i += TPB) acc += X[i] * w[i];
        { const double t = block_sum(acc, red); if (threadIdx.x == 0) total = t; }   // (block_sum leaves the total in thread 0)
        __syncthreads();
        const double r = total - dd[0];
        double rho, d1, d2; kernel_at(r * r, rho, d1, d2);
        cost = 0.5 * rho;
        if (bo != DEST_NONE) {
            const double hs = d1 + 2.0 * d2 * r * r;                             // J'J = X X', J'r = X r: H = (rho' + 2 rho'' r^2) X X'
            for (int i = threadIdx.x; i < n; i += TPB) atomicAdd(&b[bo + i], d1 * (X[i] * r));
            for (int64_t e = threadIdx.x; e < (int64_t)n * n; e += TPB) { const int i = (int)(e % n), j = (int)(e / n); if (i >= j || blk) atomicAdd(Hat(i, j), hs * (X[i] * X[j])); }
        }
    } else if (kind == NLLS_RES_DYN_LINEARSQ) {                                   // X*w - y, X square (n <= 512): J = X
        __shared__ double rs[512], gs[512];
        const double* dd = data + k * (int64_t)ndata; const double* X = dd + n;
        for (int i = threadIdx.x; i < n; i += TPB) { double t = -dd[i]; for (int j = 0; j < n; ++j) t = fma(X[i + (size_t)n * j], w[j], t); rs[i] = t; }
        __syncthreads();
        double acc = 0;
        for (int i = threadIdx.x; i < n; i += TPB) acc += rs[i] * rs[i];
        { const double t = block_sum(acc, red); if (threadIdx.x == 0) total = t; }
        __syncthreads();
        double rho, d1, d2; kernel_at(total, rho, d1, d2);
        cost = 0.5 * rho;
        if (bo != DEST_NONE) {
            for (int j = threadIdx.x; j < n; j += TPB) { double t = 0; for (int i = 0; i < n; ++i) t = fma(X[i + (size_t)n * j], rs[i], t); gs[j] = t; atomicAdd(&b[bo + j], d1 * t); }
            __syncthreads();
            for (int64_t e = threadIdx.x; e < (int64_t)n * n; e += TPB) { const int j = (int)(e % n), q = (int)(e / n); if (j < q && !blk) continue;
                double h = 0; for (int i = 0; i < n; ++i) h = fma(X[i + (size_t)n * j], X[i + (size_t)n * q], h);
                atomicAdd(Hat(j, q), robust ? d1 * h + (2.0 * d2 * gs[j]) * gs[q] : h); }
        }
    } else if (kind == NLLS_COST_DYN_LINEAR) {                                   // non-squared cost y'w: value, gradient y, Hessian 0
        const double* y = data + k * (int64_t)ndata;
        double acc = 0;
        for (int i = threadIdx.x; i < n; i += TPB) acc += y[i] * w[i];
        cost = block_sum(acc, red);
        if (bo != DEST_NONE) for (int i = threadIdx.x; i < n; i += TPB) atomicAdd(&b[bo + i], y[i]);
    } else {                                                                      // NormResidual w: J = I
        double acc = 0;
        for (int i = threadIdx.x; i < n; i += TPB) acc += w[i] * w[i];
        { const double t = block_sum(acc, red); if (threadIdx.x == 0) total = t; }
        __syncthreads();
        double rho, d1, d2; kernel_at(total, rho, d1, d2);
        cost = 0.5 * rho;
        if (bo != DEST_NONE) {
            for (int i = threadIdx.x; i < n; i += TPB) { atomicAdd(&b[bo + i], d1 * w[i]); if (!robust || d2 == 0.0) atomicAdd(Hat(i, i), d1); }
            if (robust && d2 != 0.0)                                                // H = rho' I + 2 rho'' w w'
                for (int64_t e = threadIdx.x; e < (int64_t)n * n; e += TPB) { const int i = (int)(e % n), j = (int)(e / n); if (i >= j || blk) atomicAdd(Hat(i, j), (i == j ? d1 : 0.0) + (2.0 * d2 * w[i]) * w[j]); }
        }
    }
    if (threadIdx.x == 0) partials[blockIdx.x] = cost;
}

// ================================================================================================
// vector helpers
// ================================================================================================
// update!(to, from, linsystem)   src/linearsystem.jl:206-213
__global__ void retract_kernel(const int32_t* __restrict__ kind, const int32_t* __restrict__ dim, const uint32_t* __restrict__ voff,
                               const uint32_t* __restrict__ vboff, int64_t nvar, const double* __restrict__ from,
                               const double* __restrict__ x, double* __restrict__ to) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nvar) retract_one(kind, dim, voff, vboff, i, from, x, to);
}
// maximum(abs, x) and x'x   (src/optimize.jl:149, src/iterators.jl:160; NaN propagates like Julia's maximum).
// Two stages: per-workgroup partials (max, nan flag, sum of squares), then one small finishing workgroup.
constexpr int RED_BLOCKS = 256;
__global__ __launch_bounds__(TPB) void step_stats_partial_kernel(const double* __restrict__ x, int64_t n, double* __restrict__ part) {
    __shared__ double red[TPB / 64];
    double m = 0, s = 0; bool nan = false;
    for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < n; i += (int64_t)gridDim.x * TPB) { double v = x[i]; nan |= is_nan_bits(v); m = fmax(m, fabs(v)); s += v * v; }
    double mm = block_max(m, red);
    double ss = block_sum(s, red);
    double nn = block_max(nan ? 1.0 : 0.0, red);
    if (threadIdx.x == 0) { part[3 * blockIdx.x] = mm; part[3 * blockIdx.x + 1] = nn; part[3 * blockIdx.x + 2] = ss; }
}
__global__ __launch_bounds__(TPB) void step_stats_finish_kernel(const double* __restrict__ part, int nb, double* __restrict__ out) {
    __shared__ double red[TPB / 64];
    double m = 0, s = 0, nn = 0;
    for (int i = threadIdx.x; i < nb; i += TPB) { m = fmax(m, part[3 * i]); nn = fmax(nn, part[3 * i + 1]); s += part[3 * i + 2]; }
    double mm = block_max(m, red); double ss = block_sum(s, red); double n2 = block_max(nn, red);
    if (threadIdx.x == 0) { out[1] = n2 > 0 ? __longlong_as_double(0x7ff8000000000000LL) : mm; out[2] = ss; }
}
// initlambda's max |H_ii|   src/iterators.jl:131-137
__global__ __launch_bounds__(TPB) void max_abs_diag_partial_kernel(const double* __restrict__ A, const int64_t* __restrict__ diag_off,
                                                                   const int32_t* __restrict__ bs, const uint8_t* __restrict__ rowmask, int64_t nb, int64_t ld_dense, double* __restrict__ part) {
    __shared__ double red[TPB / 64];
    double m = 0; bool nan = false;
    for (int64_t k = (int64_t)blockIdx.x * TPB + threadIdx.x; k < nb; k += (int64_t)gridDim.x * TPB) {
        const int n = bs[k]; const int64_t o = diag_off[k]; const int64_t ld = ld_dense ? ld_dense : n;
        if (o < 0 || (rowmask && !rowmask[k])) continue;
        for (int i = 0; i < n; ++i) { const double a = A[o + i + ld * i]; nan |= is_nan_bits(a); m = fmax(m, fabs(a)); }
    }
    double mm = block_max(m, red);
    double nn = block_max(nan ? 1.0 : 0.0, red);             // Julia's max propagates NaN (src/iterators.jl:131-137): so does this
    if (threadIdx.x == 0) part[blockIdx.x] = nn > 0 ? __longlong_as_double(0x7ff8000000000000LL) : mm;
}
__global__ __launch_bounds__(TPB) void max_finish_kernel(const double* __restrict__ part, int nb, double* __restrict__ out, int slot) {
    __shared__ double red[TPB / 64];
    double m = 0;
    bool nan = false;
    for (int i = threadIdx.x; i < nb; i += TPB) { nan |= is_nan_bits(part[i]); m = fmax(m, part[i]); }
    double mm = block_max(m, red);
    double nn = block_max(nan ? 1.0 : 0.0, red);
    if (threadIdx.x == 0) out[slot] = nn > 0 ? __longlong_as_double(0x7ff8000000000000LL) : mm;
}

// ================================================================================================
// optimizesingles!(problem, options, indices)   src/optimize.jl:60-76,183-205
// Every listed variable is optimised on its own -- all other variables fixed -- against the cost blocks that depend on
// it: the whole outer loop (src/optimize.jl:109-180) with the Levenberg-Marquardt iterator (src/iterators.jl:139-172) and
// the univariate linear system (src/linearsystem.jl:12-32,126-130) runs in ONE thread per variable -- or with the Newton, dogleg or
// gradient-descent iterator (src/iterators.jl:15-27,47-115,187-208); the subproblems of one launch are independent (no cost block
// holds two of them: listed variables that share a block are relaxed one after the other, in launches of independent sets).
// ================================================================================================
struct SinglesGroup { int kind; const double* data; const uint32_t* voff; RobustSpec rk; };
struct SinglesOpt { int maxiters, maxfails; double reldcost, absdcost, dstep; int iterator; };   // iterator: 0 Newton, 1 Levenberg-Marquardt, 2 dogleg, 3 gradient descent
constexpr int SGL_MAXD = 6;                                   // dof of a variable optimised this way (registry maximum)

// one cost block of the subproblem: the variable's storage comes from `vloc`, everything else from `vars`
template <int KIND>
NLLS_DEV void singles_block(const SinglesGroup& G, uint32_t k, int slot, const double* __restrict__ vars, const double* vloc,
                            bool want_gh, double& cost, double* g, double* H) {
    using R = Res<KIND>; using I = ResInfo<KIND>;
    double d[R::NDATA]; uint32_t vo[R::NDEPS];
#pragma unroll
    for (int q = 0; q < R::NDATA; ++q) d[q] = G.data[(size_t)k * R::NDATA + q];
#pragma unroll
    for (int q = 0; q < R::NDEPS; ++q) vo[q] = G.voff[(size_t)k * R::NDEPS + q];
    double st[R::NDEPS][MAXST];
    BlockGH<KIND>::load(vars, vo, st);
    static_for<R::NDEPS>([&](auto Sc) {
        constexpr int S = decltype(Sc)::value;
        if (S == slot) {
#pragma unroll
            for (int q = 0; q < I::sto(S); ++q) st[S][q] = vloc[q];
        }
    });
    BlockGH<KIND> B; B.compute_st(st, d, G.rk, false);        // an adaptive kernel variable stays fixed (it is not the one being optimised)
    cost += B.cost;
    if (!want_gh) return;
    static_for<R::NDEPS>([&](auto Sc) {
        constexpr int S = decltype(Sc)::value;
        if (S == slot) {
            constexpr int DS = I::dof(S);
#pragma unroll
            for (int j = 0; j < DS; ++j) {
                g[j] += g_elem<KIND, S>(B, j);
#pragma unroll
                for (int i = 0; i < DS; ++i) H[i + SGL_MAXD * j] += h_elem<KIND, S, S>(B, i, j);
            }
        }
    });
}

__global__ __launch_bounds__(64) void singles_lm_kernel(const SinglesGroup* __restrict__ groups, const int64_t* __restrict__ selvar, int64_t nsel,
                                                        const int64_t* __restrict__ cptr, const int32_t* __restrict__ cgroup, const uint32_t* __restrict__ cidx,
                                                        const int32_t* __restrict__ cslot, const int32_t* __restrict__ vkind, const int32_t* __restrict__ vdim,
                                                        const uint32_t* __restrict__ voffs, SinglesOpt opt, double* __restrict__ vars, int64_t* __restrict__ iters_out) {
    const int64_t t = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (t >= nsel) return;
    const int64_t v = selvar[t];
    const int kind = vkind[v], dim = vdim[v], nd = var_dof(kind, dim), ns = var_storage(kind, dim);
    const uint32_t off = voffs[v];
    double vcur[MAXST], vnext[MAXST], vbest[MAXST];
    for (int q = 0; q < ns; ++q) { vcur[q] = vars[off + q]; vbest[q] = vcur[q]; }
    double g[SGL_MAXD], H[SGL_MAXD * SGL_MAXD];
    auto evaluate = [&](const double* vloc, bool want_gh) {
        double cost = 0;
        if (want_gh) { for (int i = 0; i < SGL_MAXD; ++i) g[i] = 0; for (int i = 0; i < SGL_MAXD * SGL_MAXD; ++i) H[i] = 0; }
        for (int64_t e = cptr[t]; e < cptr[t + 1]; ++e) {
            const SinglesGroup G = groups[cgroup[e]];
            switch (G.kind) {
#define X(K) case K: singles_block<K>(G, cidx[e], cslot[e], vars, vloc, want_gh, cost, g, H); break;
                NLLS_FOR_EACH_RES(X)
#undef X
            }
        }
        return cost;
    };
    double bestcost = evaluate(vcur, true), cost = bestcost;   // src/optimize.jl:118
    double lambda = 0.0, trust = 0.0, stepsize = 1.0;          // reset!(iteratedata): every subproblem starts from the iterator's initial state
    int fails = 0, iter = 0;
    double x[SGL_MAXD];
    // (H + lam I) x = -g by LDL' (the univariate system; src/linearsolver.jl:20-32)
    auto solve = [&](double lam) {
        double Ld[SGL_MAXD * SGL_MAXD], y[SGL_MAXD];
        for (int j = 0; j < nd; ++j) for (int i = j; i < nd; ++i) Ld[i + SGL_MAXD * j] = H[i + SGL_MAXD * j] + (i == j ? lam : 0.0);
        for (int j = 0; j < nd; ++j) {
            double dj = Ld[j + SGL_MAXD * j];
            for (int k = 0; k < j; ++k) dj -= Ld[j + SGL_MAXD * k] * Ld[j + SGL_MAXD * k] * Ld[k + SGL_MAXD * k];
            Ld[j + SGL_MAXD * j] = dj;
            for (int i = j + 1; i < nd; ++i) { double s2 = Ld[i + SGL_MAXD * j];
                for (int k = 0; k < j; ++k) s2 -= Ld[i + SGL_MAXD * k] * Ld[j + SGL_MAXD * k] * Ld[k + SGL_MAXD * k];
                Ld[i + SGL_MAXD * j] = s2 / dj; }
        }
        for (int i = 0; i < nd; ++i) { double s2 = g[i]; for (int k = 0; k < i; ++k) s2 -= Ld[i + SGL_MAXD * k] * y[k]; y[i] = s2; }
        for (int i = 0; i < nd; ++i) y[i] /= Ld[i + SGL_MAXD * i];
        for (int i = nd - 1; i >= 0; --i) { double s2 = y[i]; for (int k = i + 1; k < nd; ++k) s2 -= Ld[k + SGL_MAXD * i] * y[k]; y[i] = s2; }
        for (int i = 0; i < nd; ++i) x[i] = -y[i];                             // negate!
    };
    auto maxabs_x = [&]() { double m = 0; for (int i = 0; i < nd; ++i) m = is_nan_bits(x[i]) ? x[i] : (is_nan_bits(m) ? m : fmax(m, fabs(x[i]))); return m; };
    auto trial = [&]() { var_update_real(kind, dim, vcur, x, vnext); return evaluate(vnext, false); };   // update! + cost(varnext)
    auto quad = [&](const double* u) { double q = 0; for (int j = 0; j < nd; ++j) for (int i = 0; i < nd; ++i) q += u[i] * H[i + SGL_MAXD * j] * u[j]; return q; };   // fast_bAb
    while (true) {
        ++iter;
        double maxstep = 0;
        if (opt.iterator == 0) {
            // ---- iterate!(NewtonData)   src/iterators.jl:15-27
            solve(0.0); cost = trial();
        } else if (opt.iterator == 1) {
            // ---- iterate!(LevMarData)   src/iterators.jl:139-172
            if (lambda == 0.0) { double m = 0; for (int i = 0; i < nd; ++i) m = fmax(m, fabs(H[i + SGL_MAXD * i])); lambda = m * 1e-6; }
            double mu = 2.0;
            while (true) {
                solve(lambda);                                                     // :149-153
                const double cost_ = trial();                                      // :155-157
                if (!(cost_ > bestcost) || maxabs_x() < opt.dstep) {               // :160
                    double gx = 0;                                                 // fast_bAb(H, x), dot(g, x) on the undamped H  :162-163
                    for (int j = 0; j < nd; ++j) gx += g[j] * x[j];
                    const double q = (cost_ - bestcost) / (0.5 * quad(x) + gx);
                    lambda *= q < 0.983 ? 1.0 - (2.0 * q - 1.0) * (2.0 * q - 1.0) * (2.0 * q - 1.0) : 0.1;   // :164
                    cost = cost_;
                    break;
                }
                lambda *= mu; mu *= 2.0;                                           // :169-170
                if (!(lambda < 1e300)) { cost = cost_; break; }                    // (a block that never improves: leave instead of spinning)
            }
        } else if (opt.iterator == 2) {
            // ---- iterate!(DoglegData)   src/iterators.jl:47-115
            double gnorm2 = 0; for (int i = 0; i < nd; ++i) gnorm2 += g[i] * g[i];
            const double a = gnorm2 / (quad(g) + 2.2250738585072014e-308);         // floatmin
            double cauchy[SGL_MAXD]; for (int i = 0; i < nd; ++i) cauchy[i] = -a * g[i];
            const double alpha2 = a * a * gnorm2, alpha = sqrt(alpha2); double beta = 0;
            if (trust == 0.0) trust = alpha;                                       // first step: the Cauchy point
            if (alpha < trust) { solve(0.0); double s2 = 0; for (int i = 0; i < nd; ++i) s2 += x[i] * x[i]; beta = sqrt(s2); }
            double cost_ = bestcost; int spins = 0;
            while (true) {
                double linear_approx;
                if (!(alpha < trust)) { for (int i = 0; i < nd; ++i) x[i] = (trust / alpha) * cauchy[i]; linear_approx = trust * (2 * alpha - trust) / (2 * a); }
                else if (beta <= trust) linear_approx = cost_;
                else {
                    double sq_leg = 0, c = 0;
                    for (int i = 0; i < nd; ++i) { x[i] -= cauchy[i]; sq_leg += x[i] * x[i]; c += cauchy[i] * x[i]; }
                    const double trsq = trust * trust - alpha2; double step = sqrt(c * c + sq_leg * trsq);
                    step = c <= 0 ? (-c + step) / sq_leg : trsq / (c + step);
                    for (int i = 0; i < nd; ++i) x[i] = x[i] * step + cauchy[i];
                    linear_approx = 0.5 * (a * (1 - step) * (1 - step) * gnorm2) + step * (2 - step) * cost_;
                }
                cost_ = trial();
                const double mu = (bestcost - cost_) / linear_approx;
                if (mu > 0.375) { double s2 = 0; for (int i = 0; i < nd; ++i) s2 += x[i] * x[i]; trust = fmax(trust, 3 * sqrt(s2)); }
                else if (mu < 0.125) trust *= 0.5;
                if (!(cost_ > bestcost) || maxabs_x() < opt.dstep || ++spins > 2000) break;
            }
            cost = cost_;
        } else {
            // ---- iterate!(GradientDescentData)   src/iterators.jl:187-208
            for (int i = 0; i < nd; ++i) x[i] = -g[i] * stepsize;
            double costc = trial(); int spins = 0;
            while (costc > bestcost && ++spins <= 2000) {
                double coststep = 0; for (int i = 0; i < nd; ++i) coststep += x[i] * g[i];
                const double costdiff = bestcost + coststep - costc;
                stepsize *= 0.5 * coststep / costdiff;
                for (int i = 0; i < nd; ++i) x[i] = -g[i] * stepsize;
                costc = trial();
            }
            stepsize *= 2; cost = costc;
        }
        maxstep = maxabs_x();
        // ---- src/optimize.jl:128-160
        double dcost = bestcost - cost;
        if (dcost >= 0) { bestcost = cost; fails = 0; }
        else { dcost = cost; ++fails; if (fails == 1) for (int q = 0; q < ns; ++q) vbest[q] = vcur[q]; }
        for (int q = 0; q < ns; ++q) vcur[q] = vnext[q];                       // updatefromnext!
        int conv = 0;
        conv |= (fabs(cost) == INFINITY) << 0; conv |= (int)is_nan_bits(cost) << 1;
        conv |= (dcost < bestcost * opt.reldcost) << 2; conv |= (dcost < opt.absdcost) << 3;
        conv |= (fabs(maxstep) == INFINITY) << 4; conv |= (int)is_nan_bits(maxstep) << 5;
        conv |= (maxstep < opt.dstep) << 6; conv |= (fails > opt.maxfails) << 7; conv |= (iter >= opt.maxiters) << 8;
        if (conv) break;
        evaluate(vcur, true);                                                  // :167-170
    }
    if (!(bestcost >= cost)) for (int q = 0; q < ns; ++q) vcur[q] = vbest[q];  // updatefrombest!  :173-176
    for (int q = 0; q < ns; ++q) vars[off + q] = vcur[q];
    iters_out[t] = iter;
}

// ================================================================================================
// host-side enqueue
// ================================================================================================
static int herr(nlls_ctx* c, hipError_t e, const char* what) { c->err = std::string(what) + ": " + hipGetErrorString(e); return NLLS_ERR_HIP; }
#define HIPCHK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return herr(c, e_, #expr); } while (0)

template <int KIND>
static int launch_cost(nlls_ctx* c, const Group& G, const double* vars, int64_t& pbase, const PostSolveArgs* post, bool* taken) {
    if (G.ncost > 0) {
        static const int cost_grid_max = [] { const char* e = getenv("NLLS_COST_GRID_MAX"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 2048; }();   // (A/B: workgroups of the cost sweep)
        int grid = (int)std::min<int64_t>((G.ncost + TPB - 1) / TPB, cost_grid_max);
        const bool shared = G.cost_list >= 0 && c->info.is_sparse;
        const double* data = shared ? G.lists[G.cost_list].data.p : G.data.p; const uint32_t* voff = shared ? G.lists[G.cost_list].voff.p : G.voff.p;
        // (matrix-free LM trial: the blocks in elimination order are what the loop's other two launches stream -- the cost sweep reads the same 24 bytes per block)
        if (c->mf_ok && c->mf_on && G.mf_data.p && G.mf_voff.p) { data = G.mf_data.p; voff = G.mf_voff.p; }
        if (post && taken && !*taken) {      // the first launch of the sweep takes the statistics roles along
            hipLaunchKernelGGL((cost_kernel<KIND, true>), dim3(grid + post->np + post->np3 + post->np2), dim3(TPB), 0, c->stream, vars, data, voff, (const uint32_t*)nullptr, G.ncost, G.rk, c->partials.p + pbase, grid, *post);
            *taken = true;
        } else hipLaunchKernelGGL((cost_kernel<KIND, false>), dim3(grid), dim3(TPB), 0, c->stream, vars, data, voff, (const uint32_t*)nullptr, G.ncost, G.rk, c->partials.p + pbase, grid, PostSolveArgs{});
        pbase += grid;
    }
    return NLLS_OK;
}
template <int KIND>
static int launch_fixedcost(nlls_ctx* c, const Group& G, const double* vars, int64_t& pbase) {
    if (G.nfixedcost > 0) {
        int grid = (int)std::min<int64_t>((G.nfixedcost + TPB - 1) / TPB, 2048);
        hipLaunchKernelGGL((cost_kernel<KIND, false>), dim3(grid), dim3(TPB), 0, c->stream, vars, G.data.p, G.voff.p, G.fixedcost.p, G.nfixedcost, G.rk, c->partials.p + pbase, grid, PostSolveArgs{});
        pbase += grid;
    }
    return NLLS_OK;
}
static int dyn_n_of(const Group& G) { return G.res_kind == NLLS_RES_DYN_LINEAR ? G.ndata - 1 : G.res_kind == NLLS_COST_DYN_LINEAR ? G.ndata : G.nres; }
static int launch_dyn_cost(nlls_ctx* c, const Group& G, const double* vars, int64_t& pbase, bool fixed_only) {
    const int64_t nb = fixed_only ? G.nfixedcost : G.ncost;
    if (nb > 0) {
        hipLaunchKernelGGL(dyn_block_kernel, dim3((unsigned)nb), dim3(TPB), 0, c->stream, G.res_kind, dyn_n_of(G), G.ndata, vars, G.data.p, G.voff.p,
                           fixed_only ? G.fixedcost.p : (const uint32_t*)nullptr, (const uint32_t*)nullptr, 0, (double*)nullptr, (double*)nullptr, c->partials.p + pbase, G.rk);
        pbase += nb;
    }
    return NLLS_OK;
}
int enqueue_dyn_gradhess(nlls_ctx* c, const Group& G, const double* vars, int64_t& pbase) {
    if (G.dense.n > 0) {
        hipLaunchKernelGGL(dyn_block_kernel, dim3((unsigned)G.dense.n), dim3(TPB), 0, c->stream, G.res_kind, dyn_n_of(G), G.ndata, vars, G.dense.data.p, G.dense.voff.p,
                           (const uint32_t*)nullptr, G.dense.brow.p, (int)c->info.ndof, c->A.p, c->b.p, c->partials.p + pbase, G.rk, (const uint32_t*)(c->info.is_sparse ? G.dense.aoff.p : nullptr));
        pbase += G.dense.n;
    }
    HIPCHK(hipGetLastError());
    return NLLS_OK;
}
// cost blocks whose variables are all fixed: cost only (called from the gradient sweep as well)
int enqueue_fixedcost(nlls_ctx* c, const Group& G, const double* vars, int64_t& pbase) {
    if (is_dyn_kind(G.res_kind)) return launch_dyn_cost(c, G, vars, pbase, true);
    switch (G.res_kind) {
#define X(K) case K: return launch_fixedcost<K>(c, G, vars, pbase);
        NLLS_FOR_EACH_RES(X)
#undef X
    }
    return NLLS_OK;
}
// ---- closed forms against dual numbers (nlls_check_analytic) --------------------------------------------------------------------------------------
// Every block of a group evaluated twice -- BlockGH<KIND, true> (a residual kind's own `jac`, the adaptive kernel's closed-form derivatives) and
// BlockGH<KIND, false> (everything through Dual<N> / Dual2: src/autodiff.jl:81-93,164-165 as written) -- and the largest difference of each quantity,
// relative to the largest magnitude of that quantity in the block: out[0] J, [1] J'r, [2] cost, [3] rho', [4] rho'', [5] d rho / d kernel, [6] d2 rho / d kernel d(kernel, cost).
template <int KIND>
__global__ __launch_bounds__(TPB) void check_analytic_kernel(const double* __restrict__ vars, const double* __restrict__ data, const uint32_t* __restrict__ voff, int64_t n,
                                                             RobustSpec rk, int kernel_free, double* __restrict__ out) {
    using R = Res<KIND>; using BA = BlockGH<KIND, true>; using BD = BlockGH<KIND, false>;
    __shared__ double red[TPB / 64];
    double m[7] = {0, 0, 0, 0, 0, 0, 0};
    if constexpr (!is_cost_kind<KIND>) {
        for (int64_t e = (int64_t)blockIdx.x * TPB + threadIdx.x; e < n; e += (int64_t)gridDim.x * TPB) {
            double d[R::NDATA]; uint32_t vo[R::NDEPS];
#pragma unroll
            for (int q = 0; q < R::NDATA; ++q) d[q] = data[(size_t)e * R::NDATA + q];
#pragma unroll
            for (int q = 0; q < R::NDEPS; ++q) vo[q] = voff[(size_t)e * R::NDEPS + q];
            BA a; a.compute(vars, vo, d, rk, kernel_free != 0);
            BD b; b.compute(vars, vo, d, rk, kernel_free != 0);
            auto upd = [](double& mx, double diff, double scale) { const double r = fabs(diff) / fmax(scale, 1e-300); if (!(r <= mx)) mx = r; };
            double sj = 0, sg = 0;
            for (int q = 0; q < BA::M; ++q) for (int j = 0; j < BA::NP; ++j) sj = fmax(sj, fabs(b.J[q][j]));
            for (int j = 0; j < BA::NP; ++j) sg = fmax(sg, fabs(b.g[j]));
            for (int q = 0; q < BA::M; ++q) for (int j = 0; j < BA::NP; ++j) upd(m[0], a.J[q][j] - b.J[q][j], sj);
            for (int j = 0; j < BA::NP; ++j) upd(m[1], a.g[j] - b.g[j], sg);
            upd(m[2], a.cost - b.cost, fabs(b.cost)); upd(m[3], a.dc - b.dc, fabs(b.dc)); upd(m[4], a.d2c - b.d2c, fmax(fabs(b.d2c), fabs(b.dc)));
            if constexpr (R::ADAPT) if (kernel_free) {
                double s1 = 0, s2 = 0;
                for (int i = 0; i < 3; ++i) { s1 = fmax(s1, fabs(b.dck[i])); for (int j = 0; j < 4; ++j) s2 = fmax(s2, fabs(b.d2ck[i][j])); }
                for (int i = 0; i < 3; ++i) { upd(m[5], a.dck[i] - b.dck[i], s1); for (int j = 0; j < 4; ++j) upd(m[6], a.d2ck[i][j] - b.d2ck[i][j], s2); }
            }
        }
    }
    for (int q = 0; q < 7; ++q) { const double t = block_max(m[q], red); if (threadIdx.x == 0) out[(size_t)blockIdx.x * 8 + q] = t; __syncthreads(); }
}
int enqueue_check_analytic(nlls_ctx* c, double* d_out /* [nblocks_total][8] */, int64_t* nblocks_out) {
    const double* vars = vars_ptr(c, NLLS_VARS_CURRENT); int64_t base = 0;
    for (const Group& G : c->groups) {
        if (is_dyn_kind(G.res_kind) || G.ncost == 0) continue;
        const int grid = (int)std::min<int64_t>((G.ncost + TPB - 1) / TPB, 256);
        if (d_out) switch (G.res_kind) {
#define X(K) case K: hipLaunchKernelGGL(check_analytic_kernel<K>, dim3(grid), dim3(TPB), 0, c->stream, vars, G.data.p, G.voff.p, G.ncost, G.rk, 1, d_out + base * 8); break;
            NLLS_FOR_EACH_RES(X)
#undef X
        }
        base += grid;
    }
    if (nblocks_out) *nblocks_out = base;
    return hipGetLastError() == hipSuccess ? NLLS_OK : NLLS_ERR_HIP;
}

int enqueue_reduce_partials(nlls_ctx* c, int64_t n) {
    hipLaunchKernelGGL(reduce_partials_kernel, dim3(1), dim3(TPB), 0, c->stream, c->partials.p, n, c->scalars.p);
    HIPCHK(hipGetLastError());
    return NLLS_OK;
}

// pofs / count: an LM trial leaves the partials un-reduced at partials + pofs (count of them in *count) for enqueue_trial_finish
int enqueue_sweep_cost(nlls_ctx* c, int which, int64_t pofs, int64_t* count, const PostSolveArgs* post, bool* post_taken) {
    const double* vars = vars_ptr(c, which); int64_t pbase = pofs;
    for (const Group& G : c->groups) {
        if (is_dyn_kind(G.res_kind)) { launch_dyn_cost(c, G, vars, pbase, false); continue; }
        switch (G.res_kind) {
#define X(K) case K: launch_cost<K>(c, G, vars, pbase, post, post_taken); break;
            NLLS_FOR_EACH_RES(X)
#undef X
        }
    }
    if (count) *count = pbase - pofs;
    else hipLaunchKernelGGL(reduce_partials_kernel, dim3(1), dim3(TPB), 0, c->stream, c->partials.p + pofs, pbase - pofs, c->scalars.p);
    HIPCHK(hipGetLastError());
    return NLLS_OK;
}

int enqueue_retract(nlls_ctx* c, int to, int from) {
    const int64_t nvar = c->info.nvar; if (nvar == 0) return NLLS_OK;
    hipLaunchKernelGGL(retract_kernel, dim3((unsigned)((nvar + 255) / 256)), dim3(256), 0, c->stream, c->d_var_kind.p, c->d_var_dim.p, c->d_var_off.p,
                       c->d_var_boff.p, nvar, vars_ptr(c, from), c->x.p, vars_ptr(c, to));
    HIPCHK(hipGetLastError());
    return NLLS_OK;
}
int enqueue_step_stats(nlls_ctx* c) {
    const int nb = (int)std::max<int64_t>(1, std::min<int64_t>((c->info.ndof + TPB - 1) / TPB, RED_BLOCKS));
    hipLaunchKernelGGL(step_stats_partial_kernel, dim3(nb), dim3(TPB), 0, c->stream, c->x.p, c->info.ndof, c->partials.p);
    hipLaunchKernelGGL(step_stats_finish_kernel, dim3(1), dim3(TPB), 0, c->stream, c->partials.p, nb, c->scalars.p);
    HIPCHK(hipGetLastError());
    return NLLS_OK;
}
int enqueue_max_abs_diag(nlls_ctx* c) {
    const int nb = (int)std::max<int64_t>(1, std::min<int64_t>((c->info.nblocks + TPB - 1) / TPB, RED_BLOCKS));
    hipLaunchKernelGGL(max_abs_diag_partial_kernel, dim3(nb), dim3(TPB), 0, c->stream, c->A.p, c->d_diag_off.p, c->d_blocksizes.p,
                       c->nranks > 1 ? c->d_row_mask.p : (const uint8_t*)nullptr, c->info.nblocks, c->info.is_sparse ? (int64_t)0 : c->info.ndof, c->partials.p);
    hipLaunchKernelGGL(max_finish_kernel, dim3(1), dim3(TPB), 0, c->stream, c->partials.p, nb, c->scalars.p, 3);
    HIPCHK(hipGetLastError());
    return NLLS_OK;
}

// optimizesingles!: the launch.  The caller has validated the lists (nlls_optimize_singles).
// optimizesingles! under sharding: the storage of the listed variables, between the variable set and a dense buffer of the same layout (the ranks' results are
// gathered by a sum over buffers that are zero where another rank did the work)
__global__ void copy_var_storage_kernel(const int64_t* __restrict__ sel, int64_t nsel, const uint32_t* __restrict__ var_off, const double* __restrict__ src, double* __restrict__ dst) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; if (i >= nsel) return;
    const int64_t v = sel[i];
    for (uint32_t q = var_off[v]; q < var_off[v + 1]; ++q) dst[q] = src[q];
}
__global__ void iters_to_double_kernel(const int64_t* __restrict__ it, const int64_t* __restrict__ pos, int64_t n, double* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; if (i < n) out[pos[i]] = (double)it[i];
}
int enqueue_copy_var_storage(nlls_ctx* c, const int64_t* d_sel, int64_t nsel, const double* src, double* dst) {
    if (nsel > 0) hipLaunchKernelGGL(copy_var_storage_kernel, dim3((unsigned)((nsel + 255) / 256)), dim3(256), 0, c->stream, d_sel, nsel, c->d_var_off.p, src, dst);
    return hipGetLastError() == hipSuccess ? NLLS_OK : NLLS_ERR_HIP;
}
int enqueue_iters_to_double(nlls_ctx* c, const int64_t* d_it, const int64_t* d_pos, int64_t n, double* d_out) {
    if (n > 0) hipLaunchKernelGGL(iters_to_double_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, d_it, d_pos, n, d_out);
    return hipGetLastError() == hipSuccess ? NLLS_OK : NLLS_ERR_HIP;
}
int enqueue_optimize_singles(nlls_ctx* c, int64_t nsel, const int64_t* d_selvar, const int64_t* d_cptr, const int32_t* d_cgroup, const uint32_t* d_cidx,
                             const int32_t* d_cslot, const void* d_groups, int iterator, int maxiters, int maxfails, double reldcost, double absdcost, double dstep, int64_t* d_iters) {
    if (nsel <= 0) return NLLS_OK;
    SinglesOpt o{maxiters, maxfails, reldcost, absdcost, dstep, iterator};
    hipLaunchKernelGGL(singles_lm_kernel, dim3((unsigned)((nsel + 63) / 64)), dim3(64), 0, c->stream, (const SinglesGroup*)d_groups, d_selvar, nsel, d_cptr, d_cgroup, d_cidx, d_cslot,
                       c->d_var_kind.p, c->d_var_dim.p, c->d_var_off.p, o, vars_ptr(c, NLLS_VARS_CURRENT), d_iters);
    HIPCHK(hipGetLastError());
    return NLLS_OK;
}
size_t singles_group_size() { return sizeof(SinglesGroup); }
void singles_group_fill(void* dst, const Group& G) { SinglesGroup g{G.res_kind, G.data.p, G.voff.p, G.rk}; memcpy(dst, &g, sizeof(g)); }

}  // namespace nlls
