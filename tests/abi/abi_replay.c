/* abi_replay.c -- pins, from OUTSIDE Python, what the Julia shim (nllssolver.jl_amd/julia/NLLSsolverAMD.jl) assumes about
 * libnlls_amd.so: the struct layouts it mirrors by hand and the exact ccall sequence / argument types of its device-resident
 * Levenberg-Marquardt loop (optimizeinternal! for NLLSInternal{MultiVariateLSgpu}, mirroring src/optimize.jl:109-180 and
 * src/iterators.jl:139-172).  Plain C (the header must stay C-clean); the library is dlopen()ed like `ccall((:sym, lib), ...)` does.
 *
 *   abi_replay --layout  <libnlls_amd.so>    struct layouts + every symbol the shim calls resolves     (no GPU needed)
 *   abi_replay --replay  <libnlls_amd.so>    the whole sequence on a small noise-free bundle adjustment (needs the GPU)
 */
#include <dlfcn.h>
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/nlls_amd.h"

/* ---- layouts the shim writes down by hand (struct CostGroup, struct NllsInfo in NLLSsolverAMD.jl) ---- */
_Static_assert(sizeof(nlls_cost_group) == 64, "CostGroup: 2 x Int32, NTuple{4,Float64}, Int64, 2 pointers");
_Static_assert(offsetof(nlls_cost_group, res_kind) == 0 && offsetof(nlls_cost_group, robust_kind) == 4, "CostGroup header");
_Static_assert(offsetof(nlls_cost_group, robust_params) == 8 && offsetof(nlls_cost_group, ncost) == 40, "CostGroup params / ncost");
_Static_assert(offsetof(nlls_cost_group, varind) == 48 && offsetof(nlls_cost_group, data) == 56, "CostGroup pointers");
_Static_assert(sizeof(nlls_info) == 112, "NllsInfo: 2 x Int32 + 13 x Int64");
_Static_assert(offsetof(nlls_info, is_sparse) == 0 && offsetof(nlls_info, has_schur) == 4 && offsetof(nlls_info, nvar) == 8, "NllsInfo head");
_Static_assert(offsetof(nlls_info, nblocks) == 16 && offsetof(nlls_info, ndof) == 24 && offsetof(nlls_info, nnz_data) == 32, "NllsInfo ndof");
_Static_assert(offsetof(nlls_info, var_storage) == 56 && offsetof(nlls_info, nreduced_dof) == 72 && offsetof(nlls_info, solve_mode) == 88, "NllsInfo tail");
_Static_assert(offsetof(nlls_info, nborder_dof) == 104, "NllsInfo last field");
/* struct LmOptions / mutable struct LmState of the shim (the library's own LM loop, nlls_lm_iterations) */
_Static_assert(sizeof(nlls_lm_options) == 48 && offsetof(nlls_lm_options, maxfails) == 24 && offsetof(nlls_lm_options, stoptime_ns) == 40, "LmOptions: 3 x Float64 + 3 x Int64");
_Static_assert(sizeof(nlls_lm_state) == 112 && offsetof(nlls_lm_state, timecost_ns) == 104 && offsetof(nlls_lm_state, iternum) == 24 && offsetof(nlls_lm_state, converged) == 48, "LmState head");
_Static_assert(offsetof(nlls_lm_state, linearsolvers) == 56 && offsetof(nlls_lm_state, singulartrials) == 80 && offsetof(nlls_lm_state, timegradient_ns) == 96, "LmState tail");

/* ---- the entry points the shim ccalls, with the argument types it passes ---- */
typedef int (*fn_ctx_create)(const int32_t*, int32_t, nlls_ctx**);
typedef int (*fn_ctx_destroy)(nlls_ctx*);
typedef const char* (*fn_last_error)(const nlls_ctx*);
typedef int (*fn_upload)(nlls_ctx*, int64_t, const int32_t*, const int32_t*, const uint64_t*, int32_t, const nlls_cost_group*, int32_t);
typedef int (*fn_get_info)(const nlls_ctx*, nlls_info*);
typedef int (*fn_bsm_index)(const nlls_ctx*, int64_t*, int64_t*, int64_t*, int64_t*);
typedef int (*fn_setget_vars)(nlls_ctx*, int32_t, double*);
typedef int (*fn_swapcopy)(nlls_ctx*, int32_t, int32_t);
typedef int (*fn_sweep_gradhess)(nlls_ctx*, double*);
typedef int (*fn_sweep_cost)(nlls_ctx*, int32_t, double*);
typedef int (*fn_out1)(nlls_ctx*, double*);
typedef int (*fn_damp)(nlls_ctx*, double);
typedef int (*fn_quadform)(nlls_ctx*, double*, double*);
typedef int (*fn_lm_trial)(nlls_ctx*, double, int32_t, int32_t, double*);
typedef int (*fn_retract)(nlls_ctx*, int32_t, int32_t);
typedef int (*fn_lm_iterations)(nlls_ctx*, const nlls_lm_options*, nlls_lm_state*, int64_t);

static const char* SYMS[] = {"nlls_ctx_create", "nlls_ctx_destroy", "nlls_last_error", "nlls_upload_structure", "nlls_get_info", "nlls_get_bsm_index",
    "nlls_set_variables", "nlls_get_variables", "nlls_swap_variables", "nlls_copy_variables", "nlls_sweep_gradhess", "nlls_sweep_cost", "nlls_get_grad",
    "nlls_max_abs_diag", "nlls_damp", "nlls_solve", "nlls_set_step", "nlls_get_step", "nlls_quadform", "nlls_step_maxabs", "nlls_retract", "nlls_lm_trial", "nlls_var_storage", "nlls_lm_iterations"};

static void* need(void* lib, const char* name) { void* p = dlsym(lib, name); if (!p) { fprintf(stderr, "missing symbol %s\n", name); exit(2); } return p; }
static uint64_t lcg_state = 88172645463325252ull;
static double urand(void) { lcg_state ^= lcg_state << 13; lcg_state ^= lcg_state >> 7; lcg_state ^= lcg_state << 17; return (double)(lcg_state >> 11) / 9007199254740992.0; }
static double nrand(void) { double u = urand() + 1e-300, v = urand(); return sqrt(-2.0 * log(u)) * cos(6.283185307179586 * v); }

int main(int argc, char** argv) {
    if (argc < 3) { fprintf(stderr, "usage: %s --layout|--replay <libnlls_amd.so>\n", argv[0]); return 2; }
    void* lib = dlopen(argv[2], RTLD_NOW | RTLD_LOCAL);
    if (!lib) { fprintf(stderr, "dlopen: %s\n", dlerror()); return 2; }
    for (size_t i = 0; i < sizeof(SYMS) / sizeof(SYMS[0]); ++i) need(lib, SYMS[i]);
    printf("layout ok: sizeof(nlls_cost_group)=%zu sizeof(nlls_info)=%zu offsetof(ndof)=%zu; %zu symbols resolved\n",
           sizeof(nlls_cost_group), sizeof(nlls_info), offsetof(nlls_info, ndof), sizeof(SYMS) / sizeof(SYMS[0]));
    if (strcmp(argv[1], "--replay") != 0) return 0;

    fn_ctx_create ctx_create = (fn_ctx_create)need(lib, "nlls_ctx_create"); fn_ctx_destroy ctx_destroy = (fn_ctx_destroy)need(lib, "nlls_ctx_destroy");
    fn_last_error last_error = (fn_last_error)need(lib, "nlls_last_error"); fn_upload upload = (fn_upload)need(lib, "nlls_upload_structure");
    fn_get_info get_info = (fn_get_info)need(lib, "nlls_get_info"); fn_bsm_index bsm_index = (fn_bsm_index)need(lib, "nlls_get_bsm_index");
    fn_setget_vars set_vars = (fn_setget_vars)need(lib, "nlls_set_variables"), get_vars = (fn_setget_vars)need(lib, "nlls_get_variables");
    fn_swapcopy swap_vars = (fn_swapcopy)need(lib, "nlls_swap_variables"), copy_vars = (fn_swapcopy)need(lib, "nlls_copy_variables");
    fn_sweep_gradhess sweep_gradhess = (fn_sweep_gradhess)need(lib, "nlls_sweep_gradhess"); fn_sweep_cost sweep_cost = (fn_sweep_cost)need(lib, "nlls_sweep_cost");
    fn_out1 max_abs_diag = (fn_out1)need(lib, "nlls_max_abs_diag"), step_maxabs = (fn_out1)need(lib, "nlls_step_maxabs");
    fn_damp damp = (fn_damp)need(lib, "nlls_damp"); fn_quadform quadform = (fn_quadform)need(lib, "nlls_quadform"); fn_lm_trial lm_trial = (fn_lm_trial)need(lib, "nlls_lm_trial");

    /* ---- a noise-free bundle adjustment in the shape of test/optimizeba.jl:6-47: every camera sees every point ---- */
    enum { NCAM = 6, NPT = 40, NVAR = NCAM + NPT, NOBS = NCAM * NPT };
    static double truth[NCAM * 6 + NPT * 3], start[NCAM * 6 + NPT * 3], meas[NOBS * 2]; static int64_t varind[NOBS * 2];
    for (int c = 0; c < NCAM; ++c) for (int k = 0; k < 6; ++k) truth[6 * c + k] = nrand() + ((k == 0 || k == 4) ? 1.0 : 0.0);
    for (int p = 0; p < NPT; ++p) { double* X = truth + 6 * NCAM + 3 * p; X[0] = urand() - 0.5; X[1] = urand() - 0.5; X[2] = urand() + 10.0; }
    for (int c = 0, o = 0; c < NCAM; ++c) for (int p = 0; p < NPT; ++p, ++o) {          /* camera-major, 1-based indices as stored */
        const double* P = truth + 6 * c; const double* X = truth + 6 * NCAM + 3 * p;
        varind[2 * o] = c + 1; varind[2 * o + 1] = NCAM + p + 1;
        meas[2 * o] = P[0] * X[0] + P[1] * X[1] + P[2] * X[2]; meas[2 * o + 1] = P[3] * X[0] + P[4] * X[1] + P[5] * X[2];
    }
    for (int i = 0; i < NCAM * 6 + NPT * 3; ++i) start[i] = truth[i] + 1e-3 * nrand();
    int32_t vk[NVAR], vd[NVAR]; uint64_t blockindices[NVAR];
    for (int i = 0; i < NVAR; ++i) { vk[i] = NLLS_VAR_EUCLIDEAN; vd[i] = i < NCAM ? 6 : 3; blockindices[i] = (uint64_t)(i + 1); }
    nlls_cost_group grp; memset(&grp, 0, sizeof grp);
    grp.res_kind = NLLS_RES_BA_AFFINE; grp.robust_kind = NLLS_ROBUST_NONE; grp.ncost = NOBS; grp.varind = varind; grp.data = meas;

#define CK(call) do { int rc_ = (call); if (rc_ != 0) { fprintf(stderr, "%s -> %d: %s\n", #call, rc_, ctx ? last_error(ctx) : ""); return 1; } } while (0)
    nlls_ctx* ctx = NULL;
    CK(ctx_create(NULL, 0, &ctx));                                                   /* makesymmvls_gpu */
    CK(upload(ctx, NVAR, vk, vd, blockindices, 1, &grp, 0));
    nlls_info info; CK(get_info(ctx, &info));
    int64_t boff[NVAR]; CK(bsm_index(ctx, NULL, NULL, NULL, boff));
    if (info.ndof != NCAM * 6 + NPT * 3 || info.var_storage != info.ndof || info.ncost != NOBS || boff[0] != 1 || boff[NCAM] != 6 * NCAM + 1) { fprintf(stderr, "nlls_info / boffsets do not read as the shim expects\n"); return 1; }
    /* ---- optimizeinternal!: src/optimize.jl:109-180 with the variables resident on the device ---- */
    double cost = 0, bestcost, startcost;
    CK(set_vars(ctx, NLLS_VARS_CURRENT, start)); CK(copy_vars(ctx, NLLS_VARS_NEXT, NLLS_VARS_CURRENT));
    CK(sweep_gradhess(ctx, &cost)); bestcost = startcost = cost;
    double lambda = 0, lastcost = cost; int fails = 0, iter = 0, converged = 0, trials = 0;
    while (!converged) {
        ++iter;
        /* iterate!(::LevMarData, ...): src/iterators.jl:139-172, one nlls_lm_trial per damped solve */
        if (lambda == 0) { double m; CK(max_abs_diag(ctx, &m)); lambda = 1e-6 * m; }
        double lastlambda = 0, mu = 2, c_ = 0, maxstep = 0;
        for (;;) {
            CK(lm_trial(ctx, lambda - lastlambda, NLLS_VARS_NEXT, NLLS_VARS_CURRENT, &c_)); lastlambda = lambda; ++trials;
            CK(step_maxabs(ctx, &maxstep));
            if (!(c_ > bestcost) || maxstep < 1e-15) {
                CK(damp(ctx, -lastlambda));
                double xHx, gx; CK(quadform(ctx, &xHx, &gx));
                const double q = (c_ - bestcost) / (0.5 * xHx + gx);
                lambda *= q < 0.983 ? 1 - (2 * q - 1) * (2 * q - 1) * (2 * q - 1) : 0.1;
                break;
            }
            lambda *= mu; mu *= 2;
        }
        double dcost = bestcost - c_; lastcost = c_;
        if (dcost >= 0) { bestcost = c_; fails = 0; }
        else { dcost = c_; if (++fails == 1) CK(copy_vars(ctx, NLLS_VARS_BEST, NLLS_VARS_CURRENT)); }      /* updatetobest! */
        CK(swap_vars(ctx, NLLS_VARS_CURRENT, NLLS_VARS_NEXT));                                             /* updatefromnext! */
        converged |= (isinf(c_) != 0) << 0; converged |= (isnan(c_) != 0) << 1;
        converged |= (dcost < bestcost * 1e-15) << 2; converged |= (dcost < 1e-15) << 3; converged |= (maxstep < 1e-15) << 6;
        converged |= (fails > 5) << 7; converged |= (iter >= 100) << 8;
        if (converged) break;
        CK(sweep_gradhess(ctx, NULL));                                               /* zero! + costgradhess!: enqueue only */
    }
    if (!(bestcost >= lastcost)) CK(swap_vars(ctx, NLLS_VARS_CURRENT, NLLS_VARS_BEST));                       /* updatefrombest!, src/optimize.jl:171-174 */
    static double final[NCAM * 6 + NPT * 3];
    CK(get_vars(ctx, NLLS_VARS_CURRENT, final));
    double check = 0; CK(sweep_cost(ctx, NLLS_VARS_CURRENT, &check));
    printf("replay: start %.6e -> best %.6e in %d iterations (%d LM trials), termination flags %d, cost(variables) %.6e\n", startcost, bestcost, iter, trials, converged, check);
    if (!(bestcost < 1e-15 * NOBS) || !(check < 1e-15 * NOBS)) { fprintf(stderr, "did not reach the zero-residual optimum (test/optimizeba.jl:62-75)\n"); return 1; }
    /* ---- the shim's path WITH a user callback (src/optimize.jl:128): after iterate! the trial point is fetched into problem.varnext (nlls_get_variables(ctx, 1)) and the
     *      step into linsystem.x (nlls_get_step), the callback runs, and what it left in varnext goes back down (nlls_set_variables(ctx, 1)) before updatefromnext! makes it the
     *      current point.  The "callback" here rewrites a variable (point 1, rounded to single precision, in the first three iterations) the way the reference's EM callback rewrites
     *      one (src/robustadaptive.jl:48-73): the edit must be bit for bit what the device holds as the current point after the swap, and the optimisation must go on from it.  ---- */
    {
        typedef int (*fn_get_step)(nlls_ctx*, double*); fn_get_step get_step = (fn_get_step)need(lib, "nlls_get_step");
        static double varnext[NCAM * 6 + NPT * 3], xstep[NCAM * 6 + NPT * 3], cur[NCAM * 6 + NPT * 3];
        CK(set_vars(ctx, NLLS_VARS_CURRENT, start)); CK(copy_vars(ctx, NLLS_VARS_NEXT, NLLS_VARS_CURRENT));
        CK(sweep_gradhess(ctx, &cost)); double best2 = cost, lam2 = 0; int it2 = 0, conv2 = 0, edits = 0;
        while (!conv2) {
            ++it2;
            if (lam2 == 0) { double m; CK(max_abs_diag(ctx, &m)); lam2 = 1e-6 * m; }
            double lastlambda = 0, mu = 2, c_ = 0, maxstep = 0;
            for (;;) {
                CK(lm_trial(ctx, lam2 - lastlambda, NLLS_VARS_NEXT, NLLS_VARS_CURRENT, &c_)); lastlambda = lam2;
                CK(step_maxabs(ctx, &maxstep));
                if (!(c_ > best2) || maxstep < 1e-15) { CK(damp(ctx, -lastlambda)); double xHx, gx; CK(quadform(ctx, &xHx, &gx));
                    const double q = (c_ - best2) / (0.5 * xHx + gx); lam2 *= q < 0.983 ? 1 - (2 * q - 1) * (2 * q - 1) * (2 * q - 1) : 0.1; break; }
                lam2 *= mu; mu *= 2;
            }
            CK(get_vars(ctx, NLLS_VARS_NEXT, varnext)); CK(get_step(ctx, xstep));                       /* fetchvariables!(problem.varnext, ...), linsystem.x */
            double edited[3] = {varnext[6 * NCAM], varnext[6 * NCAM + 1], varnext[6 * NCAM + 2]};
            if (it2 <= 3) { ++edits; for (int k = 0; k < 3; ++k) edited[k] = (double)(float)edited[k]; }   /* the callback edits problem.varnext: point 1 rounded to single precision */
            for (int k = 0; k < 3; ++k) varnext[6 * NCAM + k] = edited[k];
            CK(set_vars(ctx, NLLS_VARS_NEXT, varnext));                                                  /* ... and the edit goes back to the device */
            CK(sweep_cost(ctx, NLLS_VARS_NEXT, &c_));                                                    /* (the callback returns the cost of what it left there) */
            double dcost = best2 - c_; if (dcost >= 0) best2 = c_; else dcost = c_;
            CK(swap_vars(ctx, NLLS_VARS_CURRENT, NLLS_VARS_NEXT));
            CK(get_vars(ctx, NLLS_VARS_CURRENT, cur));
            for (int k = 0; k < 3; ++k) if (cur[6 * NCAM + k] != edited[k]) { fprintf(stderr, "callback path: the host's edit of varnext did not reach the device\n"); return 1; }
            conv2 |= (dcost < 1e-15) << 3; conv2 |= (maxstep < 1e-15) << 6; conv2 |= (it2 >= 60) << 8;
            if (conv2) break;
            CK(sweep_gradhess(ctx, NULL));
        }
        double check4 = 0; CK(sweep_cost(ctx, NLLS_VARS_CURRENT, &check4));
        printf("replay (callback path: varnext fetched, edited on the host, set again before updatefromnext!): best %.6e in %d iterations, %d edits, cost(variables) %.6e\n", best2, it2, edits, check4);
        if (!(check4 < 1e-15 * NOBS)) { fprintf(stderr, "the callback path did not reach the zero-residual optimum\n"); return 1; }
    }
    /* ---- the shim's OTHER iterators (Newton here; dogleg and gradient descent share the sequence): they run the reference's generic optimizeinternal! (src/optimize.jl:109-180), so the
     *      HOST vectors are the authority -- costgradhess!(linsystem, problem.variables, costs) uploads problem.variables every iteration (ls.resident == false), iterate! solves, forms
     *      problem.varnext = update(problem.variables, x) on the host and takes its cost from the device (gpucost: nlls_set_variables(ctx, 1) + nlls_sweep_cost(ctx, 1)), and
     *      updatefromnext! SWAPS THE HOST VECTORS (round 5: the shim's overload used to swap device slots only, the host vector never advanced and the loop returned its start point).
     *      The first two cameras are fixed (blockindices 0): that removes the affine gauge freedom, the undamped Newton system is definite.  Progress over >= 3 iterations is required. ---- */
    {
        typedef int (*fn_solve)(nlls_ctx*, double*); fn_solve solve = (fn_solve)need(lib, "nlls_solve");
        uint64_t bi2[NVAR]; for (int i = 0; i < NVAR; ++i) bi2[i] = i < 2 ? 0 : (uint64_t)(i - 1);
        nlls_ctx* c2 = NULL;
        if (ctx_create(NULL, 0, &c2) != 0 || upload(c2, NVAR, vk, vd, bi2, 1, &grp, 0) != 0) { fprintf(stderr, "newton replay: upload failed: %s\n", c2 ? last_error(c2) : ""); return 1; }
        nlls_info i2; if (get_info(c2, &i2) != 0 || i2.ndof != (NCAM - 2) * 6 + NPT * 3) { fprintf(stderr, "newton replay: unexpected ndof\n"); return 1; }
        int64_t boff2[NVAR]; if (bsm_index(c2, NULL, NULL, NULL, boff2) != 0) return 1;
        static double variables[NCAM * 6 + NPT * 3], varnext[NCAM * 6 + NPT * 3], xs[NCAM * 6 + NPT * 3];
        for (int i = 0; i < NCAM * 6 + NPT * 3; ++i) variables[i] = i < 12 ? truth[i] : truth[i] + 20.0 * (start[i] - truth[i]);   /* the fixed cameras sit at their true values; the others start 2e-2 away */
        double* pv = variables; double* pn = varnext; double costs[16]; int nit = 0, conv = 0; double best = 0, cN = 0;
#define CK2(call) do { int rc_ = (call); if (rc_ != 0) { fprintf(stderr, "%s -> %d: %s\n", #call, rc_, last_error(c2)); return 1; } } while (0)
        CK2(set_vars(c2, NLLS_VARS_CURRENT, pv)); CK2(sweep_gradhess(c2, &cN)); best = cN; costs[0] = cN;
        while (!conv) {
            ++nit;
            CK2(solve(c2, xs));                                             /* solve! hands back -x; the shim flips it and negate!() flips it again: xs IS the step */
            for (int i = 0, o = 0; i < NVAR; ++i) { const int dof = i < NCAM ? 6 : 3;          /* update!(problem.varnext, problem.variables, linsystem): host, src/linearsystem.jl:206-213 */
                for (int k = 0; k < dof; ++k) pn[o + k] = pv[o + k] + (bi2[i] ? xs[boff2[bi2[i] - 1] - 1 + k] : 0.0);
                o += dof; }
            CK2(set_vars(c2, NLLS_VARS_NEXT, pn)); CK2(sweep_cost(c2, NLLS_VARS_NEXT, &cN));  /* gpucost(ls, problem.varnext) */
            double dcost = best - cN; if (dcost >= 0) best = cN; else dcost = cN;
            { double* tmp = pv; pv = pn; pn = tmp; }                        /* updatefromnext!: problem.variables, problem.varnext = problem.varnext, problem.variables */
            if (nit < 16) costs[nit] = cN;
            conv |= (dcost < best * 1e-15) << 2; conv |= (dcost < 1e-15) << 3; conv |= (nit >= 12) << 8;
            if (conv) break;
            CK2(set_vars(c2, NLLS_VARS_CURRENT, pv)); CK2(sweep_gradhess(c2, &cN));           /* zero! + costgradhess!(linsystem, problem.variables, costs): the HOST vector goes up again */
        }
        double checkN = 0; CK2(set_vars(c2, NLLS_VARS_CURRENT, pv)); CK2(sweep_cost(c2, NLLS_VARS_CURRENT, &checkN));
        printf("replay (Newton through the generic loop, host vectors swapped by updatefromnext!): %.6e -> %.6e -> %.6e -> %.6e ... best %.6e in %d iterations, cost(variables) %.6e\n",
               costs[0], costs[1], costs[2], costs[3], best, nit, checkN);
        if (nit < 3 || !(costs[1] < costs[0]) || !(costs[2] < costs[1]) || !(costs[3] < costs[2]) || !(checkN < 1e-15 * NOBS)) { fprintf(stderr, "the Newton replay made no progress: the host vectors do not advance\n"); return 1; }
        CK2(ctx_destroy(c2));
    }
    /* ---- the same optimisation through the library's own loop (the shim's path when there is no user callback): one ccall ---- */
    fn_lm_iterations lm_iterations = (fn_lm_iterations)need(lib, "nlls_lm_iterations");
    CK(set_vars(ctx, NLLS_VARS_CURRENT, start)); CK(copy_vars(ctx, NLLS_VARS_NEXT, NLLS_VARS_CURRENT));
    CK(sweep_gradhess(ctx, &cost));
    nlls_lm_options lo; memset(&lo, 0, sizeof lo); lo.reldcost = 1e-15; lo.absdcost = 1e-15; lo.dstep = 1e-15; lo.maxfails = 5; lo.maxiters = 100; lo.stoptime_ns = 0;
    nlls_lm_state ls; memset(&ls, 0, sizeof ls); ls.bestcost = cost; ls.cost = cost;
    CK(lm_iterations(ctx, &lo, &ls, (int64_t)1 << 40));
    if (!(ls.bestcost >= ls.cost)) CK(swap_vars(ctx, NLLS_VARS_CURRENT, NLLS_VARS_BEST));
    double check2 = 0; CK(sweep_cost(ctx, NLLS_VARS_CURRENT, &check2));
    printf("replay (nlls_lm_iterations): best %.6e in %lld iterations (%lld LM trials), termination flags %lld, cost(variables) %.6e\n", ls.bestcost, (long long)ls.iternum,
           (long long)ls.linearsolvers, (long long)ls.converged, check2);
    CK(ctx_destroy(ctx));
    if (!(ls.bestcost < 1e-15 * NOBS) || !(check2 < 1e-15 * NOBS) || ls.converged == 0 || ls.iternum > 100) { fprintf(stderr, "nlls_lm_iterations did not reach the zero-residual optimum\n"); return 1; }
    /* ---- the collectives behind the ABI, from plain C (no Python, no PyTorch in this process: the library dlopens librccl.so.1 itself): one rank of one --
     *      nlls_comm_unique_id, nlls_comm_init_rccl, then the SAME single ccall; every entry point of the loop goes through its collective route ---- */
    if (argc > 3 && strcmp(argv[3], "--rccl") == 0) {
        typedef int (*fn_uid)(void*); typedef int (*fn_cinit)(nlls_ctx*, const void*); typedef int (*fn_shard)(nlls_ctx*, int32_t, int32_t);
        fn_uid comm_unique_id = (fn_uid)need(lib, "nlls_comm_unique_id"); fn_cinit comm_init_rccl = (fn_cinit)need(lib, "nlls_comm_init_rccl"); fn_shard set_shard = (fn_shard)need(lib, "nlls_set_shard");
        unsigned char id[128];
        ctx = NULL; CK(ctx_create(NULL, 0, &ctx));
        CK(set_shard(ctx, 0, 1)); CK(comm_unique_id(id)); CK(comm_init_rccl(ctx, id));
        CK(upload(ctx, NVAR, vk, vd, blockindices, 1, &grp, 0));
        CK(set_vars(ctx, NLLS_VARS_CURRENT, start)); CK(copy_vars(ctx, NLLS_VARS_NEXT, NLLS_VARS_CURRENT));
        CK(sweep_gradhess(ctx, &cost));
        nlls_lm_state l3; memset(&l3, 0, sizeof l3); l3.bestcost = cost; l3.cost = cost;
        CK(lm_iterations(ctx, &lo, &l3, (int64_t)1 << 40));
        if (!(l3.bestcost >= l3.cost)) CK(swap_vars(ctx, NLLS_VARS_CURRENT, NLLS_VARS_BEST));
        double check3 = 0; CK(sweep_cost(ctx, NLLS_VARS_CURRENT, &check3));
        printf("replay (collective route over the library's own RCCL communicator, one rank): best %.6e in %lld iterations (%lld LM trials), cost(variables) %.6e\n", l3.bestcost, (long long)l3.iternum,
               (long long)l3.linearsolvers, check3);
        CK(ctx_destroy(ctx));
        if (!(l3.bestcost < 1e-15 * NOBS) || !(check3 < 1e-15 * NOBS) || l3.iternum != ls.iternum) { fprintf(stderr, "the collective route did not reproduce the single-GPU loop\n"); return 1; }
    }
    return 0;
}
