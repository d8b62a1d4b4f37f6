// nlls_lm.cpp -- the Levenberg-Marquardt outer loop on the host side of the C ABI, in C++ (no Python / Julia between two trials).
//
//   optimizeinternal!  src/optimize.jl:124-171   (one pass of the while-loop body per outer iteration)
//   iterate!(::LevMarData)  src/iterators.jl:139-172
//
// Written ONLY in terms of the public entry points of include/nlls_amd.h (nlls_lm_trial, nlls_damp, nlls_quadform, nlls_step_maxabs,
// nlls_swap_variables, nlls_copy_variables, nlls_sweep_gradhess ...): it is the loop a host binding would write, statement for
// statement -- nllssolver.jl_amd/optimizer.py::OuterLoop.iteration + iterators.py::iterate_levmar are the same code in Python and
// tests/test_gpu_functional.py holds the two against each other.  Why it exists: between two trials the GPU waits for the host (the
// trial's scalars decide what is enqueued next); through an interpreter that turn-around was ~45 us of a 400 us trial.
#include <cmath>
#include <ctime>
#include <limits>
#include <vector>

#include <exception>

#include "../../include/nlls_amd.h"

// (this file sees the public header only: the boundary guard of nlls_internal.hpp, restated -- no exception leaves nlls_lm_iterations)
#define NLLS_API_BEGIN try {
#define NLLS_API_END(C) } catch (...) { return NLLS_ERR_HIP; }

namespace {
int64_t monotonic_ns() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return (int64_t)ts.tv_sec * 1000000000LL + ts.tv_nsec; }
struct Timer { int64_t& acc; int64_t t0; explicit Timer(int64_t& a) : acc(a), t0(monotonic_ns()) {} ~Timer() { acc += monotonic_ns() - t0; } };

// iterate!(levmardata, data, problem, options)   src/iterators.jl:139-172.  *cost_out = the cost of the accepted trial.
int iterate_levmar(nlls_ctx* ctx, const nlls_lm_options* opt, nlls_lm_state* st, double* cost_out) {
    if (!(st->lambda >= 0.0)) return NLLS_ERR_INVALID_ARG;                              // :140  @assert levmardata.lambda >= 0.
    int rc;
    if (st->lambda == 0.0) { double m; if ((rc = nlls_max_abs_diag(ctx, &m)) != NLLS_OK) return rc; st->lambda = m * 1e-6; }   // :142-144, :131-137
    double lastlambda = 0.0, mu = 2.0;
    for (;;) {
        double cost_ = 0.0;
        { Timer t(st->timesolver_ns);
          rc = nlls_lm_trial(ctx, st->lambda - lastlambda, NLLS_VARS_NEXT, NLLS_VARS_CURRENT, &cost_); }   // :149-157 in one call
        lastlambda = st->lambda;
        st->linearsolvers++;
        if (rc == NLLS_ERR_NOT_SPD) {
            // the damped factorisation met an exactly zero (or NaN) pivot.  A system that ITSELF holds NaN / Inf is not cured by damping:
            // the reference's factorisation does not throw on it, its step and trial cost come out NaN and '!(cost_ > bestcost)' accepts
            // them (src/optimize.jl:147-152 then report both).  Otherwise: a rejected trial -- more damping, solve again (the reference's
            // LDLFactorizations would throw here; counted in singulartrials).
            double g2 = 0.0; if ((rc = nlls_grad_sqnorm(ctx, &g2)) != NLLS_OK) return rc;
            if (!std::isfinite(st->lambda) || !std::isfinite(g2)) {
                nlls_info info; if ((rc = nlls_get_info(ctx, &info)) != NLLS_OK) return rc;
                std::vector<double> nanx((size_t)info.ndof, std::numeric_limits<double>::quiet_NaN());
                if ((rc = nlls_set_step(ctx, nanx.data())) != NLLS_OK) return rc;
                if ((rc = nlls_retract(ctx, NLLS_VARS_NEXT, NLLS_VARS_CURRENT)) != NLLS_OK) return rc;
                if ((rc = nlls_sweep_cost(ctx, NLLS_VARS_NEXT, &cost_)) != NLLS_OK) return rc;
                st->costcomputations++;
            } else {
                if (!std::isfinite(st->lambda * mu)) return NLLS_ERR_NOT_SPD;           // nothing left to damp with
                st->singulartrials++;
                st->lambda *= mu; mu *= 2.0;
                continue;
            }
        } else if (rc != NLLS_OK) return rc;
        else st->costcomputations++;
        double maxstep = 0.0; if ((rc = nlls_step_maxabs(ctx, &maxstep)) != NLLS_OK) return rc;
        if (!(cost_ > st->bestcost) || maxstep < opt->dstep) {                           // :160
            if ((rc = nlls_damp(ctx, -lastlambda)) != NLLS_OK) return rc;               // :162
            double xHx = 0.0, gx = 0.0; if ((rc = nlls_quadform(ctx, &xHx, &gx)) != NLLS_OK) return rc;
            const double q = (cost_ - st->bestcost) / (0.5 * xHx + gx);                  // :163
            st->lambda *= q < 0.983 ? 1.0 - (2.0 * q - 1.0) * (2.0 * q - 1.0) * (2.0 * q - 1.0) : 0.1;   // :164
            *cost_out = cost_;
            return NLLS_OK;
        }
        st->lambda *= mu; mu *= 2.0;                                                      // :169-170
    }
}
}  // namespace

extern "C" int nlls_lm_iterations(nlls_ctx* ctx, const nlls_lm_options* opt, nlls_lm_state* st, int64_t niter) { NLLS_API_BEGIN
    if (!ctx || !opt || !st) return NLLS_ERR_INVALID_ARG;
    // the time buckets of NLLSResult (src/structs.jl:42-44): where the trials' launches time themselves on the device (nlls_get_time_buckets) those figures replace this loop's
    // host timers -- which only see enqueues -- when the call returns
    int64_t tb0[4] = {0, 0, 0, 0}; const bool have_tb = nlls_get_time_buckets(ctx, tb0, 4) == NLLS_OK;
    const int64_t ts0 = st->timesolver_ns, tg0 = st->timegradient_ns, tc0 = st->timecost_ns;
    struct Fill { nlls_ctx* ctx; nlls_lm_state* st; const int64_t* tb0; bool have; int64_t ts0, tg0, tc0;
        ~Fill() { int64_t tb1[4]; if (!have || nlls_get_time_buckets(ctx, tb1, 4) != NLLS_OK || tb1[3] == tb0[3]) return;
                  st->timegradient_ns = tg0 + (tb1[0] - tb0[0]); st->timecost_ns = tc0 + (tb1[1] - tb0[1]); st->timesolver_ns = ts0 + (tb1[2] - tb0[2]); } } fill{ctx, st, tb0, have_tb, ts0, tg0, tc0};
    const bool timed = opt->stoptime_ns > 0;
    for (int64_t it = 0; it < niter; ++it) {
        // the deadline under sharding: every rank has its own clock -- each posts what ITS clock says now, the flags ride in the scalar gather of this
        // iteration's trials, and all ranks stop on the agreed maximum (one process: the clock is read after the iteration, as src/optimize.jl:158 does)
        if (timed) { const int rcp = nlls_comm_post_flag(ctx, monotonic_ns() > opt->stoptime_ns ? 1.0 : 0.0); if (rcp != NLLS_OK) return rcp; }
        st->iternum++;                                                                    // src/optimize.jl:124
        double cost = 0.0;
        int rc = iterate_levmar(ctx, opt, st, &cost);                                     // :126
        if (rc != NLLS_OK) return rc;
        double dcost = st->bestcost - cost;                                               // :130
        if (dcost >= 0) { st->bestcost = cost; st->fails = 0; }
        else {
            dcost = cost; st->fails++;
            if (st->fails == 1) {                                                         // :137-144 store the current best variables
                if (st->have_best) rc = nlls_swap_variables(ctx, NLLS_VARS_CURRENT, NLLS_VARS_BEST);
                else { rc = nlls_copy_variables(ctx, NLLS_VARS_BEST, NLLS_VARS_CURRENT); st->have_best = 1; }
                if (rc != NLLS_OK) return rc;
            }
        }
        if ((rc = nlls_swap_variables(ctx, NLLS_VARS_CURRENT, NLLS_VARS_NEXT)) != NLLS_OK) return rc;   // updatefromnext!  :207-209
        double maxstep = 0.0; if ((rc = nlls_step_maxabs(ctx, &maxstep)) != NLLS_OK) return rc;          // :149
        int64_t conv = 0;
        conv |= (int64_t)std::isinf(cost) << 0;
        conv |= (int64_t)std::isnan(cost) << 1;
        conv |= (int64_t)(dcost < st->bestcost * opt->reldcost) << 2;
        conv |= (int64_t)(dcost < opt->absdcost) << 3;
        conv |= (int64_t)std::isinf(maxstep) << 4;
        conv |= (int64_t)std::isnan(maxstep) << 5;
        conv |= (int64_t)(maxstep < opt->dstep) << 6;
        conv |= (int64_t)(st->fails > opt->maxfails) << 7;
        conv |= (int64_t)(st->iternum >= opt->maxiters) << 8;
        if (timed) { double late = 0.0; if ((rc = nlls_comm_agreed_flag(ctx, monotonic_ns() > opt->stoptime_ns ? 1.0 : 0.0, &late)) != NLLS_OK) return rc;
            conv |= (int64_t)(late > 0.0) << 9; }
        st->converged = conv; st->cost = cost;
        if (conv != 0) break;
        { Timer t(st->timegradient_ns);
          if ((rc = nlls_sweep_gradhess(ctx, nullptr)) != NLLS_OK) return rc; }         // :167-170 (the value is discarded there too): enqueue only
        st->gradientcomputations++;
    }
    return NLLS_OK;
    NLLS_API_END(ctx)
}
