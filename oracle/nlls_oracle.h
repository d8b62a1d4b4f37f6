/*
 * oracle/nlls_oracle.h -- TEST INFRASTRUCTURE: CPU restatement of the NLLSsolver.jl hot path.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library; it
 * is the checker, never the product.  Each function cites the reference file:line it restates
 * (paths relative to /root/reference).  Parity pins: tests/test_oracle_pins.py checks it against
 * every RNG-free known-answer the reference's own tests hold for this path.
 * The reference itself (Julia) cannot be executed in this environment (no julia binary).
 */
#ifndef NLLS_ORACLE_H
#define NLLS_ORACLE_H

#include <stdint.h>
#include "../include/nlls_amd.h" /* kind enums + nlls_cost_group only */

#ifdef __cplusplus
extern "C" {
#endif

typedef struct oracle_problem oracle_problem;
typedef struct oracle_ls oracle_ls;

/* ---- unit-level pieces (pinned by golden vectors) ------------------------------------------- */
/* src/robust.jl:7-77, src/robustadaptive.jl:25-33.  kparams: for fixed kernels = robust_params;
 * for NLLS_VAR_CONTAMINATED_GAUSSIAN kernels pass kind = -1 and kparams = (1/s1, 1/s2, w).     */
double oracle_robustify(int32_t robust_kind, const double* kparams, double cost);
void   oracle_robustifydcost(int32_t robust_kind, const double* kparams, double cost, double out[3]);
/* src/autodiff.jl:163: second-order AD of robustify w.r.t. the cost (checks the analytic forms,
 * as test/robust.jl:9 does) */
void   oracle_autorobustifydcost(int32_t robust_kind, const double* kparams, double cost, double out[3]);
/* src/autodiff.jl:164-165: value, gradient[4], hessian[4][4] of robustify(update(kernel,x), cost+x[4]) */
void   oracle_robustifydkernel(const double* cg_storage, double cost, double* value, double grad[4], double hess[16]);
/* ContaminatedGaussian(s1, s2, w) constructor, src/robustadaptive.jl:12-20: returns storage (1/s1,1/s2,w) ordered */
void   oracle_contaminated_gaussian(double s1, double s2, double w, double storage[3]);
/* src/utils.jl:38-52 (1-based run indices, as Julia returns them); returns length written */
int64_t oracle_runlengthencodesortedints(const int64_t* sortedints, int64_t n, int64_t* out);
/* src/utils.jl:71-81 dense (col-major) and :95-106 CSC */
double oracle_fast_bAb_dense(const double* A, const double* b, int64_t n);
double oracle_fast_bAb_csc(const int64_t* colptr, const int64_t* rowval, const double* nzval, const double* b, int64_t n);
/* src/linearsolver.jl:20-26: cholesky, else QR.  A col-major n x n (full). returns 0 chol, 1 qr */
int    oracle_solve_dense(double* x, const double* A, const double* b, int64_t n);
/* src/linearsolver.jl:32 + LDLFactorizations (restated Davis LDL, see .c): full symmetric CSC, 1-based */
int    oracle_solve_sparse(double* x, const int64_t* colptr, const int64_t* rowval, const double* nzval, const double* b, int64_t n);
/* src/variable.jl update(): storage_out = update(storage_in, step) */
void   oracle_var_update(int32_t kind, int32_t dim, const double* in, const double* step, double* out);

/* ---- BlockSparseMatrix (src/BlockSparseMatrix.jl) ------------------------------------------- */
/* constructor :30-47: sparsitytransposed as CSC (ncolblocks x nrowblocks, 1-based colptr/rowval).
 * Writes nzval (1-based start offsets) and returns length(data). */
int64_t oracle_bsm_build(int64_t nrowblocks, int64_t ncolblocks, const int64_t* st_colptr, const int64_t* st_rowval,
                         const int32_t* rowsizes, const int32_t* colsizes, int64_t* nzval_out);
/* Base.Matrix :245-264: dense col-major m x n */
void    oracle_bsm_to_dense(int64_t nrowblocks, int64_t ncolblocks, const int64_t* st_colptr, const int64_t* st_rowval,
                            const int64_t* nzval, const int32_t* rowsizes, const int32_t* colsizes, const double* data, double* out);
/* symmetrifyfull :198-243 (square, lower triangular BSM) */
void    oracle_bsm_symmetrify_full(int64_t nblocks, const int64_t* st_colptr, const int64_t* st_rowval, const int64_t* nzval,
                                   const int32_t* sizes, const double* data, double* out);
/* makesparseindices :141-191.  Two-call protocol: pass NULL outputs to get nnz. Outputs 1-based. */
int64_t oracle_bsm_sparse_indices(int64_t nrowblocks, int64_t ncolblocks, const int64_t* st_colptr, const int64_t* st_rowval,
                                  const int64_t* nzval, const int32_t* rowsizes, const int32_t* colsizes, int64_t datalen,
                                  int symmetrify, int64_t* colptr_out, int64_t* rowval_out, int64_t* index_out);

/* ---- problem (src/problem.jl:5-20) ----------------------------------------------------------- */
oracle_problem* oracle_problem_create(int64_t nvar, const int32_t* var_kind, const int32_t* var_dim,
                                      int32_t ngroups, const nlls_cost_group* groups);
void    oracle_problem_destroy(oracle_problem* p);
int64_t oracle_problem_storage(const oracle_problem* p);
void    oracle_set_variables(oracle_problem* p, int32_t which, const double* packed);
void    oracle_get_variables(const oracle_problem* p, int32_t which, double* packed);
/* src/cost.jl:10-13 */
double  oracle_cost(const oracle_problem* p, int32_t which);
/* per-block (cost, g, H) of src/residual.jl:57-111 for one cost: group gi, cost ci, evaluated with
 * all variables free (varflags = all ones).  g has P entries, H is P x P col-major. returns P. */
int     oracle_block_costgradhess(const oracle_problem* p, int32_t which, int32_t gi, int64_t ci,
                                  double* cost, double* g, double* H);
/* residual + jacobian (M x P col-major) of src/autodiff.jl:81-93 */
int     oracle_block_resjac(const oracle_problem* p, int32_t which, int32_t gi, int64_t ci, double* r, double* J);

/* ---- linear system (src/linearsystem.jl) ----------------------------------------------------- */
/* makesymmvls :91-124; flags: NLLS_FLAG_FORCE_SPARSE */
oracle_ls* oracle_makesymmvls(const oracle_problem* p, const uint64_t* blockindices, int32_t flags);
void    oracle_ls_destroy(oracle_ls* ls);
void    oracle_ls_info(const oracle_ls* ls, nlls_info* out);
double* oracle_ls_data(oracle_ls* ls);    /* A.data */
double* oracle_ls_b(oracle_ls* ls);
double* oracle_ls_x(oracle_ls* ls);
void    oracle_ls_bsm_index(const oracle_ls* ls, int64_t* colptr, int64_t* rowval, int64_t* nzval, int64_t* boffsets);
/* zero! :192-195 + costgradhess! (src/cost.jl:29-54) */
double  oracle_costgradhess(const oracle_problem* p, int32_t which, oracle_ls* ls);
/* gethessian :180-189 (+ symmetric CSC) then solve (H + lambda I) y = b; x = -y  (iterators.jl:149-152) */
int     oracle_solve_damped(oracle_ls* ls, double lambda);
double  oracle_max_abs_diag(const oracle_ls* ls);                    /* iterators.jl:131-137 (without 1e-6) */
double  oracle_quadform(const oracle_ls* ls, const double* x, double lambda); /* fast_bAb(H + lambda I, x) */
/* update! :206-213 */
void    oracle_update(oracle_problem* p, int32_t to, int32_t from, const oracle_ls* ls, const double* step);

/* ---- optimize! (src/optimize.jl:5-17,109-180; src/iterators.jl) ------------------------------ */
typedef struct oracle_options {      /* src/structs.jl:22-35 */
    double  reldcost, absdcost, dstep;
    int64_t maxfails, maxiters;
    double  maxtime;                 /* seconds */
    int32_t iterator;                /* 0 newton, 1 levenbergmarquardt, 2 dogleg, 3 gradientdescent (structs.jl:5) */
    int32_t store_costs;             /* record cost after every iteration into result costs[] (callbacks.jl:63-66) */
} oracle_options;
typedef struct oracle_result {       /* src/structs.jl:37-50 */
    double  startcost, bestcost;
    double  timetotal, timeinit, timecost, timegradient, timesolver;
    int64_t termination, niterations, costcomputations, gradientcomputations, linearsolvers;
    int64_t ncosts_stored;
    double  costs[512];
} oracle_result;
void oracle_default_options(oracle_options* o);
int  oracle_optimize(oracle_problem* p, const uint64_t* blockindices, const oracle_options* opt, oracle_result* res);

#ifdef __cplusplus
}
#endif
#endif
