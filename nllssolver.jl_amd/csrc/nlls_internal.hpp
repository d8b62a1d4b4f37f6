// nlls_internal.hpp -- declarations shared by the translation units of libnlls_amd.so
#pragma once

#include <exception>
#include <new>

#include "nlls_ctx.hpp"

#define NLLS_FOR_EACH_RES(X) \
    X(NLLS_RES_BA_AFFINE) X(NLLS_RES_ROSENBROCK_A) X(NLLS_RES_ROSENBROCK_B) X(NLLS_RES_ROSENBROCK_2D) \
    X(NLLS_RES_CURVE_EXP4) X(NLLS_RES_ADAPTIVE_MEAN) X(NLLS_RES_BA_SO3) X(NLLS_RES_BA_SO3_ADAPTIVE) X(NLLS_RES_LINEAR3) X(NLLS_COST_LINEAR3) X(NLLS_RES_SCALE_MIX) NLLS_USER_RES(X)

// Every extern "C" entry point runs between these two (nothing may throw or longjmp across the ccall boundary): a C++ exception -- std::bad_alloc of a host-side
// work vector, above all -- becomes NLLS_ERR_HIP with the message in nlls_last_error.
#define NLLS_API_BEGIN try {
#define NLLS_API_END(C) } catch (const std::bad_alloc&) { return nlls::api_exception(C, "out of host memory (std::bad_alloc)"); } \
    catch (const std::exception& e_) { return nlls::api_exception(C, e_.what()); } catch (...) { return nlls::api_exception(C, "unknown C++ exception"); }

namespace nlls {

inline int api_exception(nlls_ctx* c, const char* what) { if (c) { try { c->err = std::string("host exception: ") + what; } catch (...) {} } return NLLS_ERR_HIP; }

struct ResDesc { int ndeps, nres, ndata, adaptive; int sk[MAX_SLOTS], sd[MAX_SLOTS]; };
bool res_desc(int kind, ResDesc& d);
inline bool is_dyn_kind(int kind) { return kind >= NLLS_RES_DYN_LINEAR && kind <= NLLS_COST_DYN_LINEAR; }

int build_structure(nlls_ctx* c, int64_t nvar, const int32_t* var_kind, const int32_t* var_dim, const uint64_t* bi,
                    int32_t ngroups, const nlls_cost_group* groups, int32_t flags);
int select_elimination(nlls_ctx* c, int32_t flags);
int build_schur(nlls_ctx* c, int32_t flags);
std::vector<int32_t> rcm_order(const std::vector<std::vector<int32_t>>& adj);   // reverse Cuthill-McKee: perm[new position] = node (nlls_structure.cpp)

// sweeps (nlls_sweep.hip): enqueue on c->stream; cost lands in c->scalars[0]
struct PostSolveArgs;
int enqueue_sweep_cost(nlls_ctx* c, int which, int64_t pofs = 0, int64_t* count = nullptr, const PostSolveArgs* post = nullptr, bool* post_taken = nullptr);   // post: the step-statistics roles of an LM trial ride in the first launch (nlls_post.hpp)
// (nlls_cost.hip) cost-only blocks and the final reduction of the cost partials, shared with the gradient sweep
int enqueue_fixedcost(nlls_ctx* c, const Group& G, const double* vars, int64_t& pbase);
int enqueue_dyn_gradhess(nlls_ctx* c, const Group& G, const double* vars, int64_t& pbase);   // dynamic-size residual blocks (dense system): accumulate
int enqueue_reduce_partials(nlls_ctx* c, int64_t n);
int enqueue_check_analytic(nlls_ctx* c, double* d_out, int64_t* nblocks_out);   // closed-form block maths against the dual-number statement (nlls_check_analytic)
int enqueue_sweep_gradhess(nlls_ctx* c, bool want_cost = true, int which = NLLS_VARS_CURRENT, int mode = 0);   // mode 1: the reduced rows only (the matrix-free trial's gradient sweep: nlls_ctx::grad_level 1)   // which: the variable set to linearise at (the look-ahead sweep of an LM trial: NLLS_VARS_NEXT)
// vector helpers (nlls_sweep.hip)
int enqueue_retract(nlls_ctx* c, int to, int from);
int enqueue_step_stats(nlls_ctx* c);          // scalars[1] = max|x|, scalars[2] = x'x
int enqueue_max_abs_diag(nlls_ctx* c);        // scalars[3]
// optimizesingles! (nlls_sweep.hip): all arrays on the device; d_groups = singles_group_size() bytes per cost group
int enqueue_optimize_singles(nlls_ctx* c, int64_t nsel, const int64_t* d_selvar, const int64_t* d_cptr, const int32_t* d_cgroup, const uint32_t* d_cidx,
                             const int32_t* d_cslot, const void* d_groups, int iterator, int maxiters, int maxfails, double reldcost, double absdcost, double dstep, int64_t* d_iters);
int enqueue_copy_var_storage(nlls_ctx* c, const int64_t* d_sel, int64_t nsel, const double* src, double* dst);      // the listed variables' storage, src -> dst (same layout)
int enqueue_iters_to_double(nlls_ctx* c, const int64_t* d_it, const int64_t* d_pos, int64_t n, double* d_out);
size_t singles_group_size();
void singles_group_fill(void* dst, const Group& G);
int enqueue_quadform(nlls_ctx* c, const double* d_vec, int out_slot /* scalars[out], scalars[out+1] = v'Hv, b'v */);
int enqueue_post_solve(nlls_ctx* c, int retract_to = -1, int retract_from = -1, bool finish = true);
int enqueue_lm_trial_tail(nlls_ctx* c, int to, int from);   // post-solve statistics + retraction, cost sweep, and ONE finishing launch for both reductions   // enqueue_step_stats + enqueue_quadform(x, 4) for the step of the last solve, fewer launches; optionally the retraction rides along

#if defined(__HIPCC__)
// to[var i] = update(from[var i], x[its block])   (src/linearsystem.jl:206-213); fixed variables are copied
// dx: the variable's own step (its block of x)
// DX(q) = component q of the variable's own step.  No staging arrays with run-time indices (they would live in scratch memory -- 1040 bytes per lane
// of every kernel this is inlined into): Euclidean / dynamic vectors add in place, the other kinds have compile-time sizes.
template <class DXF>
__device__ __forceinline__ void retract_var_fn(int k, int d, uint32_t o, const double* __restrict__ from, double* __restrict__ to, DXF DX) {
    switch (k) {
    case NLLS_VAR_EUCLIDEAN: case NLLS_VAR_DYNAMIC: for (int q = 0; q < d; ++q) to[o + q] = from[o + q] + DX(q); return;   // v + delta (src/variable.jl:5): any length
    case NLLS_VAR_ZERO_TO_INF: case NLLS_VAR_ZERO_TO_ONE: { const double in0 = from[o], st0 = DX(0); double out0; var_update_real(k, d, &in0, &st0, &out0); to[o] = out0; return; }
    case NLLS_VAR_CONTAMINATED_GAUSSIAN: { double in[3], st[3], out[3];
#pragma unroll
        for (int q = 0; q < 3; ++q) { in[q] = from[o + q]; st[q] = DX(q); }
        var_update_real(NLLS_VAR_CONTAMINATED_GAUSSIAN, d, in, st, out);
#pragma unroll
        for (int q = 0; q < 3; ++q) to[o + q] = out[q];
        return; }
    case NLLS_VAR_POSE_SO3: { double in[12], st[6], out[12];
#pragma unroll
        for (int q = 0; q < 12; ++q) in[q] = from[o + q];
#pragma unroll
        for (int q = 0; q < 6; ++q) st[q] = DX(q);
        var_update_real(NLLS_VAR_POSE_SO3, d, in, st, out);
#pragma unroll
        for (int q = 0; q < 12; ++q) to[o + q] = out[q];
        return; }
    }
}
__device__ __forceinline__ void retract_var(int k, int d, uint32_t o, const double* __restrict__ from, const double* __restrict__ dx, double* __restrict__ to) {
    retract_var_fn(k, d, o, from, to, [dx](int q) { return dx[q]; });
}
__device__ __forceinline__ void retract_one(const int32_t* __restrict__ kind, const int32_t* __restrict__ dim, const uint32_t* __restrict__ voff,
                                            const uint32_t* __restrict__ vboff, int64_t i, const double* __restrict__ from,
                                            const double* __restrict__ x, double* __restrict__ to) {
    const int k = kind[i], d = dim[i]; const uint32_t o = voff[i], bo = vboff[i];
    if (bo == DEST_NONE) { const int st = var_storage(k, d); for (int q = 0; q < st; ++q) to[o + q] = from[o + q]; return; }
    retract_var(k, d, o, from, x + bo, to);
}
#endif
// solve (nlls_solve.hip)
int enqueue_solve(nlls_ctx* c);
int enqueue_solve_local(nlls_ctx* c);
int enqueue_solve_finish(nlls_ctx* c);
int enqueue_tiny_dense_trial(nlls_ctx* c, int to, int from, bool lookahead_follows);
int enqueue_tiny_trial_finish_pending(nlls_ctx* c);   // the finishing reduction no accumulate launch has carried   // nlls_ctx::tiny_dense: damped solve + step statistics + retraction in one launch, then the cost sweep
int enqueue_chain_solve(nlls_ctx* c, int n_band, int bw, int nbd, int H);   // the banded reduced system by the chain kernels (nlls_chain.hip)
int enqueue_reduced_solve(nlls_ctx* c);   // the factorisation + backward pass of the assembled reduced system alone (timing)
int enqueue_pack_reduce0(nlls_ctx* c);      // [cost | reduced rows | reduced b] -> redbuf
int enqueue_unpack_reduce0(nlls_ctx* c, bool with_cost = true);


// the matrix-free LM trial (nlls_mf.hip; nlls_ctx::mf_ok)
struct BsfRetract;
int enqueue_mf_solve_local(nlls_ctx* c);
int enqueue_mf_backsub(nlls_ctx* c, const BsfRetract& rt, int write_red, double* zptr, int64_t zcount, unsigned nextra, unsigned nrestwg);
int enqueue_mf_trial_finish(nlls_ctx* c); int enqueue_mf_trial_finish_now(nlls_ctx* c); struct MfFin; MfFin mf_fin_args(nlls_ctx* c); size_t mf_part_doubles(int64_t nsupernodes, int64_t nrest_wg_max);
int enqueue_mf_sweep_cost(nlls_ctx* c, int which);   // cost(vars[which]) summed as the matrix-free trial sums its cost
int enqueue_gather(nlls_ctx* c);   // (nlls_solve.hip) schur_gather_kernel: slabs -> the block cyclic reduction's tiles
uint32_t mf_wave_doubles(uint32_t ecap, int dp); int mf_batch_max(); int mf_elim_waves(); uint32_t mf_slab_doubles(int B, int dp, int tr);
int build_mf(nlls_ctx* c, int32_t ngroups, const nlls_cost_group* groups, const uint64_t* bi, int32_t flags);   // (nlls_structure.cpp)

// collectives (nlls_comm.cpp)
int comm_reduce(nlls_ctx* c, double* dev_ptr, int64_t count, int op);                 // no-op without an installed all-reduce
int comm_gather_trial_scalars(nlls_ctx* c, double seq);                               // gather + combine the trial's scalars on the device, publish them (and seq) to the host mirror
void comm_release(nlls_ctx* c);

inline double* vars_ptr(nlls_ctx* c, int which) { return c->vars[c->vars_slot[which]].p; }

}  // namespace nlls
