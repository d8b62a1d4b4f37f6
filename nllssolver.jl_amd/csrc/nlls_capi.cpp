// nlls_capi.cpp -- the extern "C" boundary declared in include/nlls_amd.h.
#include <algorithm>
#include <atomic>
#include <chrono>

#include "nlls_internal.hpp"

using namespace nlls;

namespace {
constexpr int32_t NLLS_MAX_GRAPH_NODES = 1 << 27;      // nlls_rcm_order / nlls_nd_tiles: nodes of a reduced-system graph
int fail(nlls_ctx* c, int code, const std::string& msg) { if (c) c->err = msg; return code; }
int herr(nlls_ctx* c, hipError_t e, const char* what) { c->err = std::string(what) + ": " + hipGetErrorString(e); return NLLS_ERR_HIP; }
#define HIPCHK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return herr(ctx, e_, #expr); } while (0)
// (every entry point launches on ctx->stream: make the context's device current first -- two contexts on different devices in one
// process, or a caller that changed the current device, must not end up on a foreign stream)
#define NEED_READY() do { if (!ctx) return NLLS_ERR_INVALID_ARG; if (!ctx->ready) return fail(ctx, NLLS_ERR_NOT_READY, "nlls_upload_structure has not succeeded"); (void)hipSetDevice(ctx->device); } while (0)
#define NEED_GRAD_LAZY() do { NEED_READY(); if (!ctx->have_grad) return fail(ctx, NLLS_ERR_NOT_READY, "nlls_sweep_gradhess has not been run"); } while (0)
// (... and with the reduced rows summed over ranks: every entry point but the LM trial itself reads them as if one GPU had swept all cost blocks)
#define NEED_GRAD() do { NEED_GRAD_LAZY(); TRY(ensure_grad(ctx, 2)); TRY(ensure_reduced_summed(ctx)); } while (0)
#define TRY(expr) do { int rc_ = (expr); if (rc_ != NLLS_OK) return rc_; } while (0)

// copy `count` scalars starting at `slot` to the pinned mirror and wait
int fetch_scalars(nlls_ctx* ctx, int slot, int count) {
    HIPCHK(hipMemcpyAsync(ctx->h_scalars + slot, ctx->scalars.p + slot, sizeof(double) * count, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return NLLS_OK;
}
bool valid_set(int w) { return w >= 0 && w < 3; }
// Look-ahead sweep (nlls_ctx::spec_pending): A and b may hold the linearisation at the last trial's point instead of the current one.  Whoever needs them for the
// CURRENT point comes through here: after the swap of an accepted trial they simply ARE the current point's (a hit); otherwise the current point is swept again (a miss:
// one accumulate launch, the damping kept) and the look-ahead stays off until the next sweep the caller asks for.
// `level` (round 6, nlls_ctx::grad_level): 1 -- the reduced rows suffice (the matrix-free LM trial); 2 -- all of A.data and b.  A linearisation that holds less than is asked for
// is swept (again) here, at CURRENT: the matrix-free trial never forms the eliminated rows, and nlls_sweep_gradhess(ctx, NULL) defers its sweep to the first call that says
// which level it needs.
int ensure_grad(nlls_ctx* ctx, int level) {
    if (ctx->spec_pending) {
        ctx->spec_pending = false;
        if (!ctx->spec_stale && ctx->grad_phys == ctx->vars_slot[NLLS_VARS_CURRENT]) ctx->spec_hits++;
        else { ctx->spec_misses++; ctx->spec_armed = false; ctx->grad_level = 0; }
        ctx->spec_stale = false;
    }
    if (ctx->grad_level >= level && ctx->grad_phys == ctx->vars_slot[NLLS_VARS_CURRENT]) return NLLS_OK;
    const double lam = ctx->lambda;
    TRY(enqueue_sweep_gradhess(ctx, false, NLLS_VARS_CURRENT, level == 1 ? 1 : 0));
    ctx->lambda = lam; ctx->solved = false;
    return NLLS_OK;
}
int ensure_grad_current(nlls_ctx* ctx) { return ensure_grad(ctx, 2); }
// a variable set is about to be written: a look-ahead sweep of it is stale, and so is a linearisation at it that is not (fully) formed yet
void spec_note_write(nlls_ctx* ctx, int32_t which) {
    if (ctx->vars_slot[which] != ctx->grad_phys) return;
    if (ctx->spec_pending) ctx->spec_stale = true;
    else if (ctx->grad_level < 2) ctx->grad_level = 0;          // (what is formed on demand would be formed at the NEW values: the caller sweeps again after writing CURRENT -- every iterator does)
}
// phase events (nlls_ctx::phase_on): record event k on the stream
void phase_mark(nlls_ctx* ctx, int k) { if (ctx->phase_on && (size_t)k < ctx->phase_ev.size()) (void)hipEventRecord(ctx->phase_ev[k], ctx->stream); }
// is this trial matrix-free?  (nlls_ctx::mf_ok: the structure qualifies; mf_on: not switched off; the trial starts at CURRENT, one rank, no collective route)
bool mf_trial(const nlls_ctx* ctx, int32_t from) { return ctx->mf_ok && ctx->mf_on && from == NLLS_VARS_CURRENT && ctx->nranks == 1 && !ctx->reduce_fn && ctx->info.is_sparse && !ctx->tiny_dense; }
}  // namespace

extern "C" {

int nlls_ctx_create(const int32_t* device_ids, int32_t ndev, nlls_ctx** out) { NLLS_API_BEGIN
    if (!out) return NLLS_ERR_INVALID_ARG;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return NLLS_ERR_NO_DEVICE;   // never falls back to a CPU path
    int dev = (device_ids && ndev > 0) ? device_ids[0] : 0;
    if (dev < 0 || dev >= count) return NLLS_ERR_INVALID_ARG;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return NLLS_ERR_HIP;
    if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0) return NLLS_ERR_NO_DEVICE;  // the code object is gfx950-only
    nlls_ctx* c = new (std::nothrow) nlls_ctx();
    if (!c) return NLLS_ERR_HIP;
    c->device = dev; c->num_cus = prop.multiProcessorCount;
    { int khz = 100000; if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) != hipSuccess || khz <= 0) khz = 100000; c->tb_ns_per_tick = 1e6 / (double)khz; }
    { const char* e = getenv("NLLS_NO_LOOKAHEAD_SWEEP"); if (e && e[0] == '1') c->spec_on = false; }
    { const char* e = getenv("NLLS_MATERIALIZE"); if (e && e[0] == '1') c->mf_on = false; }
    { const char* e = getenv("NLLS_TINY_DENSE"); if (e && e[0] == '0') c->tiny_dense_on = false; }
    { const char* e = getenv("NLLS_TINY_FIN_ROLE"); if (e && e[0] == '0') c->tiny_fin_role = false; }             // (A/B: the trial's finishing reduction always in a launch of its own)
    { const char* e = getenv("NLLS_ELIM_TILED"); if (e && e[0] == '1') c->elim_mfma = false; }
    { const char* e = getenv("NLLS_DENSE_T64"); if (e && e[0] == '1') c->dense_t128 = false; }
    { const char* e = getenv("NLLS_EAGER_STAGE0"); if (e && e[0] == '1') c->lazy_stage0 = false; }
    { const char* e = getenv("NLLS_POST_SPLIT"); if (e && e[0] == '1') c->post_fuse = false; }
    { const char* e = getenv("NLLS_ELIM_SPLIT"); if (e && e[0] == '1') c->elim_split = true; }
    { const char* e = getenv("NLLS_DENSE_STEP_BACKWARD"); if (e && e[0] == '1') c->dense_fused_bwd = false; }
    { const char* e = getenv("NLLS_DENSE_T128_MIN"); if (e) c->dense_t128_min = atoi(e); }
    if (hipSetDevice(dev) != hipSuccess || hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { delete c; return NLLS_ERR_HIP; }
    c->own_stream = true;
    if (hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming) != hipSuccess) { delete c; return NLLS_ERR_HIP; }
    *out = c;
    return NLLS_OK;
    NLLS_API_END(nullptr)
}

int nlls_ctx_destroy(nlls_ctx* ctx) { NLLS_API_BEGIN
    if (!ctx) return NLLS_OK;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    comm_release(ctx);
    if (ctx->h_scalars) (void)hipHostFree(ctx->h_scalars);
    if (ctx->stream2) { (void)hipStreamSynchronize(ctx->stream2); (void)hipStreamDestroy(ctx->stream2); }
    for (auto& e : ctx->prof_ev) (void)hipEventDestroy(e);
    for (auto& e : ctx->phase_ev) (void)hipEventDestroy(e);
    if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
    if (ctx->ev_join) (void)hipEventDestroy(ctx->ev_join);
    hipStream_t s = ctx->own_stream ? ctx->stream : nullptr;
    delete ctx;
    if (s) (void)hipStreamDestroy(s);
    return NLLS_OK;
    NLLS_API_END(ctx)
}

const char* nlls_last_error(const nlls_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int nlls_set_stream(nlls_ctx* ctx, void* hip_stream) { NLLS_API_BEGIN
    if (!ctx) return NLLS_ERR_INVALID_ARG;
    (void)hipStreamSynchronize(ctx->stream);
    if (ctx->own_stream && ctx->stream) { (void)hipStreamDestroy(ctx->stream); ctx->own_stream = false; }
    if (hip_stream) ctx->stream = reinterpret_cast<hipStream_t>(hip_stream);
    else { if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) return NLLS_ERR_HIP; ctx->own_stream = true; }
    return NLLS_OK;
    NLLS_API_END(ctx)
}

int nlls_set_shard(nlls_ctx* ctx, int32_t rank, int32_t nranks) { NLLS_API_BEGIN
    if (!ctx || nranks < 1 || rank < 0 || rank >= nranks) return NLLS_ERR_INVALID_ARG;
    ctx->rank = ctx->shard_rank = rank; ctx->nranks = ctx->shard_nranks = nranks; ctx->replicated = false; ctx->ready = false;
    return NLLS_OK;
    NLLS_API_END(ctx)
}

int nlls_var_storage(int32_t k, int32_t d) { NLLS_API_BEGIN return var_storage(k, d); NLLS_API_END(nullptr) }
int nlls_var_dof(int32_t k, int32_t d) { NLLS_API_BEGIN return var_dof(k, d); NLLS_API_END(nullptr) }
int nlls_res_ndeps(int32_t k) { NLLS_API_BEGIN ResDesc d; return res_desc(k, d) ? d.ndeps : NLLS_ERR_UNSUPPORTED; NLLS_API_END(nullptr) }
int nlls_res_nres(int32_t k) { NLLS_API_BEGIN ResDesc d; return res_desc(k, d) ? d.nres : NLLS_ERR_UNSUPPORTED; NLLS_API_END(nullptr) }
int nlls_res_ndata(int32_t k) { NLLS_API_BEGIN ResDesc d; return res_desc(k, d) ? d.ndata : NLLS_ERR_UNSUPPORTED; NLLS_API_END(nullptr) }
int nlls_res_slot_kind(int32_t k, int32_t slot, int32_t* vk, int32_t* vd) { NLLS_API_BEGIN
    ResDesc d; if (!res_desc(k, d)) return NLLS_ERR_UNSUPPORTED;
    if (slot < 0 || slot >= d.ndeps) return NLLS_ERR_INVALID_ARG;
    if (vk) *vk = d.sk[slot]; if (vd) *vd = d.sd[slot];
    return NLLS_OK;
    NLLS_API_END(nullptr)
}
int nlls_rcm_order(int32_t n, const int64_t* adjptr, const int32_t* adj, int32_t* perm_out) { NLLS_API_BEGIN
    if (n < 0 || (n > 0 && (!adjptr || !perm_out))) return NLLS_ERR_INVALID_ARG;
    if (n > NLLS_MAX_GRAPH_NODES) return NLLS_ERR_INVALID_ARG;          // (a size no reduced system has: refused before any work vector is sized by it)
    std::vector<std::vector<int32_t>> a((size_t)n);
    for (int32_t i = 0; i < n; ++i) { if (adjptr[i + 1] < adjptr[i]) return NLLS_ERR_INVALID_ARG;
        for (int64_t q = adjptr[i]; q < adjptr[i + 1]; ++q) { if (adj[q] < 0 || adj[q] >= n || adj[q] == i) return NLLS_ERR_INVALID_ARG; a[i].push_back(adj[q]); } }
    const std::vector<int32_t> p = nlls::rcm_order(a);
    for (int32_t i = 0; i < n; ++i) perm_out[i] = p[i];
    return NLLS_OK;
    NLLS_API_END(nullptr)
}

int nlls_nd_tiles(int32_t n, int32_t nborder, const int64_t* adjptr, const int32_t* adj, const int32_t* dof, int32_t* tile_of, int32_t* row_in_tile,
                  int32_t max_tiles, int32_t* parent, int32_t* level, int64_t* colptr, int64_t max_rows, int32_t* rows) {
    if (n > NLLS_MAX_GRAPH_NODES || nborder > NLLS_MAX_GRAPH_NODES) return NLLS_ERR_INVALID_ARG;
    if (n < 0 || nborder < 0 || (n > 0 && !adjptr) || (n + nborder > 0 && (!dof || !tile_of || !row_in_tile)) || !parent || !level || !colptr || (max_rows > 0 && !rows)) return NLLS_ERR_INVALID_ARG;
    std::vector<std::vector<int32_t>> a((size_t)n);
    for (int32_t i = 0; i < n; ++i) { if (adjptr[i + 1] < adjptr[i]) return NLLS_ERR_INVALID_ARG;
        for (int64_t q = adjptr[i]; q < adjptr[i + 1]; ++q) { if (adj[q] < 0 || adj[q] >= n || adj[q] == i) return NLLS_ERR_INVALID_ARG; a[i].push_back(adj[q]); }
        std::sort(a[i].begin(), a[i].end()); a[i].erase(std::unique(a[i].begin(), a[i].end()), a[i].end()); }
    nlls::TspSym sym;
    if (!nlls::tsp_symbolic(a, std::vector<int32_t>(dof, dof + n + nborder), nborder, sym)) return NLLS_ERR_INVALID_ARG;
    if (sym.nt > max_tiles || sym.ntiles_lower - sym.nt > max_rows) return NLLS_ERR_INVALID_ARG;
    for (int32_t i = 0; i < n + nborder; ++i) { tile_of[i] = sym.tile_of[i]; row_in_tile[i] = sym.row_in_tile[i]; }
    colptr[0] = 0;
    for (int k = 0; k < sym.nt; ++k) { parent[k] = sym.parent[k]; level[k] = sym.level[k]; int64_t q = colptr[k]; for (int32_t t : sym.cstruct[k]) rows[q++] = t; colptr[k + 1] = q; }
    return sym.nt;
}

int nlls_upload_structure(nlls_ctx* ctx, int64_t nvar, const int32_t* var_kind, const int32_t* var_dim, const uint64_t* blockindices,
                          int32_t ngroups, const nlls_cost_group* groups, int32_t flags) {
    if (!ctx || nvar < 0 || ngroups < 0 || (nvar && (!var_kind || !var_dim || !blockindices)) || (ngroups && !groups)) return NLLS_ERR_INVALID_ARG;
    try {
        (void)hipSetDevice(ctx->device);
        ctx->err_sub = NLLS_SUB_NONE;
        int rc = build_structure(ctx, nvar, var_kind, var_dim, blockindices, ngroups, groups, flags);
        // an eliminated block with more neighbours than the Schur kernels stage in LDS: solve the full system instead of failing
        // (decided on the sub-code the structure builder left, not on the error text; the retry declines by itself -- dense size
        // guard in build_schur -- when the full system is too large to factor densely)
        if (rc == NLLS_ERR_UNSUPPORTED && !(flags & NLLS_FLAG_NO_SCHUR) && ctx->err_sub == NLLS_SUB_SCHUR_SHAPE) {
            ctx->err_sub = NLLS_SUB_NONE;
            rc = build_structure(ctx, nvar, var_kind, var_dim, blockindices, ngroups, groups, flags | NLLS_FLAG_NO_SCHUR);
        }
        return rc;
    }
    catch (const std::exception& e) { return fail(ctx, NLLS_ERR_HIP, std::string("host exception: ") + e.what()); }
    catch (...) { return nlls::api_exception(ctx, "unknown C++ exception"); }
}

int nlls_get_info(const nlls_ctx* ctx, nlls_info* out) { NLLS_API_BEGIN
    if (!ctx || !out || !ctx->ready) return ctx ? NLLS_ERR_NOT_READY : NLLS_ERR_INVALID_ARG;
    *out = ctx->info; return NLLS_OK;
    NLLS_API_END(const_cast<nlls_ctx*>(ctx))
}

int nlls_get_bsm_index(const nlls_ctx* ctx, int64_t* colptr, int64_t* rowval, int64_t* nzval, int64_t* boffsets) { NLLS_API_BEGIN
    if (!ctx || !ctx->ready) return ctx ? NLLS_ERR_NOT_READY : NLLS_ERR_INVALID_ARG;
    if (ctx->info.is_sparse) {
        if (colptr) for (size_t i = 0; i < ctx->it_colptr.size(); ++i) colptr[i] = ctx->it_colptr[i] + 1;
        if (rowval) for (size_t i = 0; i < ctx->it_rowval.size(); ++i) rowval[i] = ctx->it_rowval[i] + 1;
        if (nzval) for (size_t i = 0; i < ctx->it_nzval.size(); ++i) nzval[i] = ctx->it_nzval[i] + 1;
    }
    if (boffsets) for (int64_t i = 0; i < ctx->info.nblocks; ++i) boffsets[i] = ctx->boffsets[i] + 1;
    return NLLS_OK;
    NLLS_API_END(const_cast<nlls_ctx*>(ctx))
}

int nlls_set_variables(nlls_ctx* ctx, int32_t which, const double* packed) { NLLS_API_BEGIN
    NEED_READY(); if (!valid_set(which) || !packed) return NLLS_ERR_INVALID_ARG;
    spec_note_write(ctx, which);
    if (which == NLLS_VARS_CURRENT) { ctx->sweeps_since_set = 0; ctx->tb_prev_end = 0.0; }         // (a new starting point: its first trial gets no look-ahead sweep, see nlls_sweep_gradhess)
    HIPCHK(hipMemcpyAsync(vars_ptr(ctx, which), packed, sizeof(double) * ctx->info.var_storage, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return NLLS_OK;
    NLLS_API_END(ctx)
}
int nlls_get_variables(nlls_ctx* ctx, int32_t which, double* packed) { NLLS_API_BEGIN
    NEED_READY(); if (!valid_set(which) || !packed) return NLLS_ERR_INVALID_ARG;
    HIPCHK(hipMemcpyAsync(packed, vars_ptr(ctx, which), sizeof(double) * ctx->info.var_storage, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return NLLS_OK;
    NLLS_API_END(ctx)
}
int nlls_swap_variables(nlls_ctx* ctx, int32_t a, int32_t b) { NLLS_API_BEGIN
    NEED_READY(); if (!valid_set(a) || !valid_set(b)) return NLLS_ERR_INVALID_ARG;
    std::swap(ctx->vars_slot[a], ctx->vars_slot[b]); return NLLS_OK;
    NLLS_API_END(ctx)
}
int nlls_copy_variables(nlls_ctx* ctx, int32_t dst, int32_t src) { NLLS_API_BEGIN
    NEED_READY(); if (!valid_set(dst) || !valid_set(src)) return NLLS_ERR_INVALID_ARG;
    if (dst != src) spec_note_write(ctx, dst);
    if (dst != src) HIPCHK(hipMemcpyAsync(vars_ptr(ctx, dst), vars_ptr(ctx, src), sizeof(double) * ctx->info.var_storage, hipMemcpyDeviceToDevice, ctx->stream));
    return NLLS_OK;
    NLLS_API_END(ctx)
}

// Lazy stage 0 (collective route): nlls_sweep_gradhess(ctx, NULL) leaves the reduced rows of A.data and b as this rank's share -- all an LM trial
// takes from them is linear in them (the reduced-reduced blocks and b_R enter [S | s], which is summed over ranks anyway; x'Hx and g'x are sums of
// the ranks' scalars), so the per-iteration all-reduce of [cost | reduced rows | reduced b] and its pack / unpack launches are skipped.  Whatever else
// reads them (the initial damping's max |diag|, nlls_get_grad, nlls_solve, ...) sums them first, here: a collective -- every rank makes the same calls.
static int ensure_reduced_summed(nlls_ctx* ctx) {
    if (ctx->reduced_summed) return NLLS_OK;
    ctx->reduced_summed = true;
    if (!ctx->reduce_fn || ctx->nranks <= 1) return NLLS_OK;
    ctx->n_stage0++;
    TRY(enqueue_pack_reduce0(ctx)); TRY(comm_reduce(ctx, ctx->redbuf.p, ctx->redbuf_len, NLLS_REDUCE_SUM)); TRY(enqueue_unpack_reduce0(ctx, false));
    return NLLS_OK;
}
int nlls_sweep_gradhess(nlls_ctx* ctx, double* cost_out) { NLLS_API_BEGIN
    NEED_READY();
    // (a sweep the caller asks for: the look-ahead may try again behind the next trial -- but not behind the FIRST trial from a new starting point: the initial damping
    //  (1e-6 of the largest diagonal entry, src/iterators.jl:131-137) is the one guess of the loop that is routinely rejected -- five times in a row at BASELINE config 5 --,
    //  and a look-ahead behind it is a sweep thrown away plus the current point swept again)
    ctx->spec_armed = ctx->sweeps_since_set >= 1; ctx->sweeps_since_set++;
    if (ctx->spec_pending) {
        const bool hit = !cost_out && !ctx->spec_stale && ctx->grad_phys == ctx->vars_slot[NLLS_VARS_CURRENT];
        ctx->spec_pending = false; ctx->spec_stale = false;
        if (hit) { ctx->spec_hits++; ctx->lambda = 0.0; ctx->have_grad = true; ctx->solved = false; ctx->reduced_summed = true; return NLLS_OK; }   // already enqueued behind the trial
        ctx->spec_misses++;
    }
    // Matrix-free LM trial (nlls_ctx::mf_ok): between two iterations nothing is enqueued here -- the first call that needs the linearisation says how much of it
    // (ensure_grad: nlls_lm_trial the reduced rows, everything else all of A.data and b), and it is formed then, at CURRENT.
    if (!cost_out && mf_trial(ctx, NLLS_VARS_CURRENT)) {
        ctx->grad_level = 0; ctx->grad_phys = ctx->vars_slot[NLLS_VARS_CURRENT];
        ctx->lambda = 0.0; ctx->have_grad = true; ctx->solved = false; ctx->reduced_summed = true; ctx->tE_valid = false; ctx->step_cached = false;
        return NLLS_OK;
    }
    // cost_out == NULL: the caller does not want the cost (the outer loop between iterations, src/optimize.jl:167-170
    // discards it) -- the sweep is then only enqueued: no partial-sum kernel, no synchronisation
    if (ctx->reduce_fn) {
        // collective (include/nlls_amd.h): this rank's blocks, then ONE sum over ranks of [cost | reduced rows of A.data | reduced part of b]
        const bool lazy = !cost_out && ctx->lazy_stage0 && ctx->info.is_sparse && !ctx->elim_slab;
        if (ctx->phase_on && ctx->phase_ev.size() >= 8) {      // (the previous sweep's pair has long completed: the trials between synchronise)
            float ms = 0.f; if (ctx->phase_sweeps >= 0 && hipEventElapsedTime(&ms, ctx->phase_ev[6], ctx->phase_ev[7]) == hipSuccess) { ctx->phase_ms[5] += ms; ctx->phase_sweeps++; } else (void)hipGetLastError();
            (void)hipEventRecord(ctx->phase_ev[6], ctx->stream);
        }
        TRY(enqueue_sweep_gradhess(ctx, !lazy));
        if (ctx->phase_on && ctx->phase_ev.size() >= 8) (void)hipEventRecord(ctx->phase_ev[7], ctx->stream);
        ctx->lambda = 0.0; ctx->have_grad = true; ctx->solved = false;
        ctx->reduced_summed = !lazy;
        if (lazy) return NLLS_OK;                                            // nothing is summed now: ensure_reduced_summed, or never
        if (ctx->nranks > 1) { ctx->n_stage0++; TRY(enqueue_pack_reduce0(ctx)); TRY(comm_reduce(ctx, ctx->redbuf.p, ctx->redbuf_len, NLLS_REDUCE_SUM)); TRY(enqueue_unpack_reduce0(ctx, true)); }
        else TRY(comm_reduce(ctx, ctx->scalars.p, 1, NLLS_REDUCE_SUM));      // (one rank through the route: the same number of collectives)
        if (!cost_out) return NLLS_OK;
        TRY(fetch_scalars(ctx, 0, 1)); *cost_out = ctx->h_scalars[0];
        return NLLS_OK;
    }
    // (where the matrix-free trial applies the cost is ITS sum of the blocks -- the one nlls_sweep_cost and every trial report: one fixed sum, bit for bit the same everywhere)
    const bool mfcost = cost_out && mf_trial(ctx, NLLS_VARS_CURRENT);
    TRY(enqueue_sweep_gradhess(ctx, cost_out != nullptr && !mfcost));
    if (mfcost) TRY(enqueue_mf_sweep_cost(ctx, NLLS_VARS_CURRENT));
    ctx->lambda = 0.0; ctx->have_grad = true; ctx->solved = false; ctx->reduced_summed = true;
    if (!cost_out) return NLLS_OK;
    TRY(fetch_scalars(ctx, 0, 1));
    *cost_out = ctx->h_scalars[0];
    return NLLS_OK;
    NLLS_API_END(ctx)
}
int nlls_sweep_cost(nlls_ctx* ctx, int32_t which, double* cost_out) { NLLS_API_BEGIN
    NEED_READY(); if (!valid_set(which)) return NLLS_ERR_INVALID_ARG;
    // (where the matrix-free trial applies, every cost is ONE fixed sum -- the trial's: cost(problem) == result.bestcost bit for bit)
    if (mf_trial(ctx, NLLS_VARS_CURRENT)) TRY(enqueue_mf_sweep_cost(ctx, which)); else
    TRY(enqueue_sweep_cost(ctx, which));
    TRY(comm_reduce(ctx, ctx->scalars.p, 1, NLLS_REDUCE_SUM));       // (collective mode: the ranks' partial costs)
    TRY(fetch_scalars(ctx, 0, 1));
    if (cost_out) *cost_out = ctx->h_scalars[0];
    return NLLS_OK;
    NLLS_API_END(ctx)
}

int nlls_get_grad(nlls_ctx* ctx, double* b_out) { NLLS_API_BEGIN
    NEED_GRAD(); if (!b_out) return NLLS_ERR_INVALID_ARG;
    HIPCHK(hipMemcpyAsync(b_out, ctx->b.p, sizeof(double) * ctx->info.ndof, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return NLLS_OK;
    NLLS_API_END(ctx)
}
int nlls_get_bsm_data(nlls_ctx* ctx, double* data_out) { NLLS_API_BEGIN
    NEED_GRAD(); if (!data_out) return NLLS_ERR_INVALID_ARG;
    HIPCHK(hipMemcpyAsync(data_out, ctx->A.p, sizeof(double) * ctx->info.nnz_data, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return NLLS_OK;
    NLLS_API_END(ctx)
}
int nlls_max_abs_diag(nlls_ctx* ctx, double* out) { NLLS_API_BEGIN
    NEED_GRAD(); TRY(enqueue_max_abs_diag(ctx)); TRY(comm_reduce(ctx, ctx->scalars.p + 3, 1, NLLS_REDUCE_MAX)); TRY(fetch_scalars(ctx, 3, 1));
    if (out) *out = ctx->h_scalars[3]; return NLLS_OK;
    NLLS_API_END(ctx)
}
int nlls_grad_sqnorm(nlls_ctx* ctx, double* out) { NLLS_API_BEGIN
    NEED_GRAD(); TRY(enqueue_quadform(ctx, ctx->b.p, 6)); TRY(comm_reduce(ctx, ctx->scalars.p + 6, 2, NLLS_REDUCE_SUM)); TRY(fetch_scalars(ctx, 6, 2));
    if (out) *out = ctx->h_scalars[7]; return NLLS_OK;
    NLLS_API_END(ctx)
}
int nlls_grad_quadform(nlls_ctx* ctx, double* out) { NLLS_API_BEGIN
    NEED_GRAD(); TRY(enqueue_quadform(ctx, ctx->b.p, 6)); TRY(comm_reduce(ctx, ctx->scalars.p + 6, 2, NLLS_REDUCE_SUM)); TRY(fetch_scalars(ctx, 6, 2));
    if (out) *out = ctx->h_scalars[6]; return NLLS_OK;
    NLLS_API_END(ctx)
}

// what a non-zero factorisation status means: a bad pivot (NLLS_ERR_NOT_SPD: the iterators damp more and retry) or a hand-off inside a one-launch backward pass that
// timed out (0x40000000, nlls_bcr.hip: not a property of the system -- NLLS_ERR_HIP; NLLS_BCR_LEVEL_BACKWARD=1 / NLLS_DENSE_STEP_BACKWARD=1 select the per-level launches)
static int status_error(nlls_ctx* ctx, int32_t st, const char* what) {
    if (st == 0x40000000) return fail(ctx, NLLS_ERR_HIP, "a hand-off between workgroups of the one-launch backward substitution timed out (0.5 s): not a pivot failure -- NLLS_BCR_LEVEL_BACKWARD=1 / NLLS_DENSE_STEP_BACKWARD=1 select the per-level launches");
    return fail(ctx, NLLS_ERR_NOT_SPD, std::string(what) + " (code " + std::to_string(st) + ")");
}
int nlls_damp(nlls_ctx* ctx, double delta) { NLLS_API_BEGIN NEED_GRAD_LAZY(); ctx->lambda += delta; return NLLS_OK; NLLS_API_END(ctx) }

int nlls_solve(nlls_ctx* ctx, double* x_out) { NLLS_API_BEGIN
    NEED_GRAD();
    TRY(enqueue_solve(ctx));
    // what the iterators ask about the step next -- max |x|, |x|, x'Hx, g'x (src/optimize.jl:149, src/iterators.jl:160-163) --
    // is computed behind the solve and comes back with the same synchronisation; the queries then answer from the host
    ctx->step_cached = false;
    const bool precompute = ctx->nranks == 1;
    if (precompute) { TRY(enqueue_post_solve(ctx)); }
    int32_t status[4] = {0, 0, 0, 0};
    if (precompute) HIPCHK(hipMemcpyAsync(ctx->h_scalars + 1, ctx->scalars.p + 1, sizeof(double) * 10, hipMemcpyDeviceToHost, ctx->stream));   // ... and the status in [10]
    else HIPCHK(hipMemcpyAsync(status, ctx->d_status.p, sizeof(status), hipMemcpyDeviceToHost, ctx->stream));
    if (x_out) HIPCHK(hipMemcpyAsync(x_out, ctx->x.p, sizeof(double) * ctx->info.ndof, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    ctx->solved = true;
    if (precompute) { status[0] = (int32_t)ctx->h_scalars[10]; ctx->step_cached = true; ctx->c_maxabs = ctx->h_scalars[1]; ctx->c_sumsq = ctx->h_scalars[2]; ctx->c_gx = ctx->h_scalars[5]; ctx->c_xAx = ctx->h_scalars[8]; ctx->c_xx = ctx->h_scalars[9]; }
    if (status[0] != 0) return status_error(ctx, status[0], "factorisation met a non-positive pivot");
    return NLLS_OK;
    NLLS_API_END(ctx)
}
// One Levenberg-Marquardt trial in one call and one synchronisation (src/iterators.jl:149-157): uniformscaling!(H, dlambda),
// solve!, negate!, update!(to, from, x), cost(to).  Same kernels, same order as the separate entry points.
int nlls_lm_trial(nlls_ctx* ctx, double dlambda, int32_t to, int32_t from, double* cost_out) { NLLS_API_BEGIN
    NEED_GRAD_LAZY(); if (!valid_set(to) || !valid_set(from) || to == from) return NLLS_ERR_INVALID_ARG;
    const bool mf = mf_trial(ctx, from);           // matrix-free: the eliminated rows of A.data are neither needed nor formed (nlls_mf.hip)
    TRY(ensure_grad(ctx, mf ? 1 : 2));             // (a trial right behind a REJECTED one: the look-ahead sweep of that trial's point is in A and b)
    const bool collective = ctx->reduce_fn != nullptr && ctx->info.is_sparse && !ctx->replicated;
    if (!collective) TRY(ensure_reduced_summed(ctx));
    if (ctx->nranks != 1 && !collective) return fail(ctx, NLLS_ERR_UNSUPPORTED, "nlls_lm_trial under nlls_set_shard needs an all-reduce (nlls_comm_init_rccl / nlls_set_allreduce), or the *_local / *_finish pairs");
    ctx->lambda += dlambda;
    ctx->step_cached = false;
    if (collective) {
        // the sharded trial, end to end on this rank's stream: local elimination, ONE sum of [S | s] over ranks, the reduced system solved on
        // every rank (each then holds the reduced part of the step: x is never summed), own back-substitution, retraction, own cost blocks,
        // and one gather of the ranks' scalars -- combined on the device and published to the host mirror as the single-GPU trial does
        if (ctx->nranks > 1 && !ctx->reduced_summed) ctx->n_lazy_trials++;
        phase_mark(ctx, 0);
        TRY(enqueue_solve_local(ctx));
        phase_mark(ctx, 1);
        if (ctx->solve_mode == SOLVE_TSPARSE && ctx->tsp.nslots_assembled < ctx->tsp.nslots) {
            // tile-sparse layout [assembled tiles | fill tiles | strips | s]: the fill tiles and the strips are zero on every rank until the factorisation -- the assembled
            // prefix and s are what is summed (two collectives: on the 100 x 100 camera grid most of the volume was zeros)
            TRY(comm_reduce(ctx, ctx->S.p, ctx->tsp.nslots_assembled * (int64_t)TSP_TE, NLLS_REDUCE_SUM));
            TRY(comm_reduce(ctx, ctx->S.p + ctx->s_elems, ctx->nred, NLLS_REDUCE_SUM));
        } else TRY(comm_reduce(ctx, ctx->S.p, (int64_t)ctx->s_elems + ctx->nred, NLLS_REDUCE_SUM));
        phase_mark(ctx, 2);
        if (ctx->nranks == 1) { ctx->trial_to = to; ctx->trial_from = from; }      // (one rank through the route: the same launches as the unsharded trial)
        ctx->replicate_xr = true; int rc = enqueue_solve_finish(ctx); ctx->replicate_xr = false; ctx->trial_to = ctx->trial_from = -1; TRY(rc);
        phase_mark(ctx, 4);
        double* const mirror = ctx->h_scalars_dev; ctx->h_scalars_dev = nullptr;           // (the rank's own scalars are not what the host waits for)
        rc = enqueue_lm_trial_tail(ctx, to, from); ctx->h_scalars_dev = mirror; TRY(rc);
        TRY(comm_gather_trial_scalars(ctx, (double)ctx->trial_seq));
        phase_mark(ctx, 5);
    } else if (ctx->tiny_dense) {
        const bool la = ctx->spec_on && ctx->spec_armed && from == NLLS_VARS_CURRENT;
        TRY(enqueue_tiny_dense_trial(ctx, to, from, la && ctx->tiny_fin_role));
        if (la) { TRY(enqueue_sweep_gradhess(ctx, false, to)); ctx->spec_pending = true; ctx->spec_stale = false; }
        TRY(enqueue_tiny_trial_finish_pending(ctx));      // (no accumulate launch took the finishing reduction along)
    } else {
    { double* st = ctx->h_scalars + 40; st[0] = st[1] = st[2] = st[3] = 0.0; }       // (the launches of this trial stamp them: device-timed buckets)
    ctx->trial_to = to; ctx->trial_from = from;    // (the back-substitution launch may take the retraction with it: enqueue_solve_finish)
    ctx->mf_use = mf; if (mf) ctx->mf_trials++;
    { const int rc = enqueue_solve(ctx); ctx->trial_to = ctx->trial_from = -1; ctx->mf_use = false; TRY(rc); }
    const bool la = ctx->spec_on && ctx->spec_armed && ctx->nranks == 1 && ctx->info.is_sparse && from == NLLS_VARS_CURRENT;
    ctx->tail_zero_for_lookahead = la; ctx->heavy_rows_zeroed = false;
    // (matrix-free trial with a look-ahead sweep behind it: the trial's finishing workgroup rides in that sweep's launch -- nlls_sweep.hip -- unless rows have to be zeroed in between)
    ctx->mf_fin_defer = mf && la && ctx->nzero == 0; ctx->mf_fin_pending = false;
    { const int rc = enqueue_lm_trial_tail(ctx, to, from); ctx->tail_zero_for_lookahead = false; ctx->mf_fin_defer = false; TRY(rc); }     // step statistics + quadratic form (+ retraction) and the cost sweep, one finishing launch
    // the look-ahead sweep: the gradient sweep of the trial point, enqueued behind the finishing launch (the host reads the trial's scalars while it runs)
    if (la) { const int rc = enqueue_sweep_gradhess(ctx, false, to, mf ? 1 : 0); ctx->heavy_rows_zeroed = false; TRY(rc); ctx->spec_pending = true; ctx->spec_stale = false; }
    if (ctx->mf_fin_pending) TRY(enqueue_mf_trial_finish_now(ctx));      // (no launch of the sweep took the finishing workgroup along)
    }
    // (sparse systems: the finishing launch has written the scalars -- in [10] the factorisation status -- to the pinned host mirror itself)
    if ((!ctx->info.is_sparse && !ctx->tiny_dense) || !ctx->h_scalars_dev) {
        HIPCHK(hipMemcpyAsync(ctx->h_scalars, ctx->scalars.p, sizeof(double) * 12, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
    } else {
        // spin on the sequence numbers the finishing launch publishes (a few milliseconds at most: then fall back to the synchronisation,
        // after which the values are there in any case)
        volatile double* hs = ctx->h_scalars; const double seq = (double)ctx->trial_seq;
        const auto t0 = std::chrono::steady_clock::now(); bool seen = false;
        for (uint64_t spin = 0;; ++spin) {
            if (hs[32] == seq && hs[33] == seq) { seen = true; break; }
            // (a trial of the problems this path is built for ends within a millisecond; a long one sleeps in the synchronisation instead of
            //  burning a core)
            if ((spin & 255) == 255) { if (std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(1500)) break; __builtin_ia32_pause(); }
        }
        std::atomic_thread_fence(std::memory_order_acquire);
        if (!seen) HIPCHK(hipStreamSynchronize(ctx->stream));
    }
    {   // device-timed buckets (nlls_ctx::tb_*): only a trial whose launches all stamped (the sparse single-GPU routes) counts
        const double* st = ctx->h_scalars + 40;
        if (!collective && st[0] > 0.0 && st[2] >= st[0] && st[3] >= st[2]) {
            const double k = ctx->tb_ns_per_tick;
            ctx->tb_solver_ns += (int64_t)((st[2] - st[0]) * k); ctx->tb_cost_ns += (int64_t)((st[3] - st[2]) * k); ctx->tb_trials++;
            if (ctx->tb_prev_end > 0.0 && st[0] >= ctx->tb_prev_end) ctx->tb_grad_ns += (int64_t)((st[0] - ctx->tb_prev_end) * k);
            ctx->tb_prev_end = st[3];
        }
    }
    if (collective && ctx->phase_on && ctx->phase_ev.size() >= 6) {      // (the trial has ended: every event has completed)
        if (hipEventSynchronize(ctx->phase_ev[5]) == hipSuccess) {
            float ms = 0.f; bool ok = true; double d[5];
            for (int k = 0; k < 5 && ok; ++k) { ok = hipEventElapsedTime(&ms, ctx->phase_ev[k], ctx->phase_ev[k + 1]) == hipSuccess; d[k] = ms; }
            if (ok) { for (int k = 0; k < 5; ++k) ctx->phase_ms[k] += d[k]; ctx->phase_trials++; } else (void)hipGetLastError();
        }
    }
    const int32_t status[1] = {(int32_t)ctx->h_scalars[10]};
    ctx->comm_gathered = collective;
    if (collective) ctx->comm_agreed = ctx->h_scalars[11];      // the ranks' posted termination flags, combined by maximum in the same gather (nlls_comm_agreed_flag)
    ctx->status_known_zero = status[0] == 0;       // (nothing has touched the device's status word since: the next solve need not reset it)
    ctx->solved = true;
    ctx->step_cached = true; ctx->c_maxabs = ctx->h_scalars[1]; ctx->c_sumsq = ctx->h_scalars[2]; ctx->c_gx = ctx->h_scalars[5]; ctx->c_xAx = ctx->h_scalars[8]; ctx->c_xx = ctx->h_scalars[9];
    if (status[0] != 0) return status_error(ctx, status[0], "factorisation met a non-positive pivot");
    if (cost_out) *cost_out = ctx->h_scalars[0];
    return NLLS_OK;
    NLLS_API_END(ctx)
}
// The part of one Levenberg-Marquardt trial that follows the solve (src/iterators.jl:155-163), for sharded runs: with the
// step x complete on this rank (after the stage-2 reduction) -- update!(to, from, x), cost(to), fast_bAb(H, x), dot(g, x),
// maximum(abs, x), |x|^2 -- in one enqueue and one synchronisation.  out = [cost, x'Hx, g'x, max|x|, |x|^2]; under
// nlls_set_shard the first three are this rank's PARTIAL sums (the caller adds them over ranks), the last two are global.
int nlls_trial_local(nlls_ctx* ctx, int32_t to, int32_t from, double* out) { NLLS_API_BEGIN
    NEED_GRAD(); if (!valid_set(to) || !valid_set(from) || to == from) return NLLS_ERR_INVALID_ARG;
    TRY(enqueue_lm_trial_tail(ctx, to, from));     // step statistics + quadratic form + retraction in one launch, the cost sweep, one finishing launch
    if (!out) return NLLS_OK;                      // enqueue only: the eleven scalars stay on the device (reduce buffer 3) for a device-side gather
    HIPCHK(hipMemcpyAsync(ctx->h_scalars, ctx->scalars.p, sizeof(double) * 11, hipMemcpyDeviceToHost, ctx->stream));   // [10]: the factorisation status
    HIPCHK(hipStreamSynchronize(ctx->stream));
    out[0] = ctx->h_scalars[0]; out[1] = ctx->h_scalars[8]; out[2] = ctx->h_scalars[5]; out[3] = ctx->h_scalars[1]; out[4] = ctx->h_scalars[2];
    if (ctx->nranks > 1) {
        // sharded: max|x| and |x|^2 over THIS rank's share of the step (its own eliminated blocks; rank 0 also the reduced part), and the
        // factorisation status as a sixth value instead of an error -- a rank-local zero pivot must not leave this rank out of the
        // collective its peers are about to enter: the caller reduces all six and raises on every rank
        out[4] = ctx->h_scalars[9]; out[5] = ctx->h_scalars[10];
        return NLLS_OK;
    }
    if ((int32_t)ctx->h_scalars[10] != 0) return status_error(ctx, (int32_t)ctx->h_scalars[10], "factorisation met a zero pivot");
    return NLLS_OK;
    NLLS_API_END(ctx)
}
// nlls_solve_finish_async for a sharded LM trial: the reduced part of the step stays on every rank (no stage-2 reduction afterwards)
int nlls_solve_finish_replicated(nlls_ctx* ctx) { NLLS_API_BEGIN
    NEED_GRAD(); ctx->replicate_xr = true; const int rc = enqueue_solve_finish(ctx); ctx->replicate_xr = false;
    if (rc != NLLS_OK) return rc;
    ctx->solved = true; ctx->step_cached = false; return NLLS_OK;
    NLLS_API_END(ctx)
}
// this rank's share of a variable set: its own eliminated blocks' variables, on rank 0 also everything else; zeros elsewhere --
// the sum over ranks is the complete set (what a sharded optimisation hands back at the end)
int nlls_get_variables_owned(nlls_ctx* ctx, int32_t which, double* packed) { NLLS_API_BEGIN
    TRY(nlls_get_variables(ctx, which, packed));
    if (ctx->nranks == 1) return NLLS_OK;
    for (int64_t i = 0; i < ctx->info.nvar; ++i) {
        const uint64_t bi = ctx->blockindices[i];
        int owner = 0;
        if (bi > 0 && ctx->is_elim.size() >= bi && ctx->is_elim[bi - 1] && !ctx->owner_of_block.empty()) owner = ctx->owner_of_block[bi - 1];
        if (owner != ctx->rank) for (uint32_t q = ctx->var_off[i]; q < ctx->var_off[i + 1]; ++q) packed[q] = 0.0;
    }
    return NLLS_OK;
    NLLS_API_END(ctx)
}
// optimizesingles!(problem, options, indices)  src/optimize.jl:60-76,183-205
int nlls_optimize_singles(nlls_ctx* ctx, int64_t nsel, const int64_t* varindices, const int64_t* cptr, const int32_t* cgroup, const int64_t* cindex,
                          const int32_t* cslot, int32_t iterator, int32_t maxiters, int32_t maxfails, double reldcost, double absdcost, double dstep, int64_t* iters_out) {
    NEED_READY();
    if (iterator < 0 || iterator > 3) return fail(ctx, NLLS_ERR_INVALID_ARG, "nlls_optimize_singles: iterator must be 0 (Newton), 1 (Levenberg-Marquardt), 2 (dogleg) or 3 (gradient descent)");
    if (nsel < 0 || (nsel > 0 && (!varindices || !cptr))) return NLLS_ERR_INVALID_ARG;
    // Under nlls_set_shard (round 5; /root/reference/src/optimize.jl:60-76 is embarrassingly parallel per variable): every rank passes the SAME lists -- the caller's
    // variable and cost-block numbering -- and relaxes the variables whose cost blocks it owns (an eliminated variable's blocks all live on the rank that owns it:
    // build_structure); the results are gathered with the installed all-reduce, once.  A variable whose blocks are spread over ranks (a camera) is declined on every rank.
    const bool sharded = ctx->nranks > 1;
    if (sharded && (!ctx->reduce_fn || ctx->presharded)) return fail(ctx, NLLS_ERR_UNSUPPORTED, "nlls_optimize_singles under nlls_set_shard needs an installed all-reduce and a library-partitioned upload (not NLLS_FLAG_PRESHARDED)");
    if (nsel == 0) return NLLS_OK;
    const int64_t nc = cptr[nsel];
    if (cptr[0] != 0 || nc < 0 || (nc > 0 && (!cgroup || !cindex || !cslot))) return NLLS_ERR_INVALID_ARG;
    for (int64_t i = 0; i < nsel; ++i) {
        const int64_t v = varindices[i] - 1;
        if (v < 0 || v >= ctx->info.nvar || cptr[i + 1] < cptr[i]) return fail(ctx, NLLS_ERR_INVALID_ARG, "nlls_optimize_singles: bad variable index or cost list");
        if (var_dof(ctx->var_kind[v], ctx->var_dim[v]) > 6) return fail(ctx, NLLS_ERR_UNSUPPORTED, "nlls_optimize_singles: variable with more than 6 degrees of freedom");
    }
    // this rank's share: the variables all of whose blocks are local, with the blocks' LOCAL indices
    std::vector<int64_t> sel, cp(1, 0), pos; std::vector<uint32_t> cidx; std::vector<int32_t> cg, cs; double spread = 0.0;
    sel.reserve((size_t)nsel); cidx.reserve((size_t)nc); cg.reserve((size_t)nc); cs.reserve((size_t)nc);
    for (int64_t i = 0; i < nsel; ++i) {
        int64_t mine = 0; const size_t mark = cidx.size();
        for (int64_t e = cptr[i]; e < cptr[i + 1]; ++e) {
            if (cgroup[e] < 0 || cgroup[e] >= (int32_t)ctx->groups.size()) return fail(ctx, NLLS_ERR_INVALID_ARG, "nlls_optimize_singles: bad cost group");
            const Group& G = ctx->groups[cgroup[e]];
            const int64_t nglob = G.local_of.empty() ? G.ncost : (int64_t)G.local_of.size();
            if (cindex[e] < 0 || cindex[e] >= nglob || cslot[e] < 0 || cslot[e] >= G.ndeps) return fail(ctx, NLLS_ERR_INVALID_ARG, "nlls_optimize_singles: bad cost index or slot");
            if (G.adaptive && cslot[e] == 0) return fail(ctx, NLLS_ERR_UNSUPPORTED, "nlls_optimize_singles: the adaptive kernel variable cannot be optimised on its own");
            if (is_dyn_kind(G.res_kind)) return fail(ctx, NLLS_ERR_UNSUPPORTED, "nlls_optimize_singles: dynamic-size cost blocks are not handled by the per-variable kernel");
            const int64_t loc = G.local_of.empty() ? cindex[e] : (int64_t)G.local_of[(size_t)cindex[e]];
            if (loc >= 0) { ++mine; cidx.push_back((uint32_t)loc); cg.push_back(cgroup[e]); cs.push_back(cslot[e]); }
        }
        const int64_t all = cptr[i + 1] - cptr[i];
        // a variable without any block is relaxed (trivially) by rank 0
        if (mine == all && (all > 0 || ctx->rank == 0 || !sharded)) { sel.push_back(varindices[i] - 1); pos.push_back(i); cp.push_back((int64_t)cidx.size()); }
        else { cidx.resize(mark); cg.resize(mark); cs.resize(mark); if (mine != 0) spread = 1.0; }
    }
    if (sharded) {          // a variable whose blocks are spread over ranks: declined everywhere (one small collective, so that no rank is left in the gather below)
        DevBuf<double> df; HIPCHK(df.alloc(1)); HIPCHK(hipMemcpyAsync(df.p, &spread, sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        TRY(comm_reduce(ctx, df.p, 1, NLLS_REDUCE_MAX));
        HIPCHK(hipMemcpyAsync(&spread, df.p, sizeof(double), hipMemcpyDeviceToHost, ctx->stream)); HIPCHK(hipStreamSynchronize(ctx->stream));
        if (spread != 0.0) return fail(ctx, NLLS_ERR_UNSUPPORTED, "nlls_optimize_singles under nlls_set_shard: a listed variable's cost blocks are spread over ranks (only variables whose blocks one rank owns -- the eliminated ones -- are relaxed in parallel)");
    }
    const int64_t nloc = (int64_t)sel.size();
    std::vector<unsigned char> gbuf(singles_group_size() * ctx->groups.size());
    for (size_t g = 0; g < ctx->groups.size(); ++g) singles_group_fill(gbuf.data() + g * singles_group_size(), ctx->groups[g]);
    DevBuf<int64_t> d_sel, d_cptr, d_iters, d_pos, d_all; DevBuf<int32_t> d_cgroup, d_cslot; DevBuf<uint32_t> d_cidx; DevBuf<unsigned char> d_groups;
    if (cidx.empty()) { cidx.push_back(0); cg.push_back(0); cs.push_back(0); }
    if (nloc > 0) { HIPCHK(d_sel.upload(sel)); HIPCHK(d_cptr.upload(cp)); HIPCHK(d_cgroup.upload(cg)); HIPCHK(d_cslot.upload(cs)); HIPCHK(d_cidx.upload(cidx)); HIPCHK(d_iters.alloc((size_t)nloc)); }
    HIPCHK(d_groups.upload(gbuf));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    ctx->have_grad = false; ctx->solved = false; ctx->step_cached = false; ctx->tE_valid = false;   // the variables change under the linear system
    spec_note_write(ctx, NLLS_VARS_CURRENT); ctx->spec_pending = false; ctx->spec_stale = false; ctx->grad_level = 0;   // (... and under a look-ahead sweep of this very set: its A and b are of the point before the relaxation)
    if (nloc > 0) TRY(enqueue_optimize_singles(ctx, nloc, d_sel.p, d_cptr.p, d_cgroup.p, d_cidx.p, d_cslot.p, d_groups.p, iterator, maxiters, maxfails, reldcost, absdcost, dstep, d_iters.p));
    if (!sharded) {
        if (iters_out) HIPCHK(hipMemcpyAsync(iters_out, d_iters.p, sizeof(int64_t) * nsel, hipMemcpyDeviceToHost, ctx->stream));      // (unsharded: nloc == nsel, the caller's order)
        HIPCHK(hipStreamSynchronize(ctx->stream));
        return NLLS_OK;
    }
    // the gather: [storage of the variables this rank relaxed, zero elsewhere | their iteration counts at the caller's positions] summed over ranks; then every listed
    // variable's storage is taken from the sum -- exact: each entry has one non-zero contribution
    const size_t nst = (size_t)ctx->info.var_storage; DevBuf<double> gb; HIPCHK(gb.alloc(nst + (size_t)nsel));
    HIPCHK(hipMemsetAsync(gb.p, 0, sizeof(double) * (nst + (size_t)nsel), ctx->stream));
    double* cur = vars_ptr(ctx, NLLS_VARS_CURRENT);
    if (nloc > 0) { HIPCHK(d_pos.upload(pos)); TRY(enqueue_copy_var_storage(ctx, d_sel.p, nloc, cur, gb.p)); TRY(enqueue_iters_to_double(ctx, d_iters.p, d_pos.p, nloc, gb.p + nst)); }
    TRY(comm_reduce(ctx, gb.p, (int64_t)(nst + (size_t)nsel), NLLS_REDUCE_SUM));
    { std::vector<int64_t> all((size_t)nsel); for (int64_t i = 0; i < nsel; ++i) all[(size_t)i] = varindices[i] - 1; HIPCHK(d_all.upload(all)); }
    TRY(enqueue_copy_var_storage(ctx, d_all.p, nsel, gb.p, cur));
    std::vector<double> hit((size_t)nsel);
    HIPCHK(hipMemcpyAsync(hit.data(), gb.p + nst, sizeof(double) * (size_t)nsel, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    if (iters_out) for (int64_t i = 0; i < nsel; ++i) iters_out[i] = (int64_t)hit[(size_t)i];
    return NLLS_OK;
}
int nlls_get_time_buckets(nlls_ctx* ctx, int64_t* out, int32_t n) { NLLS_API_BEGIN
    if (!ctx || !out || n < 1) return NLLS_ERR_INVALID_ARG;
    const int64_t v[4] = {ctx->tb_grad_ns, ctx->tb_cost_ns, ctx->tb_solver_ns, ctx->tb_trials};
    for (int i = 0; i < n && i < 4; ++i) out[i] = v[i];
    return NLLS_OK;
    NLLS_API_END(ctx)
}
int nlls_get_phase_times(nlls_ctx* ctx, double* out, int32_t n) { NLLS_API_BEGIN
    if (!ctx || !out || n < 1) return NLLS_ERR_INVALID_ARG;
    const double v[8] = {ctx->phase_ms[0], ctx->phase_ms[1], ctx->phase_ms[2], ctx->phase_ms[3], ctx->phase_ms[4], ctx->phase_ms[5], (double)ctx->phase_trials, (double)ctx->phase_sweeps};
    for (int i = 0; i < n && i < 8; ++i) out[i] = v[i];
    return NLLS_OK;
    NLLS_API_END(ctx)
}
int nlls_set_option(nlls_ctx* ctx, int32_t option, int64_t value) { NLLS_API_BEGIN
    if (!ctx) return NLLS_ERR_INVALID_ARG;
    switch (option) {
    case NLLS_OPT_MATERIALIZE: ctx->mf_on = value == 0; return NLLS_OK;          // (takes effect with the next nlls_lm_trial / nlls_sweep_gradhess; what A and b hold is tracked either way)
    case NLLS_OPT_LOOKAHEAD:   ctx->spec_on = value != 0; return NLLS_OK;
    case NLLS_OPT_PHASE_EVENTS:
        ctx->phase_on = value != 0; (void)hipSetDevice(ctx->device);
        if (ctx->phase_on && ctx->phase_ev.empty()) { ctx->phase_ev.resize(8); for (auto& e : ctx->phase_ev) if (hipEventCreate(&e) != hipSuccess) return NLLS_ERR_HIP; }
        if (ctx->phase_on) { for (double& v : ctx->phase_ms) v = 0.0; ctx->phase_trials = ctx->phase_sweeps = 0; }
        return NLLS_OK;
    }
    return fail(ctx, NLLS_ERR_INVALID_ARG, "nlls_set_option: unknown option");
    NLLS_API_END(ctx)
}
int nlls_get_solve_stats(nlls_ctx* ctx, int64_t* out, int32_t n) { NLLS_API_BEGIN
    NEED_READY(); if (!out || n < 1) return NLLS_ERR_INVALID_ARG;
    int32_t status[16] = {0};
    HIPCHK(hipMemcpyAsync(status, ctx->d_status.p, sizeof(status), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    const int64_t vals[27] = {status[0], (int64_t)status[2] << 10, (int64_t)status[3] << 10, ctx->solve_mode, ctx->nelim_groups, ctx->bw,
                              ctx->bcr.ready ? ctx->bcr.mfma_issued : 0, ctx->bcr.ready ? ctx->bcr.launches : 0, ctx->bcr.ready ? (int64_t)ctx->bcr.levels.size() : 0, ctx->n_band,
                              status[4] /* pivots the floor of the last undamped band solve dropped */, ctx->n_stage0, ctx->n_lazy_trials, ctx->red_reordered, ctx->bw_caller, ctx->dense_window ? 1 : 0,
                              ctx->tsp.ready ? ctx->tsp.nt : 0, ctx->tsp.ready ? (int64_t)ctx->tsp.levels.size() : 0, ctx->tsp.ready ? ctx->tsp.nslots : 0, ctx->tsp.ready ? ctx->tsp.launches : 0, ctx->tsp.ready ? ctx->tsp.products : 0,
                              ctx->spec_hits, ctx->spec_misses /* look-ahead sweeps used / thrown away */,
                              ctx->mf_trials, ctx->mf_reduced_sweeps, ctx->full_sweeps /* matrix-free LM trials, sweeps of the reduced rows only, full accumulate sweeps since the upload */,
                              ctx->bcr.ready ? 16 * ctx->bcr.NT : 0 /* unknowns per block of the block cyclic reduction */};
    for (int i = 0; i < n && i < 27; ++i) out[i] = vals[i];
    return NLLS_OK;
    NLLS_API_END(ctx)
}
int nlls_set_step(nlls_ctx* ctx, const double* x) { NLLS_API_BEGIN
    NEED_READY(); if (!x) return NLLS_ERR_INVALID_ARG;
    ctx->tE_valid = false; ctx->step_cached = false;   // the step is no longer the one the last solve produced
    HIPCHK(hipMemcpyAsync(ctx->x.p, x, sizeof(double) * ctx->info.ndof, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return NLLS_OK;
    NLLS_API_END(ctx)
}
int nlls_get_step(nlls_ctx* ctx, double* x_out) { NLLS_API_BEGIN
    NEED_READY(); if (!x_out) return NLLS_ERR_INVALID_ARG;
    HIPCHK(hipMemcpyAsync(x_out, ctx->x.p, sizeof(double) * ctx->info.ndof, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return NLLS_OK;
    NLLS_API_END(ctx)
}
int nlls_step_maxabs(nlls_ctx* ctx, double* out) { NLLS_API_BEGIN
    NEED_READY(); if (ctx->step_cached) { if (out) *out = ctx->c_maxabs; return NLLS_OK; }
    TRY(enqueue_step_stats(ctx)); TRY(fetch_scalars(ctx, 1, 2));
    if (out) *out = ctx->h_scalars[1]; return NLLS_OK;
    NLLS_API_END(ctx)
}
int nlls_step_norm(nlls_ctx* ctx, double* out) { NLLS_API_BEGIN
    NEED_READY(); if (ctx->step_cached) { if (out) *out = std::sqrt(ctx->c_sumsq); return NLLS_OK; }
    TRY(enqueue_step_stats(ctx)); TRY(fetch_scalars(ctx, 1, 2));
    if (out) *out = std::sqrt(ctx->h_scalars[2]); return NLLS_OK;
    NLLS_API_END(ctx)
}
int nlls_quadform(nlls_ctx* ctx, double* xHx_out, double* gx_out) { NLLS_API_BEGIN
    NEED_GRAD_LAZY();
    if (ctx->step_cached) { if (xHx_out) *xHx_out = ctx->c_xAx + ctx->lambda * ctx->c_xx; if (gx_out) *gx_out = ctx->c_gx; return NLLS_OK; }   // damping may have changed since
    TRY(ensure_grad(ctx, 2)); TRY(ensure_reduced_summed(ctx));
    TRY(enqueue_quadform(ctx, ctx->x.p, 4)); TRY(comm_reduce(ctx, ctx->scalars.p + 4, 2, NLLS_REDUCE_SUM)); TRY(fetch_scalars(ctx, 4, 2));
    if (xHx_out) *xHx_out = ctx->h_scalars[4]; if (gx_out) *gx_out = ctx->h_scalars[5];
    return NLLS_OK;
    NLLS_API_END(ctx)
}
int nlls_retract(nlls_ctx* ctx, int32_t to, int32_t from) { NLLS_API_BEGIN
    NEED_READY(); if (!valid_set(to) || !valid_set(from) || to == from) return NLLS_ERR_INVALID_ARG;
    spec_note_write(ctx, to);
    return enqueue_retract(ctx, to, from);
    NLLS_API_END(ctx)
}

// ---- multi-GPU: local phases + reduce buffers.  Under nlls_set_shard(rank, nranks > 1):
//   nlls_sweep_cost / nlls_quadform / nlls_max_abs_diag / nlls_grad_* return this rank's PARTIAL values (the caller
//   sums, or takes the max of, them over ranks); the *_local / *_finish pairs bracket the buffer reductions.
int nlls_sweep_gradhess_local(nlls_ctx* ctx) { NLLS_API_BEGIN
    NEED_READY(); ctx->spec_pending = false; ctx->spec_stale = false; TRY(enqueue_sweep_gradhess(ctx));
    ctx->lambda = 0.0; ctx->have_grad = true; ctx->solved = false; ctx->reduced_summed = true;     // (the caller sums the reduce buffer)
    if (ctx->nranks > 1) TRY(enqueue_pack_reduce0(ctx));
    return NLLS_OK;                                  // enqueue only: the reduce buffer is complete in stream order
    NLLS_API_END(ctx)
}
int nlls_sweep_gradhess_finish(nlls_ctx* ctx, double* cost_out) { NLLS_API_BEGIN
    NEED_GRAD();
    if (ctx->nranks > 1) TRY(enqueue_unpack_reduce0(ctx));
    if (!cost_out) return NLLS_OK;                   // cost not wanted (src/optimize.jl:169 discards it): no synchronisation
    TRY(fetch_scalars(ctx, 0, 1)); *cost_out = ctx->h_scalars[0]; return NLLS_OK;
    NLLS_API_END(ctx)
}
int nlls_sweep_cost_local(nlls_ctx* ctx, int32_t which) { NLLS_API_BEGIN NEED_READY(); if (!valid_set(which)) return NLLS_ERR_INVALID_ARG; return enqueue_sweep_cost(ctx, which); NLLS_API_END(ctx) }
int nlls_sweep_cost_finish(nlls_ctx* ctx, double* cost_out) { NLLS_API_BEGIN NEED_READY(); TRY(fetch_scalars(ctx, 0, 1)); if (cost_out) *cost_out = ctx->h_scalars[0]; return NLLS_OK; NLLS_API_END(ctx) }
int nlls_solve_local(nlls_ctx* ctx) { NLLS_API_BEGIN
    NEED_GRAD(); TRY(enqueue_solve_local(ctx));
    return NLLS_OK;                                  // enqueue only, as above
    NLLS_API_END(ctx)
}
int nlls_solve_finish(nlls_ctx* ctx, double* x_out) { NLLS_API_BEGIN
    NEED_GRAD(); TRY(enqueue_solve_finish(ctx));
    int32_t status[4] = {0, 0, 0, 0};
    HIPCHK(hipMemcpyAsync(status, ctx->d_status.p, sizeof(status), hipMemcpyDeviceToHost, ctx->stream));
    if (x_out) HIPCHK(hipMemcpyAsync(x_out, ctx->x.p, sizeof(double) * ctx->info.ndof, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    ctx->solved = true;
    if (status[0] != 0) return status_error(ctx, status[0], "factorisation met a zero pivot");
    return NLLS_OK;
    NLLS_API_END(ctx)
}
int nlls_solve_finish_async(nlls_ctx* ctx) { NLLS_API_BEGIN           // enqueue only: the status comes home with nlls_trial_local's scalars
    NEED_GRAD(); TRY(enqueue_solve_finish(ctx)); ctx->solved = true; ctx->step_cached = false; return NLLS_OK;
    NLLS_API_END(ctx)
}
int nlls_get_reduce_buffer(nlls_ctx* ctx, int32_t stage, void** dev_ptr, int64_t* count) { NLLS_API_BEGIN
    NEED_READY(); if (!dev_ptr || !count) return NLLS_ERR_INVALID_ARG;
    if (stage == 0) { *dev_ptr = ctx->redbuf.p; *count = ctx->redbuf_len; return NLLS_OK; }                                   // after sweep_gradhess_local
    if (stage == 1) { *dev_ptr = ctx->S.p; *count = (int64_t)ctx->s_elems + ctx->nred; return NLLS_OK; }                      // after solve_local: [S | s]
    if (stage == 2) { *dev_ptr = ctx->x.p; *count = ctx->info.ndof; return NLLS_OK; }                                         // after solve_finish: x
    // after nlls_trial_local(out = NULL): the trial's scalars, to be GATHERED (not summed) -- [0] cost, [8] x'Hx, [5] g'x (partial sums),
    // [1] max|x|, [9] |x|^2 over this rank's share of the step, [10] factorisation status
    if (stage == 3) { *dev_ptr = ctx->scalars.p; *count = 11; return NLLS_OK; }
    return fail(ctx, NLLS_ERR_INVALID_ARG, "unknown reduce stage");
    NLLS_API_END(ctx)
}
int nlls_get_step_shard(nlls_ctx* ctx, void** dev_ptr_x, int64_t* reduced_count, int64_t* own_offset, int64_t* own_count) { NLLS_API_BEGIN
    NEED_READY();
    if (dev_ptr_x) *dev_ptr_x = ctx->x.p; if (reduced_count) *reduced_count = ctx->nred;
    if (own_offset) *own_offset = 0; if (own_count) *own_count = ctx->info.ndof;
    return NLLS_OK;
    NLLS_API_END(ctx)
}
int nlls_get_grad_owned(nlls_ctx* ctx, double* b_out) { NLLS_API_BEGIN
    TRY(nlls_get_grad(ctx, b_out));
    if (ctx->nranks > 1) {
        std::vector<double> mask((size_t)ctx->info.ndof);
        HIPCHK(hipMemcpy(mask.data(), ctx->d_dof_mask.p, sizeof(double) * mask.size(), hipMemcpyDeviceToHost));
        for (size_t i = 0; i < mask.size(); ++i) if (mask[i] == 0.0) b_out[i] = 0.0;
    }
    return NLLS_OK;
    NLLS_API_END(ctx)
}
int nlls_get_shard_info(nlls_ctx* ctx, int64_t* out, int32_t n) { NLLS_API_BEGIN
    NEED_READY(); if (!out || n < 1) return NLLS_ERR_INVALID_ARG;
    const int64_t vals[6] = {ctx->rank, ctx->nranks, ctx->local_ncost, ctx->local_nnz_data, ctx->local_ndof, ctx->replicated ? ctx->shard_nranks : 0};
    for (int i = 0; i < n && i < 6; ++i) out[i] = vals[i];
    return NLLS_OK;
    NLLS_API_END(ctx)
}

// ---- timing helpers: HIP events on the context's stream around `reps` back-to-back enqueues -------------
static int time_loop(nlls_ctx* ctx, int reps, float* ms_avg, int (*fn)(nlls_ctx*)) {
    if (reps < 1 || !ms_avg) return NLLS_ERR_INVALID_ARG;
    hipEvent_t e0, e1; HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
    TRY(fn(ctx));   // warm-up
    HIPCHK(hipEventRecord(e0, ctx->stream));
    for (int i = 0; i < reps; ++i) TRY(fn(ctx));
    HIPCHK(hipEventRecord(e1, ctx->stream));
    HIPCHK(hipEventSynchronize(e1));
    float ms = 0; HIPCHK(hipEventElapsedTime(&ms, e0, e1));
    *ms_avg = ms / reps;
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    return NLLS_OK;
}

int nlls_time_sweep_gradhess(nlls_ctx* ctx, int32_t reps, float* ms_avg) { NLLS_API_BEGIN
    NEED_READY(); ctx->spec_pending = false; ctx->spec_stale = false; int rc = time_loop(ctx, reps, ms_avg, [](nlls_ctx* c) { return enqueue_sweep_gradhess(c); });
    ctx->have_grad = true; ctx->reduced_summed = true; return rc;
    NLLS_API_END(ctx)
}
int nlls_time_sweep_accumulate(nlls_ctx* ctx, int32_t reps, float* ms_avg) { NLLS_API_BEGIN
    NEED_READY(); ctx->spec_pending = false; ctx->spec_stale = false; int rc = time_loop(ctx, reps, ms_avg, [](nlls_ctx* c) { return enqueue_sweep_gradhess(c, false); });
    ctx->have_grad = true; ctx->reduced_summed = true; return rc;
    NLLS_API_END(ctx)
}
int nlls_time_sweep_cost(nlls_ctx* ctx, int32_t reps, float* ms_avg) { NLLS_API_BEGIN
    NEED_READY(); return time_loop(ctx, reps, ms_avg, [](nlls_ctx* c) { return enqueue_sweep_cost(c, NLLS_VARS_CURRENT); });
    NLLS_API_END(ctx)
}
int nlls_time_solve(nlls_ctx* ctx, int32_t reps, float* ms_avg) { NLLS_API_BEGIN
    NEED_GRAD(); return time_loop(ctx, reps, ms_avg, [](nlls_ctx* c) { return enqueue_solve(c); });
    NLLS_API_END(ctx)
}
int nlls_get_memory_info(nlls_ctx* ctx, int64_t* out, int32_t n) { NLLS_API_BEGIN
    NEED_READY(); if (!out || n < 4) return NLLS_ERR_INVALID_ARG;
    out[0] = ctx->hot_bytes; out[1] = (int64_t)ctx->arena.n; out[2] = (int64_t)sizeof(double) * ctx->info.nnz_data; out[3] = (int64_t)sizeof(double) * ((int64_t)ctx->s_elems + ctx->nred);
    if (n >= 5) out[4] = (int64_t)sizeof(double) * ((ctx->solve_mode == SOLVE_TSPARSE ? ctx->tsp.nslots_assembled * (int64_t)TSP_TE : (int64_t)ctx->s_elems) + ctx->nred);   // what a sharded trial sums over ranks
    return NLLS_OK;
    NLLS_API_END(ctx)
}
int nlls_check_analytic(nlls_ctx* ctx, double* out, int32_t n) { NLLS_API_BEGIN
    NEED_READY(); if (!out || n < 7) return NLLS_ERR_INVALID_ARG;
    (void)hipSetDevice(ctx->device);
    int64_t nb = 0; TRY(enqueue_check_analytic(ctx, nullptr, &nb));
    for (int q = 0; q < 7; ++q) out[q] = 0.0;
    if (nb == 0) return NLLS_OK;
    nlls::DevBuf<double> d; HIPCHK(d.alloc((size_t)nb * 8));
    TRY(enqueue_check_analytic(ctx, d.p, nullptr));
    std::vector<double> h((size_t)nb * 8);
    HIPCHK(hipMemcpyAsync(h.data(), d.p, sizeof(double) * h.size(), hipMemcpyDeviceToHost, ctx->stream)); HIPCHK(hipStreamSynchronize(ctx->stream));
    for (int64_t b = 0; b < nb; ++b) for (int q = 0; q < 7; ++q) { const double v = h[(size_t)b * 8 + q]; if (!(v <= out[q])) out[q] = v; }   // (a NaN difference comes through)
    return NLLS_OK;
    NLLS_API_END(ctx)
}
int nlls_flush_cache(nlls_ctx* ctx, int64_t bytes) { NLLS_API_BEGIN
    if (!ctx || bytes <= 0 || bytes > ((int64_t)8 << 30)) return NLLS_ERR_INVALID_ARG;
    (void)hipSetDevice(ctx->device);
    const size_t half = ((size_t)bytes / 2 + 255) & ~(size_t)255;
    if (ctx->flushbuf.n < 2 * half) { HIPCHK(ctx->flushbuf.alloc(2 * half)); HIPCHK(hipMemsetAsync(ctx->flushbuf.p, 1, 2 * half, ctx->stream)); }
    HIPCHK(hipMemcpyAsync(ctx->flushbuf.p + half, ctx->flushbuf.p, half, hipMemcpyDeviceToDevice, ctx->stream));
    return NLLS_OK;
    NLLS_API_END(ctx)
}
int nlls_profile_sweep(nlls_ctx* ctx, int32_t on, float* ms_avg, float* ms_min, float* ms_max, int64_t* nsamples) { NLLS_API_BEGIN
    if (!ctx) return NLLS_ERR_INVALID_ARG;
    (void)hipSetDevice(ctx->device);
    if (ms_avg || ms_min || ms_max || nsamples) {            // read what has been recorded so far (synchronises the stream)
        HIPCHK(hipStreamSynchronize(ctx->stream));
        const int64_t cap = (int64_t)ctx->prof_ev.size() / 2, n = std::min(ctx->prof_count, cap);
        double sum = 0; float mn = 1e30f, mx = 0.f;
        for (int64_t i = 0; i < n; ++i) { float ms = 0.f; if (hipEventElapsedTime(&ms, ctx->prof_ev[2 * i], ctx->prof_ev[2 * i + 1]) != hipSuccess) continue; sum += ms; mn = std::min(mn, ms); mx = std::max(mx, ms); }
        // where the fused accumulate kernel stamped itself (first workgroup start .. last workgroup end), that span is the figure:
        // it is what a kernel trace reports for the launch; the event pairs also hold the dispatch latency in front of it
        const int64_t nk = std::min<int64_t>(ctx->prof_kcount, PROF_SLOTS);
        if (nk > 0 && ctx->prof_clk.p) {
            std::vector<unsigned long long> h(2 * (size_t)PROF_MAXWG);
            sum = 0; mn = 1e30f; mx = 0.f; int64_t ok = 0;
            int khz = 100000; (void)hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, ctx->device); if (khz <= 0) khz = 100000;
            const double ms_per_tick = 1.0 / (double)khz;
            for (int64_t i = 0; i < nk; ++i) {
                const unsigned nwg = ctx->prof_nwg[i]; if (!nwg) continue;
                HIPCHK(hipMemcpy(h.data(), ctx->prof_clk.p + (size_t)i * 2 * PROF_MAXWG, sizeof(unsigned long long) * 2 * (size_t)nwg, hipMemcpyDeviceToHost));
                unsigned long long t0 = ~0ull, t1 = 0; for (unsigned w = 0; w < nwg; ++w) { t0 = std::min(t0, h[w]); t1 = std::max(t1, h[nwg + w]); }
                if (t1 <= t0) continue;
                const float ms = (float)((double)(t1 - t0) * ms_per_tick); sum += ms; mn = std::min(mn, ms); mx = std::max(mx, ms); ++ok;
            }
            if (ms_avg) *ms_avg = ok ? (float)(sum / ok) : 0.f; if (ms_min) *ms_min = ok ? mn : 0.f; if (ms_max) *ms_max = mx; if (nsamples) *nsamples = ok;
        } else {
            if (ms_avg) *ms_avg = n ? (float)(sum / n) : 0.f; if (ms_min) *ms_min = n ? mn : 0.f; if (ms_max) *ms_max = mx; if (nsamples) *nsamples = n;
        }
    }
    if (on && !ctx->prof_clk.p) { if (ctx->prof_clk.alloc((size_t)PROF_SLOTS * 2 * PROF_MAXWG) != hipSuccess) return NLLS_ERR_HIP; }
    if (on) ctx->prof_kcount = 0;
    if (on && ctx->prof_ev.empty()) { ctx->prof_ev.resize(128); for (auto& e : ctx->prof_ev) if (hipEventCreate(&e) != hipSuccess) return NLLS_ERR_HIP; }
    ctx->prof_sweep = on != 0; if (on) ctx->prof_count = 0;
    return NLLS_OK;
    NLLS_API_END(ctx)
}
// the event pairs alone: begin .. end of the accumulate dispatch(es) as the command processor stamped them -- what rocprofv3 --kernel-trace reports per dispatch
int nlls_profile_sweep_dispatch(nlls_ctx* ctx, float* ms_avg, float* ms_min, float* ms_max, int64_t* nsamples) { NLLS_API_BEGIN
    if (!ctx) return NLLS_ERR_INVALID_ARG;
    (void)hipSetDevice(ctx->device);
    HIPCHK(hipStreamSynchronize(ctx->stream));
    const int64_t cap = (int64_t)ctx->prof_ev.size() / 2, n = std::min(ctx->prof_count, cap);
    double sum = 0; float mn = 1e30f, mx = 0.f; int64_t ok = 0;
    for (int64_t i = 0; i < n; ++i) { float ms = 0.f; if (hipEventElapsedTime(&ms, ctx->prof_ev[2 * i], ctx->prof_ev[2 * i + 1]) != hipSuccess) { (void)hipGetLastError(); continue; } sum += ms; mn = std::min(mn, ms); mx = std::max(mx, ms); ++ok; }
    if (ms_avg) *ms_avg = ok ? (float)(sum / ok) : 0.f; if (ms_min) *ms_min = ok ? mn : 0.f; if (ms_max) *ms_max = mx; if (nsamples) *nsamples = ok;
    return NLLS_OK;
    NLLS_API_END(ctx)
}
int nlls_time_reduced_solve(nlls_ctx* ctx, int32_t reps, float* ms_avg) { NLLS_API_BEGIN
    NEED_GRAD();
    if (ctx->nred == 0 || ctx->elim_slab) { if (ms_avg) *ms_avg = 0.f; return NLLS_OK; }   // (slab + gather assembly: the tiles are consumed in place)
    if (ctx->solve_mode == SOLVE_BAND && ctx->bcr.ready) {
        // block cyclic reduction copies [S | s] into its own tiles: assemble once, then time the factorisation + backward pass alone
        TRY(enqueue_solve_local(ctx));
        const int rc = time_loop(ctx, reps, ms_avg, [](nlls_ctx* c) { return enqueue_reduced_solve(c); });
        ctx->S_zeroed = false; ctx->solved = false;
        return rc;
    }
    // the dense, one-wave and chain solvers factor S IN PLACE: every repetition assembles it again, and the assembly alone is timed and subtracted
    float ms_both = 0.f, ms_asm = 0.f;
    int rc = time_loop(ctx, reps, &ms_both, [](nlls_ctx* c) { c->S_zeroed = false; int r = enqueue_solve_local(c); return r != NLLS_OK ? r : enqueue_reduced_solve(c); });
    if (rc == NLLS_OK) rc = time_loop(ctx, reps, &ms_asm, [](nlls_ctx* c) { c->S_zeroed = false; return enqueue_solve_local(c); });
    ctx->S_zeroed = false; ctx->solved = false;
    if (ms_avg) *ms_avg = ms_both > ms_asm ? ms_both - ms_asm : 0.f;
    return rc;
    NLLS_API_END(ctx)
}

}  // extern "C"
