// nlls_comm.cpp -- the collectives of the sharded Levenberg-Marquardt loop, behind the C ABI (include/nlls_amd.h, "collectives").
//
// One primitive: an in-place all-reduce (sum / max) of doubles in device memory, ordered on the context's stream.  Either the caller
// installs it (nlls_set_allreduce: any transport; the tests use gloo on host copies so that several ranks can share one GPU), or the
// library brings its own: RCCL (nlls_comm_init_rccl), loaded with dlopen on first use -- an unsharded process never needs librccl, and
// under PyTorch the already-loaded librccl.so.1 is the one that answers.  The trial's scalars are GATHERED with the same primitive: every
// rank writes its row of a [nranks][16] buffer (zeros elsewhere) and the sum over ranks is the gather; one small kernel combines the rows
// (sums, maxima, the factorisation status) and publishes them to the pinned host mirror the single-GPU trial uses.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include "nlls_internal.hpp"

using namespace nlls;

namespace {
int fail(nlls_ctx* c, int code, const std::string& msg) { if (c) c->err = msg; return code; }

struct Rccl {
    void* lib = nullptr;
    decltype(&ncclGetUniqueId) get_unique_id = nullptr; decltype(&ncclCommInitRank) comm_init_rank = nullptr;
    decltype(&ncclAllReduce) all_reduce = nullptr; decltype(&ncclCommDestroy) comm_destroy = nullptr; decltype(&ncclGetErrorString) error_string = nullptr;
    decltype(&ncclCommCount) comm_count = nullptr; decltype(&ncclCommUserRank) comm_user_rank = nullptr; decltype(&ncclCommCuDevice) comm_cu_device = nullptr;
    std::string err;
    bool load() {
        if (lib) return true;
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) { lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL); if (lib) break; }
        if (!lib) { err = std::string("librccl.so.1 cannot be loaded: ") + dlerror(); return false; }
        get_unique_id = reinterpret_cast<decltype(get_unique_id)>(dlsym(lib, "ncclGetUniqueId"));
        comm_init_rank = reinterpret_cast<decltype(comm_init_rank)>(dlsym(lib, "ncclCommInitRank"));
        all_reduce = reinterpret_cast<decltype(all_reduce)>(dlsym(lib, "ncclAllReduce"));
        comm_destroy = reinterpret_cast<decltype(comm_destroy)>(dlsym(lib, "ncclCommDestroy"));
        error_string = reinterpret_cast<decltype(error_string)>(dlsym(lib, "ncclGetErrorString"));
        comm_count = reinterpret_cast<decltype(comm_count)>(dlsym(lib, "ncclCommCount"));
        comm_user_rank = reinterpret_cast<decltype(comm_user_rank)>(dlsym(lib, "ncclCommUserRank"));
        comm_cu_device = reinterpret_cast<decltype(comm_cu_device)>(dlsym(lib, "ncclCommCuDevice"));
        if (!get_unique_id || !comm_init_rank || !all_reduce || !comm_destroy || !error_string) { err = "librccl lacks one of ncclGetUniqueId / ncclCommInitRank / ncclAllReduce / ncclCommDestroy"; lib = nullptr; return false; }
        return true;
    }
};
Rccl& rccl() { static Rccl r; return r; }
static_assert(sizeof(ncclUniqueId) == 128, "nlls_comm_unique_id hands out 128 bytes");

// the library's own all-reduce: RCCL on the context's stream, no synchronisation
int rccl_allreduce(void* user, void* dev_ptr, int64_t count, int32_t op, void* hip_stream) {
    nlls_ctx* c = static_cast<nlls_ctx*>(user);
    const ncclResult_t r = rccl().all_reduce(dev_ptr, dev_ptr, (size_t)count, ncclFloat64, op == NLLS_REDUCE_MAX ? ncclMax : ncclSum,
                                             static_cast<ncclComm_t>(c->rccl_comm), static_cast<hipStream_t>(hip_stream));
    if (r != ncclSuccess) { c->err = std::string("ncclAllReduce: ") + rccl().error_string(r); return 1; }
    return 0;
}

// row `rank` of the gather buffer <- this rank's trial scalars, zeros everywhere else
// (slot 11: the rank's posted termination flag, nlls_comm_post_flag -- non-negative, combined by maximum)
__global__ void gather_pack_kernel(const double* __restrict__ scalars, double* __restrict__ g, int rank, int nranks, double posted) {
    for (int i = threadIdx.x; i < 16 * nranks; i += blockDim.x) g[i] = (i >> 4) == rank ? ((i & 15) < 11 ? scalars[i & 15] : ((i & 15) == 11 ? posted : 0.0)) : 0.0;
}
// rows -> the reduced scalars, in the slots the single-GPU trial fills: [0] cost, [1] max|x|, [2] |x|^2, [5] g'x, [8] x'Ax, [9] x'x, [10] status
// (sharded ranks report max|x| and |x|^2 over their OWN share of the step in [1] and [9]: nlls_trial_local)
__global__ void gather_combine_kernel(const double* __restrict__ g, int nranks, double* __restrict__ out, double* __restrict__ host_out, double seq) {
    if (threadIdx.x != 0) return;
    double cost = 0, mx = 0, gx = 0, xax = 0, xx = 0, st = 0, fl = 0;
    for (int r = 0; r < nranks; ++r) { const double* q = g + 16 * r; fl = q[11] > fl ? q[11] : fl;
        cost += q[0]; gx += q[5]; xax += q[8]; xx += nranks > 1 ? q[9] : q[2];
        mx = (q[1] > mx || q[1] != q[1]) ? q[1] : mx;          // (a NaN step must reach the host: src/optimize.jl:150-151)
        st = q[10] > st ? q[10] : st; }
    out[0] = cost; out[1] = mx; out[2] = xx; out[5] = gx; out[8] = xax; out[9] = xx; out[10] = st; out[11] = fl;
    if (host_out) {
        host_out[11] = fl; host_out[0] = cost; host_out[1] = mx; host_out[2] = xx; host_out[5] = gx; host_out[8] = xax; host_out[9] = xx; host_out[10] = st;
        __threadfence_system();
        reinterpret_cast<volatile double*>(host_out)[32] = seq; reinterpret_cast<volatile double*>(host_out)[33] = seq;
    }
}
}  // namespace

namespace nlls {
int comm_reduce(nlls_ctx* c, double* dev_ptr, int64_t count, int op) {
    if (!c->reduce_fn || count <= 0 || c->replicated) return NLLS_OK;      // (replicas: every rank holds the whole problem, nothing to sum)
    if (c->reduce_fn(c->reduce_user, dev_ptr, count, op, c->stream) != 0) { if (c->err.empty()) c->err = "the installed all-reduce failed"; return NLLS_ERR_HIP; }
    return NLLS_OK;
}
int comm_gather_trial_scalars(nlls_ctx* c, double seq) {
    const int nr = c->nranks;
    if (c->gatherbuf.n < (size_t)16 * nr && c->gatherbuf.alloc((size_t)16 * nr) != hipSuccess) return fail(c, NLLS_ERR_HIP, "gather buffer alloc");
    hipLaunchKernelGGL(gather_pack_kernel, dim3(1), dim3(64), 0, c->stream, c->scalars.p, c->gatherbuf.p, c->rank, nr, c->comm_posted);
    const int rc = comm_reduce(c, c->gatherbuf.p, (int64_t)16 * nr, NLLS_REDUCE_SUM); if (rc != NLLS_OK) return rc;
    hipLaunchKernelGGL(gather_combine_kernel, dim3(1), dim3(64), 0, c->stream, c->gatherbuf.p, nr, c->scalars.p, c->h_scalars_dev, seq);
    return hipGetLastError() == hipSuccess ? NLLS_OK : fail(c, NLLS_ERR_HIP, "gather launch");
}
void comm_release(nlls_ctx* c) {
    if (c->rccl_comm && rccl().lib) { (void)rccl().comm_destroy(static_cast<ncclComm_t>(c->rccl_comm)); }
    c->rccl_comm = nullptr; c->reduce_fn = nullptr; c->reduce_user = nullptr;
}
}  // namespace nlls

extern "C" {
int nlls_set_allreduce(nlls_ctx* ctx, nlls_allreduce_fn fn, void* user) { NLLS_API_BEGIN
    if (!ctx) return NLLS_ERR_INVALID_ARG;
    comm_release(ctx);
    ctx->reduce_fn = fn; ctx->reduce_user = fn ? user : nullptr;
    return NLLS_OK;
    NLLS_API_END(ctx)
}
int nlls_comm_unique_id(void* id128) { NLLS_API_BEGIN
    if (!id128) return NLLS_ERR_INVALID_ARG;
    if (!rccl().load()) return NLLS_ERR_UNSUPPORTED;
    ncclUniqueId id; if (rccl().get_unique_id(&id) != ncclSuccess) return NLLS_ERR_HIP;
    memcpy(id128, &id, 128);
    return NLLS_OK;
    NLLS_API_END(nullptr)
}
int nlls_comm_init_rccl(nlls_ctx* ctx, const void* id128) { NLLS_API_BEGIN
    if (!ctx || !id128) return NLLS_ERR_INVALID_ARG;
    if (!rccl().load()) return fail(ctx, NLLS_ERR_UNSUPPORTED, rccl().err);
    (void)hipSetDevice(ctx->device);
    comm_release(ctx);
    ncclUniqueId id; memcpy(&id, id128, 128);
    ncclComm_t comm = nullptr;
    const ncclResult_t r = rccl().comm_init_rank(&comm, ctx->shard_nranks, id, ctx->shard_rank);
    if (r != ncclSuccess) return fail(ctx, NLLS_ERR_HIP, std::string("ncclCommInitRank: ") + rccl().error_string(r));
    ctx->rccl_comm = comm; ctx->reduce_fn = rccl_allreduce; ctx->reduce_user = ctx;
    return NLLS_OK;
    NLLS_API_END(ctx)
}
int nlls_comm_post_flag(nlls_ctx* ctx, double value) { NLLS_API_BEGIN
    if (!ctx || !(value >= 0.0)) return NLLS_ERR_INVALID_ARG;
    ctx->comm_posted = value; ctx->comm_gathered = false;      // (a new iteration: the answer must come from a trial that follows this post)
    return NLLS_OK;
    NLLS_API_END(ctx)
}
int nlls_comm_agreed_flag(nlls_ctx* ctx, double local_value, double* out) { NLLS_API_BEGIN
    if (!ctx || !out) return NLLS_ERR_INVALID_ARG;
    // the agreed value only when the last trial really gathered it: with an all-reduce installed on a DENSE system (or before the first trial) nothing
    // writes comm_agreed, and answering 0 would silently disable the caller's deadline (advisor, round 4)
    // (replicas -- dense systems, systems without an eliminated set under nlls_set_shard -- enter no collective: nothing waits on a peer, each replica keeps its own deadline)
    *out = (ctx->reduce_fn && ctx->comm_gathered) ? ctx->comm_agreed : local_value;
    return NLLS_OK;
    NLLS_API_END(ctx)
}
int nlls_comm_info(nlls_ctx* ctx, int64_t* out, int32_t n) { NLLS_API_BEGIN
    if (!ctx || !out || n < 4) return NLLS_ERR_INVALID_ARG;
    out[0] = 1; out[1] = 0; out[2] = ctx->device; out[3] = 0;
    if (ctx->rccl_comm && rccl().lib && rccl().comm_count && rccl().comm_user_rank) {
        int cnt = 0, rk = 0, dev = ctx->device; ncclComm_t comm = static_cast<ncclComm_t>(ctx->rccl_comm);
        ncclResult_t r = rccl().comm_count(comm, &cnt); if (r == ncclSuccess) r = rccl().comm_user_rank(comm, &rk);
        if (r == ncclSuccess && rccl().comm_cu_device) r = rccl().comm_cu_device(comm, &dev);
        if (r != ncclSuccess) return fail(ctx, NLLS_ERR_HIP, std::string("ncclCommCount / ncclCommUserRank: ") + rccl().error_string(r));
        out[0] = cnt; out[1] = rk; out[2] = dev; out[3] = 1;
    } else if (ctx->reduce_fn) { out[0] = ctx->nranks; out[1] = ctx->rank; out[3] = 2; }
    return NLLS_OK;
    NLLS_API_END(ctx)
}
}
