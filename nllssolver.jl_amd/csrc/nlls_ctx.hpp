// nlls_ctx.hpp -- context, device buffers and the work lists built at nlls_upload_structure time.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/nlls_amd.h"
#include "nlls_kinds.hpp"
#include "nlls_devbuf.hpp"
#include "nlls_bcr.hpp"
#include "nlls_tsp.hpp"

namespace nlls {

// ---- accumulate work lists ----------------------------------------------------------------------
// One tile = the share of one workgroup: a run of block rows whose A.data / b segments are staged
// as one LDS image (light), or one (part of a) heavy row.
struct Tile {
    uint32_t e0, e1;        // entry range in the list
    uint32_t row0, nrows;   // rows of the list covered (index into RowInfo)
    int64_t  data_off;      // start of the image's A.data segment
    uint32_t data_len;      // doubles of A.data in the image
    uint32_t b_off, b_len;  // b segment
    uint32_t flags;
};
constexpr uint32_t TILE_PARTIAL = 1;   // rows shared with other tiles: flush the image with atomics
constexpr uint32_t TILE_DIRECT  = 2;   // segment too large for LDS: off-diagonal blocks go straight to HBM atomics
constexpr uint32_t TILE_NOZERO  = 4;   // exclusive rows + unique off-diagonal destinations: every image entry is written, no zero fill

struct RowInfo {
    uint32_t diag_off;      // image-relative offset of the diagonal block
    uint32_t b_off;         // image-relative offset of the row's part of b (>= data_len)
};

// dest word of the entry's own slot
constexpr uint32_t DEST_NONE       = 0xFFFFFFFFu;
constexpr uint32_t OWN_ROW_MASK    = 0xFFFFu;
constexpr uint32_t OWN_COST_OWNER  = 1u << 16;   // this entry adds the block's cost to the total
constexpr uint32_t OWN_KERNEL_FREE = 1u << 17;   // adaptive residual whose kernel variable is optimised
constexpr uint32_t OWN_COPY_SHIFT  = 18;         // which of the ACC_COPIES diagonal accumulators of its row the entry adds to
constexpr uint32_t ACC_COPIES      = 4;          // (spreads the LDS atomics of a row's entries over distinct addresses)

// ---- the folded accumulate sweep (round 5): every block evaluated ONCE -------------------------------------------
// A group whose entry lists split into {one light-only list that holds every cost block, heavy-only lists} (bundle adjustment: point rows
// light, camera rows heavy, the adaptive kernel's one long row heavy) used to evaluate each block once per ROLE -- three times at BASELINE
// config 5.  Folded, the light pass alone evaluates the block and also forms the block's contributions to its heavy rows (diagonal block, b,
// and the row's off-diagonal blocks that all its entries share); a tile sums them per heavy row it touches in LDS (a "slot" = one heavy
// row inside one tile) and leaves one record per slot in a slab; `gh_fold_gather_kernel` then sums every heavy row's records IN A FIXED
// ORDER and writes the row -- no atomics in HBM, A.data bit-reproducible as before, each byte of it still written exactly once.
constexpr int FOLD_MAX_HEAVY = 3;          // heavy slots of one group
constexpr uint32_t FOLD_SLOT_NONE = 0x3F;  // (6 bits of slot + 4 bits of accumulator copy per heavy slot in EntryList::fslot)
constexpr int FOLD_MAX_CW = 128;           // doubles of one record
struct FoldTile {            // per light tile
    uint32_t slab_off;       // first double of the tile's records in Group::fslab
    uint8_t  ns[FOLD_MAX_HEAVY];   // slots per heavy slot h (records of h start behind those of h - 1)
    uint8_t  pad;
};
struct FoldHeavy {           // per heavy slot h of a folded group (kernel argument)
    int32_t slot;            // the residual's slot T
    int32_t ds, nsym, cw;    // dof of T, lower triangle of its diagonal block, doubles per record: [nsym | ds | shared off-diagonal blocks (T, t) in slot order]
    int32_t xmask;           // bit t: block (T, t) is stored in T's row and shared by all entries of the row -> part of the record
    int32_t copies;          // LDS accumulator copies per slot (power of two)
    int32_t maxns;           // most slots of h in any tile
    int32_t xdof[4];         // dof of the residual's slot t (size of block (T, t): ds x xdof[t], column-major)
};
struct FoldRow {             // one heavy row: where its record components go, and which records are its own
    int64_t  data_off;       // A.data offset of the row's segment
    uint32_t diag_off;       // ... of its diagonal block inside the segment
    uint32_t b_off;          // offset in b
    uint32_t xoff[4];        // segment-relative offset of block (T, t), DEST_NONE if not stored
    uint32_t cbeg, cend;     // its records: Group::fcons[cbeg .. cend) = offsets into the slab, in tile order
    uint32_t h, pad;
};

struct EntryList {          // all (cost, slot) incidences of one cost group and one slot, sorted by block row
    int slot = 0;
    int64_t n = 0;
    DevBuf<double>   data;  // [n][ndata]
    DevBuf<uint32_t> voff;  // [n][ndeps] storage offsets of the block's variables
    DevBuf<uint32_t> dest;  // [n][ndeps] see above
    DevBuf<RowInfo>  rows;
    DevBuf<Tile>     light, heavy;
    int64_t nlight = 0, nheavy = 0;
    bool unique_dest = false;
    uint32_t light_lds = 0, heavy_lds = 0;   // max image doubles over the tiles
    // heavy-only lists whose entries write nothing but their own row (no off-diagonal block of theirs is stored in it) and share their flag word:
    // the heavy pass then streams 4 bytes per other slot instead of the 8 + 8 of voff / dest (BA camera rows: 20 instead of 32 bytes per entry)
    bool compact = false; uint32_t own_flags = 0;
    DevBuf<uint32_t> hvoff;  // [n][ndeps - 1]: storage offsets of the OTHER slots' variables, slot order
    DevBuf<uint32_t> fslot;  // folded sweep, light list only: per entry, 10 bits per heavy slot h -- slot of the entry's heavy row inside its tile (FOLD_SLOT_NONE: that variable is fixed) | accumulator copy << 6
    DevBuf<FoldTile> ftiles; // ... per light tile
    DevBuf<uint32_t> frowx;  // ... per light row [4]: tile-relative A.data offset of the row's block (slot, t) that all its entries share (DEST_NONE: none)
};

struct DenseList {          // dense linear system: one entry per cost
    int64_t n = 0;
    DevBuf<double>   data;  // [n][ndata]
    DevBuf<uint32_t> voff;  // [n][ndeps]
    DevBuf<uint32_t> brow;  // [n][ndeps] dof offset of the slot's block in b, DEST_NONE if fixed
    DevBuf<uint32_t> aoff;  // dynamic-size groups of a BLOCK-SPARSE system: [n] offset of the variable's diagonal block in A.data (empty: dense system)
};

struct Group {
    int res_kind = 0, ndeps = 0, ndata = 0, nres = 0, adaptive = 0;
    RobustSpec rk{};
    int64_t ncost = 0;
    // cost-order arrays (cost sweep)
    DevBuf<double>   data;      // [ncost][ndata]
    DevBuf<uint32_t> voff;      // [ncost][ndeps]
    DevBuf<uint32_t> fixedcost; // costs without any free variable (only their cost counts in the sweep)
    int64_t nfixedcost = 0;
    EntryList lists[MAX_SLOTS];
    // folded sweep (see FoldTile): the light list `fold_ls` carries every block; lists[fold_ls].fslot / .ftiles hold its per-entry / per-tile words
    bool fold = false; int fold_ls = -1, fold_nh = 0; FoldHeavy fh[FOLD_MAX_HEAVY] = {}; uint32_t fold_lds = 0, fold_unique = 0, fold_shared = 0;   // fold_lds: doubles of LDS per workgroup; fold_unique: bit t -- the light rows' blocks (ls, t) have one writer each; fold_shared: bit t -- they are shared by all entries of their row
    DevBuf<FoldRow> frows; DevBuf<uint32_t> fcons; DevBuf<double> fslab; int64_t nfrows = 0;
    DenseList dense;
    std::vector<int32_t> local_of;   // sharded upload (the library partitions): cost block k of the caller's group -> its index in this rank's arrays, -1: another rank's.  Empty: every block is local
    // the cost sweep's view of the blocks: a light entry list that holds EVERY cost of the group exactly once (all its slot's variables are
    // free) serves it instead of the cost-order arrays -- the same 24 bytes per block the next gradient sweep streams, so that inside the LM
    // loop (cost sweep of the accepted trial, then the gradient sweep) they are read from the memory-side cache, and the cost-order arrays
    // (another 24 bytes per block) stay out of the loop's working set.  -1: none (cost order).  The sum is taken in list order: a fixed order.
    int cost_list = -1;
    // matrix-free LM trial (round 6, nlls_mf.hip): the group's blocks in ELIMINATION order -- supernode by supernode (launch order), member by member, column block by column
    // block of [E] -- so that lane l of a wavefront's batch finds its block at obs0 + first member * ncb + l.  Same record form as an entry list (data + one storage offset per slot):
    // the cost sweep of an LM trial reads these arrays too (the loop's working set holds the blocks once)
    DevBuf<double> mf_data; DevBuf<uint32_t> mf_voff;
};

// ---- Schur / solve structures ---------------------------------------------------------------------
// one fast-path supernode, in launch order (position in d_fast_groups): everything a kernel needs to start on it comes with ONE
// uniform 32-byte load instead of a chain of dependent ones (group list -> group -> neighbour pointer -> neighbour records)
// the finishing reduction of the small dense system's LM trial (nlls_wave.hpp dense_fin_body): cost partials -> out[0], the trial's scalars -> the pinned host mirror
struct DenseFin { const double* cpart; int64_t ncp; double* out; double* host_out; double seq; };
constexpr int TINY_DENSE_MAX_WGS = 256;     // workgroups (= images of [A | b]) per cost group in the small dense system's sweep
constexpr int64_t TRIAL_COST_POFS = 4096;   // offset of the cost partials of an LM trial in nlls_ctx::partials (the post-solve partials end at 3584)
struct ElimDesc {
    uint32_t v0, nmem;       // first member (index into the elimination arrays), members
    uint32_t nd, rc_off;     // columns of [E] (neighbour dof), offset of their reduced columns in d_elim_rc
    int64_t dg0;             // A.data offset of the first member's diagonal block
    uint32_t eb0, obs0;      // b offset of the first member; first record of the supernode in Group::mf_data / mf_voff (matrix-free trial)
};
struct MfDesc {              // a supernode of the matrix-free trial (nlls_mf.hip), launch order: those of several batches first
    uint32_t v0, nmem, nd, rc_off;   // as ElimDesc
    uint32_t eb0, obs0;              // b offset of the first member; first record in Group::mf_data / mf_voff
    uint32_t B, slab;                // members per batch; offset of the supernode's share in nlls_ctx::slab (GatherCon offsets point into it)
    //                                  B: members per batch (one lane per cost block: at most 64 / blocks per member, chosen so that the four wavefronts get equal shares)
};
struct SchurNbr {            // one off-diagonal block touching an eliminated block
    int64_t off;             // offset in A.data
    uint32_t rcol;           // dof offset of the neighbour in the reduced system
    uint16_t dim;            // neighbour block size
    uint16_t trans;          // 0: stored as (elim x nbr) [dv x du]; 1: stored as (nbr x elim) [du x dv]
};
constexpr int SOLVE_SMALL = 0, SOLVE_DENSE = 1, SOLVE_BAND = 2, SOLVE_TSPARSE = 3;   // (3: tile-sparse LDL' in a nested-dissection order, nlls_tsp.hip)
constexpr unsigned PROF_SLOTS = 16, PROF_MAXWG = 16384;
constexpr int NLLS_SUB_NONE = 0, NLLS_SUB_SCHUR_SHAPE = 1;     // SCHUR_SHAPE: the Schur kernels cannot stage this structure -- the full system may still be solvable
struct SchurCopy {           // a reduced-reduced block copied from A.data into S
    int64_t off; uint32_t r, c; uint16_t rows, cols;
};
// Deterministic assembly of the reduced system (no atomics): every supernode leaves its share of S -- one contiguous column-major
// block per pair of its neighbour blocks, then its share of the rhs -- in a slab of its own; one wavefront per block pair of S then
// sums the shares in a fixed order and writes the block cyclic reduction's tiles directly (schur_gather_kernel).
struct GatherJob {
    int64_t copy_off;        // A.data offset of the reduced-reduced block under this pair (-1: none)
    uint32_t r0, c0;         // reduced dof of the block's first row / column (r0 >= c0); rhs jobs: r0 only
    uint16_t rows, cols;
    uint16_t kind;           // 0: block pair, 1: rhs segment of a reduced block, 2: identity on the padding behind the band
    uint16_t copy_trans;     // the stored block is the transpose (the border reordering flipped it)
    uint32_t cbeg, cend;     // contributions
    uint32_t boff;           // rhs jobs: offset in b
    uint32_t pad_;
};
// one share of a block pair: ld > 0 -- a block in a supernode's slab at offset off, leading dimension ld (aux = 1: stored as the transpose of the pair's block in S);
// ld == 0 -- a member of a small supernode, formed on the fly: E_A' (C_v + lambda I)^-1 E_B with E_A at A.data + off, E_B at A.data + aux
// (rhs segments: b_v at b + aux), the inverse at Cinv + cinv
struct GatherCon { uint32_t off, ld, aux, cinv; };

}  // namespace nlls

struct nlls_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    hipStream_t stream2 = nullptr;           // side stream: heavy-row tiles run beside the light-row tiles
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    // in-situ profile of the accumulate launches (nlls_profile_sweep): event pairs recorded around them inside the caller's own loop
    bool prof_sweep = false; std::vector<hipEvent_t> prof_ev; int64_t prof_count = 0;
    hipEvent_t prof_e0 = nullptr, prof_e1 = nullptr; bool prof_taken = false;   // the event pair of the sweep being enqueued, handed to its fused / folded launch (hipExtLaunchKernelGGL)
    nlls::DevBuf<unsigned long long> prof_clk; int64_t prof_kcount = 0; unsigned prof_nwg[16] = {0};   // [PROF_SLOTS][2][PROF_MAXWG] start / end stamp of every workgroup (100 MHz constant clock) of the fused accumulate launch
    std::string err;
    int err_sub = 0;                         // why the last nlls_upload_structure declined (NLLS_SUB_*): control flow never reads the error text
    int rank = 0, nranks = 1; bool presharded = false;
    // what nlls_set_shard asked for; rank / nranks above are what the uploaded problem RUNS with: a problem that does not shard (a dense system, no eliminated set to
    // partition by) runs as replicas -- every rank the whole problem, rank 0 of 1, no collective (replicated)
    int shard_rank = 0, shard_nranks = 1; bool replicated = false;
    // collectives behind the ABI (nlls_comm.cpp): the installed all-reduce, and the library's own RCCL communicator when it is that
    nlls_allreduce_fn reduce_fn = nullptr; void* reduce_user = nullptr; void* rccl_comm = nullptr;
    nlls::DevBuf<double> gatherbuf;          // [nranks][16]: the ranks' trial scalars, gathered by a sum over rows that are zero elsewhere
    double comm_posted = 0.0, comm_agreed = 0.0;   // nlls_comm_post_flag / nlls_comm_agreed_flag: slot 11 of the gather rows, combined by maximum
    bool comm_gathered = false;                    // the last LM trial went through the collective route and gathered the flags: comm_agreed is this iteration's (a dense system, or no trial yet: the local value decides)

    // ---- structure ------------------------------------------------------------------------------
    bool ready = false;
    nlls_info info{};
    std::vector<int32_t> var_kind, var_dim;
    std::vector<uint32_t> var_off;           // storage offsets (nvar+1)
    std::vector<uint64_t> blockindices;      // as uploaded (1-based, 0 fixed)
    std::vector<int32_t> blocksizes;         // per block
    std::vector<int64_t> boffsets;           // 0-based dof offset per block (nblocks+1)
    std::vector<int64_t> it_colptr, it_rowval, it_nzval;   // BSM indicestransposed, 0-based
    std::vector<int64_t> diag_off;           // per block: offset of the diagonal block (sparse) / dense index
    std::vector<nlls::Group> groups;

    // ---- device state -----------------------------------------------------------------------------
    nlls::DevBuf<double> vars[3];            // problem.variables / varnext / varbest (src/problem.jl:9-12)
    int vars_slot[3] = {0, 1, 2};            // logical -> physical (swaps are pointer swaps)
    nlls::DevBuf<double> A, b, x;
    nlls::DevBuf<int32_t> d_var_kind, d_var_dim;
    nlls::DevBuf<uint32_t> d_var_off, d_var_boff;   // d_var_boff: dof offset of the variable's block, DEST_NONE if fixed
    nlls::DevBuf<int64_t> d_diag_off;        // per block
    nlls::DevBuf<int32_t> d_blocksizes;
    nlls::DevBuf<int64_t> d_zero_off;        // segments that must be zeroed before a sweep (shared / split rows)
    nlls::DevBuf<uint32_t> d_zero_len;
    int64_t nzero = 0;
    nlls::DevBuf<uint32_t> d_zero_b_off, d_zero_b_len;
    nlls::DevBuf<double> partials;           // per-workgroup cost partials
    int num_cus = 256;
    nlls::DevBuf<double> scalars;            // small device scratch for scalar results
    double* h_scalars = nullptr;             // pinned host mirror
    int64_t trial_seq = 0;                   // sequence number the trial's finishing launch publishes in h_scalars[32], [33]
    double* h_scalars_dev = nullptr;         // ... as the device sees it (the trial's finishing launch writes the scalars there itself)
    int64_t npartials = 0;
    double lambda = 0.0;                     // accumulated uniformscaling! (src/iterators.jl:149,162)
    // Look-ahead sweep (round 5): nlls_lm_trial enqueues the gradient sweep of the TRIAL point behind the trial's own launches, before the host has seen the trial's cost --
    // the accepted path (swap CURRENT <-> NEXT, nlls_sweep_gradhess(ctx, NULL)) then finds A and b already being computed and the GPU does not idle through the host's
    // turn-around; a rejected trial (the next nlls_lm_trial without a swap) sweeps the current point again first.  Transparent: same kernels, same data, same results.
    int grad_phys = -1;                      // physical variable slot A and b are the linearisation of
    bool spec_on = true, spec_pending = false, spec_stale = false, spec_armed = true;   // spec_on: NLLS_NO_LOOKAHEAD_SWEEP unset; pending: A, b belong to grad_phys, not (yet) to CURRENT; stale: that slot was written since; armed: the last look-ahead was used (a miss disarms until the next real sweep)
    int64_t spec_hits = 0, spec_misses = 0;  // nlls_get_solve_stats
    int64_t sweeps_since_set = 0;            // gradient sweeps the caller has asked for since nlls_set_variables(CURRENT)
    bool tail_zero_for_lookahead = false, heavy_rows_zeroed = false;   // the look-ahead sweep's zero fill in the trial's finishing launch (trial_finish_kernel)
    // the small dense system (fewer than 64 unknowns, nothing eliminated, one rank: curve fits, Rosenbrock): the sweep leaves one image of [A | b] per workgroup in
    // dense_slab and ONE gathering launch sums them (no zero fill, no atomics on HBM, no mirror launch); an LM trial is one single-workgroup launch for
    // damping + factorisation + step statistics + retraction, then the cost sweep.  NLLS_TINY_DENSE=0 keeps the general kernels (A/B)
    bool tiny_dense = false, tiny_dense_on = true; nlls::DevBuf<double> dense_slab; int64_t dense_slab_wgs = 0, dense_slab_used = 0;
    // An LM trial followed by its look-ahead sweep is FOUR launches: [damped solve + statistics + retraction] (one wavefront), the cost sweep, [the finishing reduction as
    // workgroup 0 + the look-ahead accumulate sweep], the gather.  dense_fin: a finishing reduction waiting for the accumulate launch that carries it (NLLS_TINY_FIN_ROLE=0:
    // always a launch of its own, A/B)
    nlls::DenseFin dense_fin{}; bool dense_fin_pending = false, tiny_fin_role = true;
    bool have_grad = false;
    // Device-timed NLLSResult buckets (round 6; src/structs.jl:37-50, filled at src/iterators.jl:152,157): the launches of an LM trial leave the constant clock (100 MHz) in the pinned
    // mirror -- h_scalars[40] start of the assembly launch, [41] of the back-substitution, [42] start of the cost launch (matrix-free trial: of the finishing workgroup; the cost rides in
    // the back-substitution), [43] end of the finishing workgroup -- one thread each, off every critical path.  nlls_lm_trial turns them into nanoseconds: solver [40]..[42], cost
    // [42]..[43], gradient = end of the previous trial .. [40] (the sweep between two trials and the host's turn-around).  nlls_get_time_buckets; nlls_lm_iterations reports them.
    int64_t tb_grad_ns = 0, tb_cost_ns = 0, tb_solver_ns = 0, tb_trials = 0; double tb_prev_end = 0.0, tb_ns_per_tick = 10.0;
    // Phase timing of the COLLECTIVE trial (NLLS_OPT_PHASE_EVENTS; bench.py --gpus N): events recorded on the stream at the phase boundaries of nlls_lm_trial -- local assembly,
    // the [S | s] all-reduce, reduced solve, back-substitution, trial tail -- and around the gradient sweep; off by default (an event is a marker packet on the queue)
    bool phase_on = false; std::vector<hipEvent_t> phase_ev; double phase_ms[6] = {0, 0, 0, 0, 0, 0}; int64_t phase_trials = 0, phase_sweeps = 0;
    double* stamp_ptr() const { return h_scalars_dev ? h_scalars_dev + 40 : nullptr; }
    // Matrix-free LM trial (round 6; nlls_mf.hip).  Two-slot Schur problems whose eliminated blocks all sit on the fast path: nlls_lm_trial evaluates the cost blocks of every
    // supernode inside the elimination launch and again inside the back-substitution launch -- the point rows of A.data (151 of its 151.5 MB at BASELINE config 4) are never
    // written or read by the loop.  What stays materialised is the reduced rows (camera diagonal blocks, their part of b: what `grad_level` 1 means) -- the gradient sweep between
    // two iterations shrinks to the reduced slot's pass; b's eliminated part is written by the elimination launch itself.  A.data in the reference's layout is formed on demand:
    // every entry point that reads it (nlls_get_bsm_data, nlls_solve, nlls_max_abs_diag, ...) sweeps in full first (ensure_grad level 2).  NLLS_FLAG_MATERIALIZE / NLLS_MATERIALIZE=1 /
    // nlls_set_option(NLLS_OPT_MATERIALIZE): the round-5 path.
    bool mf_ok = false, mf_on = true; int mf_group = -1, mf_ps = -1;     // eligibility (build_mf), run-time switch, the cost group and its eliminated slot
    int grad_level = 0;                      // what A and b hold of the linearisation at grad_phys: 0 nothing, 1 the reduced rows, 2 everything
    bool mf_step = false;                    // the last solve was matrix-free: its back-substitution launch has left the trial's cost and the step statistics as rows of partials in mf_q (mf_rows of them)
    nlls::DevBuf<double> mf_q; int mf_rows = 0; bool mf_fin_defer = false, mf_fin_pending = false;   // (the finishing workgroup may ride in the look-ahead sweep's launch)
    nlls::DevBuf<nlls::MfDesc> d_mf_desc; int64_t mf_nbig = 0; size_t mf_lds = 0; uint32_t mf_ecap = 0, mf_wsz = 0; bool mf_use = false;    // per-supernode partials of the step's quadratic form; dynamic LDS of the two launches
    int64_t mf_trials = 0, mf_reduced_sweeps = 0, full_sweeps = 0;   // diagnostics (nlls_get_solve_stats [23..25])
    std::vector<int64_t> h_erow; std::vector<int64_t> h_eptr; std::vector<int64_t> h_enbr_block; std::vector<nlls::ElimDesc> h_elim_desc; std::vector<uint32_t> h_fast_voff;   // host copies kept between build_schur and build_mf

    // ---- sharding ------------------------------------------------------------------------------------
    bool replicate_xr = false;               // the step's reduced part is written on every rank (sharded LM trial without the stage-2 reduction)
    int ps_np = 0, ps_np2 = 0;                 // partial counts of the last enqueue_post_solve (for the trial's finishing launch)
    int dense_t128_min = 16;                   // ... only while the trailing matrix has at least this many 128-blocks per side (fewer: the 64 x 64 kernel fills the chip better)
    bool dense_pad128 = false;                 // the dense layout is padded to a multiple of 128 rows (windowed and look-ahead factorisations: 128-column panels only)
    bool dense_window = false;                 // dense LDL' restricted to the band of the (re-ordered) reduced system + the border strip: O(n w^2) instead of n^3 / 3 (build_schur decides)
    bool dense_t128 = true;                    // dense LDL': 128 x 128 tiles in the two-panel trailing update (NLLS_DENSE_T64=1: the 64 x 64 kernel, for A/B runs)
    bool dense_fused_bwd = true;               // dense LDL': the backward substitution in one launch (NLLS_DENSE_STEP_BACKWARD=1: one launch per 64-column block, for A/B runs)
    // LM trial with the retraction inside the back-substitution launch and the step statistics / quadratic form inside the cost sweep's launch
    // (one rank, every eliminated block on the fast path, Euclidean eliminated variables): no launch of its own for them.  NLLS_POST_SPLIT=1: off (A/B)
    nlls::DevBuf<uint32_t> d_fast_voff;      // where the variable of each eliminated member is stored (elimination order)
    nlls::DevBuf<uint32_t> d_rest_var; nlls::DevBuf<int32_t> d_rest_red;   // the other variables, and where their step starts in the reduced solution (-1: fixed)
    bool fast_all_euclid = false, post_fuse = true, retract_done = false; int trial_to = -1, trial_from = -1;
    bool elim_split = false;                   // NLLS_ELIM_SPLIT=1: the assembly of the reduced system in three launches (A/B)
    bool elim_mfma = true;                     // narrow supernodes (nd + 1 <= 64) are eliminated on the matrix cores (NLLS_ELIM_TILED=1: the register-tiled kernel, for A/B runs)
    bool elim_selected = false;
    std::vector<int32_t> owner_of_block;
    int64_t local_ncost = 0, local_nnz_data = 0, local_ndof = 0;
    int64_t nred_ranges = 0, redbuf_len = 0;  // stage-0 reduce buffer: [cost | reduced rows of A.data | reduced part of b]
    nlls::DevBuf<int64_t> d_red_off; nlls::DevBuf<uint32_t> d_red_len, d_red_dst, d_red_which;
    nlls::DevBuf<double> redbuf;
    nlls::DevBuf<double> d_dof_mask;         // 1 for dof this rank accounts for in global reductions (quadratic forms, max diag)
    nlls::DevBuf<uint8_t> d_blk_mask, d_row_mask;
    size_t s_elems = 0;                      // S occupies S.p[0, s_elems); the rhs vector s follows it (one reduce buffer)
    // ---- solve ---------------------------------------------------------------------------------------
    std::vector<uint8_t> is_elim;            // per block
    int64_t nelim = 0, nred = 0;             // blocks eliminated / dof of the reduced system
    int64_t nelim_all = 0;                   // ... eliminated over ALL ranks of a pre-sharded upload (= nelim otherwise): nothing rank-local may gate a collective or the solver choice
    nlls::DevBuf<int64_t> d_elim_ptr;        // CSR over eliminated blocks -> SchurNbr
    nlls::DevBuf<nlls::SchurNbr> d_elim_nbr;
    nlls::DevBuf<int64_t> d_elim_diag;       // A.data offset of C_v
    nlls::DevBuf<uint32_t> d_elim_boff;      // dof offset in b/x
    nlls::DevBuf<uint16_t> d_elim_dim;
    nlls::DevBuf<uint32_t> d_elim_group;     // supernodes: runs of eliminated blocks with identical neighbour sets
    int64_t nelim_groups = 0, n_fast_groups = 0, n_slow_groups = 0;
    nlls::DevBuf<uint32_t> d_fast_groups, d_slow_groups, d_slow_blocks;   // d_slow_blocks: members of the slow supernodes
    nlls::DevBuf<nlls::ElimDesc> d_elim_desc; nlls::DevBuf<uint32_t> d_elim_rc;   // per fast supernode (launch order): descriptor, reduced column of every E column
    nlls::DevBuf<uint32_t> d_fast_members;   // members of the fast supernodes
    nlls::DevBuf<uint8_t> d_blk_slowmask;    // d_blk entries NOT in rows of fast members (and owned by this rank)
    nlls::DevBuf<nlls::SchurCopy> d_blk_slow; int64_t nblk_slow = 0;   // the same as a compact list
    // collective route with the reduced rows NOT yet summed over ranks (lazy stage 0, nlls_sweep_gradhess(ctx, NULL)): every rank accounts for ITS share
    // of the reduced rows -- the compact list with the reduced-reduced blocks on every rank, the all-blocks mask, and the dof mask of g'x
    nlls::DevBuf<nlls::SchurCopy> d_blk_slow_lazy; int64_t nblk_slow_lazy = 0;
    nlls::DevBuf<uint8_t> d_blk_mask_lazy; nlls::DevBuf<double> d_dof_mask_lazy;
    bool reduced_summed = true;              // false between a lazy sweep and the first entry point that needs the summed rows (ensure_reduced_summed)
    int64_t n_stage0 = 0, n_lazy_trials = 0; // (diagnostics: nlls_get_solve_stats [11], [12])
    bool lazy_stage0 = true;                 // NLLS_EAGER_STAGE0=1: sum the reduced rows behind every sweep, as round 2 did (A/B)
    nlls::DevBuf<double> tE;                 // E_v s of the last solve per fast member (s = reduced solution): reused by the quadratic form
    bool tE_valid = false; int64_t n_fast_members = 0;
    bool status_known_zero = false;          // the host has read the last solve's status and it was 0: the next solve need not reset it on the device
    bool S_zeroed = false;                   // the last solve's back-substitution left S zero-filled for the next one (saves the memset launches)
    bool step_cached = false; double c_maxabs = 0, c_sumsq = 0, c_gx = 0, c_xAx = 0, c_xx = 0;   // host copies of the last solve's step statistics
    nlls::DevBuf<double> Cinv;               // (C_v + lambda I)^-1 of the fast-path members, fast_dv^2 doubles per eliminated block
    int fast_dv = 0, fast_maxk = 0, fast_maxk_narrow = 0;
    int64_t n_fast_narrow = 0;               // fast supernodes with nd + 1 <= 64 come first in d_fast_groups
    int64_t n_fast_n60 = 0;                  // ... and among them those with nd <= 60 (two tile waves suffice) first of all
    int max_elim_dim = 0, max_nbr_dof = 0;
    // generic (LDS-staged) elimination, per supernode class: d_slow_groups = [supernodes whose pair accumulators fit in LDS | those that add
    // every member's products straight into S]; each class is one launch with the LDS its own widest supernode needs (up to gfx950's 160 KB)
    int64_t n_slow_acc = 0; int slow_nd_acc = 0, slow_nd_noacc = 0; size_t elim_lds_acc = 0, elim_lds_noacc = 0;
    int64_t n_band = 0; int nbd = 0, bw = 0;   // reduced ordering: [banded part | border dof | rhs]
    int red_reordered = 0; int64_t bw_caller = -1;   // the banded part is in reverse Cuthill-McKee order (narrower than the caller's block order, whose half bandwidth was bw_caller)
    double damped_floor = 1e-11;             // pivot floor of DAMPED solves (block cyclic reduction and tile-sparse LDL'): relative to the unknown's original diagonal entry; 0 with NLLS_FLAG_NO_PIVOT_FLOOR
    int solve_mode = 0, band_CH = 0, band_H = 0, band_SEG = 0, band_NSEG = 0;
    bool band_blocked = true;                // blocked (MFMA) band factorisation when the bandwidth allows
    bool band_twisted = true;               // factor the band from both ends at once (two workgroups) when the layout allows
    bool gather_ready = false, tiles_zeroed = false; std::vector<uint32_t> h_slab_off;   // the gather index (d_gjobs / d_gcons / slab) exists; the tiles the gather writes into are zero
    bool elim_slab = false;                 // slab + gather assembly straight into the block cyclic reduction's tiles (single rank, fast-path supernodes only)
    nlls::DevBuf<double> slab; nlls::DevBuf<uint32_t> d_slab_off, d_slab_groups; int64_t n_slab60 = 0, n_slabnar = 0, n_slabwide = 0; nlls::DevBuf<nlls::GatherJob> d_gjobs; nlls::DevBuf<nlls::GatherCon> d_gcons; int64_t n_gjobs = 0;
    nlls::TspSolver tsp;                    // tile-sparse LDL' of a reduced system that is neither a narrow band nor small (nlls_tsp.hip)
    nlls::BcrSolver bcr;                    // block cyclic reduction of the bordered band (nlls_bcr.hip): the default band solver when it supports the shape
    nlls::DevBuf<nlls::SchurCopy> d_copy;    // reduced-reduced blocks
    int64_t ncopy = 0;
    nlls::DevBuf<nlls::SchurCopy> d_blk;     // every stored block with full-system dof offsets (quadratic forms)
    int64_t nblk = 0;
    nlls::DevBuf<uint32_t> d_red_boff;       // reduced dof -> dof offset in b/x
    nlls::DevBuf<double> S, Lwork;           // reduced system (+ rhs vector in its tail), factor workspace
    double* s_ptr() const { return S.p + s_elems; }
    nlls::DevBuf<double> Yelim;              // C^-1 * [E | b] per eliminated block (reused by back-substitution)
    nlls::DevBuf<int32_t> d_status;          // factorisation status
    // The buffers an LM iteration touches, moved into ONE contiguous allocation at the end of an upload (compact_hot_set, nlls_structure.cpp):
    // 250 MB at BASELINE config 4 against a 256 MB memory-side cache.  Scattered over separate allocations they alias in that cache by
    // the luck of their physical placement -- the accumulate launch inside the LM loop took 41 .. 49 us from process to process; side by
    // side in one window (and the arrays the loop never reads kept out of it) it takes 41 in every process.
    nlls::DevBuf<char> arena, arena_pre;
    int64_t hot_bytes = 0;                   // bytes of the hot set (what an LM iteration reads or writes): nlls_get_memory_info
    nlls::DevBuf<char> flushbuf;             // nlls_flush_cache: foreign traffic for cold-cache timings
    bool solved = false;
};
