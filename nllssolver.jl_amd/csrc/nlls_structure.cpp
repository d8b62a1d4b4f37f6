// nlls_structure.cpp -- host-side symbolic phase of nlls_upload_structure.
//
// Replaces makesymmvls (src/linearsystem.jl:91-124): block sizes, the sparse/dense decision
// (src/utils.jl:108-120), the BlockSparseMatrix pattern + offsets in the reference's exact layout
// (src/BlockSparseMatrix.jl:30-47), boffsets (src/linearsystem.jl:36-41) -- and builds the device
// work lists the reference has no counterpart for: per-(cost type, slot) entry lists sorted by
// block row, LDS-image tiles, and the Schur elimination lists.
#include <algorithm>
#include <numeric>
#include <map>
#include <unordered_map>

#include "nlls_ctx.hpp"
#include "nlls_internal.hpp"

namespace nlls {

bool res_desc(int kind, ResDesc& d) {
    switch (kind) {
#define X(K) case K: d.ndeps = Res<K>::NDEPS; d.nres = is_cost_kind<K> ? 0 : Res<K>::M; d.ndata = Res<K>::NDATA; d.adaptive = Res<K>::ADAPT; \
        static_assert(Res<K>::NDEPS <= MAX_SLOTS, "at most MAX_SLOTS variables per cost block"); \
        for (int i = 0; i < MAX_SLOTS; ++i) { d.sk[i] = i < Res<K>::NDEPS ? Res<K>::SK[i] : 0; d.sd[i] = i < Res<K>::NDEPS ? Res<K>::SD[i] : 0; } return true;
        NLLS_FOR_EACH_RES(X)
#undef X
    // dynamic-size kinds (src/autodiff.jl:96-121): counts that depend on the variable's run-time length n are -1 here and fixed per group
    // at upload (build_structure)
    case NLLS_RES_DYN_LINEAR: d = ResDesc{1, 1, -1, 0, {NLLS_VAR_DYNAMIC, 0, 0, 0}, {0, 0, 0, 0}}; return true;
    case NLLS_RES_DYN_NORM:   d = ResDesc{1, -1, 0, 0, {NLLS_VAR_DYNAMIC, 0, 0, 0}, {0, 0, 0, 0}}; return true;
    case NLLS_RES_DYN_LINEARSQ: d = ResDesc{1, -1, -2, 0, {NLLS_VAR_DYNAMIC, 0, 0, 0}, {0, 0, 0, 0}}; return true;   // ndata = n + n*n
    case NLLS_COST_DYN_LINEAR:  d = ResDesc{1, 0, -3, 0, {NLLS_VAR_DYNAMIC, 0, 0, 0}, {0, 0, 0, 0}}; return true;   // an AbstractCost (nres 0); ndata = n
    }
    return false;
}

static int fail(nlls_ctx* c, int code, const std::string& msg) { c->err = msg; return code; }

#define HIPCHK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return fail(c, NLLS_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); } while (0)

// tiling parameters (see DESIGN.md "accumulate kernel")
constexpr uint32_t LIGHT_MAX_ENTRIES = 256;    // one entry per lane of a 256-thread workgroup (192: 47.1 instead of 43.5 us in situ, 128: 60 -- tools/sweep_ab.py)
constexpr uint32_t LIGHT_IMG_MAX     = 6144;   // doubles of LDS image (48 KiB) -> 3 workgroups per CU
constexpr uint32_t HEAVY_ROW_ENTRIES = 128;    // rows with more entries get a workgroup of their own
constexpr uint32_t HEAVY_MAX_ENTRIES_DEFAULT = 1024;   // entries per heavy tile (two wavefronts x 8 pipeline stages); longer rows are split (PARTIAL)

// ---- the hot set ------------------------------------------------------------------------------------------------------------
// every device buffer the LM loop reads or writes, in the order they are laid out in the arena.  NOT in it: the cost-order arrays of a
// group whose cost sweep reads a light entry list instead (Group::cost_list), voff / dest of compact heavy lists (one word per tile is
// read), the per-block neighbour records of the generic elimination, the all-blocks list of the quadratic form (the loop uses the short
// one), the chain / dense solvers' workspace when block cyclic reduction solves the band.
// (round 6: the matrix-free trial's own buffers -- Group::mf_data / mf_voff, the slabs, its descriptors and partials, 60 MB at BASELINE config 4 -- stay OUTSIDE: the arena is the
// working set of the MATERIALISING loop, 250 of the cache's 256 MB; with them inside it was 275 MB and the accumulate launch of that loop went from 45 to 51 us.  The matrix-free loop's
// own working set is 100 MB: wherever its buffers lie, it fits.)
namespace {
struct HotItem { void** pp; size_t bytes; bool* owned; };
template <class T> void hot(std::vector<HotItem>& v, DevBuf<T>& b) { if (b.p && b.n) v.push_back(HotItem{reinterpret_cast<void**>(&b.p), b.n * sizeof(T), &b.owned}); }
void hot_set(nlls_ctx* c, std::vector<HotItem>& v) {
    hot(v, c->A); hot(v, c->b); hot(v, c->x); for (auto& q : c->vars) hot(v, q);
    for (Group& G : c->groups) {
        for (EntryList& E : G.lists) { if (E.n == 0) continue;
            if (G.fold && E.slot != G.fold_ls) continue;          // folded sweep: the heavy lists are never read
            hot(v, E.data); hot(v, E.rows); hot(v, E.light); hot(v, E.heavy); hot(v, E.hvoff);
            if (!E.compact) { hot(v, E.voff); hot(v, E.dest); }
            hot(v, E.fslot); hot(v, E.ftiles); hot(v, E.frowx); }
        if (G.fold) { hot(v, G.frows); hot(v, G.fcons); hot(v, G.fslab); }
        if (G.cost_list < 0 || !c->info.is_sparse) { hot(v, G.data); hot(v, G.voff); }
        hot(v, G.fixedcost); hot(v, G.dense.data); hot(v, G.dense.voff); hot(v, G.dense.brow);
    }
    hot(v, c->d_var_kind); hot(v, c->d_var_dim); hot(v, c->d_var_off); hot(v, c->d_var_boff); hot(v, c->d_diag_off); hot(v, c->d_blocksizes);
    hot(v, c->d_zero_off); hot(v, c->d_zero_len); hot(v, c->d_zero_b_off); hot(v, c->d_zero_b_len); hot(v, c->partials); hot(v, c->scalars);
    hot(v, c->S); hot(v, c->Cinv); hot(v, c->tE); hot(v, c->d_status); hot(v, c->d_copy); hot(v, c->d_red_boff); hot(v, c->d_blk_slow);
    hot(v, c->d_elim_desc); hot(v, c->d_elim_rc); hot(v, c->d_elim_diag); hot(v, c->d_elim_boff); hot(v, c->d_elim_dim); hot(v, c->d_fast_members); hot(v, c->d_fast_groups);
    if (c->bcr.ready) { hot(v, c->bcr.ws); hot(v, c->bcr.d_upd); hot(v, c->bcr.d_elim); } else hot(v, c->Lwork);
}
}  // namespace
// move the hot set into one allocation (called at the end of a successful upload; NLLS_NO_ARENA=1: leave every buffer where hipMalloc put it)
static int compact_hot_set(nlls_ctx* c) {
    if (getenv("NLLS_NO_ARENA")) { std::vector<HotItem> v; hot_set(c, v); size_t t = 0; for (const HotItem& it : v) t += it.bytes; c->hot_bytes = (int64_t)t; return NLLS_OK; }
    std::vector<HotItem> v; hot_set(c, v);
    size_t total = 0; for (const HotItem& it : v) if (*it.owned) total += (it.bytes + 255) & ~(size_t)255;
    c->hot_bytes = (int64_t)total;
    if (total == 0) return NLLS_OK;
    // the arena was reserved BEFORE the first buffer of this upload (build_structure): physical placement is best while the device memory
    // is untouched -- allocated here, behind hundreds of megabytes of allocations, the same window came out at 40 .. 46 us from process
    // to process, reserved first at 40 .. 41.7 (tools/sweep_ab.py).  An estimate that fell short falls back to allocating now.
    DevBuf<char> fresh;
    if (c->arena_pre.p && c->arena_pre.n >= total) fresh = std::move(c->arena_pre);
    else { c->arena_pre.release();
        if (fresh.alloc(total) != hipSuccess) { (void)hipGetLastError(); return NLLS_OK; } }   // no room for a second copy of the working set: the buffers stay where hipMalloc put them (placement is an optimisation, not a requirement)
    size_t off = 0;
    for (HotItem& it : v) { if (!*it.owned) continue;
        HIPCHK(hipMemcpyAsync(fresh.p + off, *it.pp, it.bytes, hipMemcpyDeviceToDevice, c->stream));
        it.bytes = (it.bytes + 255) & ~(size_t)255; off += it.bytes; }
    HIPCHK(hipStreamSynchronize(c->stream));
    off = 0;
    for (HotItem& it : v) { if (!*it.owned) continue; (void)hipFree(*it.pp); *it.pp = fresh.p + off; *it.owned = false; off += it.bytes; }
    c->arena = std::move(fresh);                             // (the previous upload's arena, if any, is freed here: nothing points into it any more)
    if (c->bcr.ready) c->bcr.geom.ws = c->bcr.ws.p;          // the one cached device pointer
    return NLLS_OK;
}

struct HostList { std::vector<int64_t> cost; std::vector<int64_t> rowptr; std::vector<int64_t> rows; };   // the incidences (cost, slot) of one group and slot, sorted by block row
struct SlotHost { std::vector<uint32_t> dest; std::vector<Tile> light, heavy; };                          // what build_structure uploaded for that list, kept for build_fold

// The folded accumulate sweep of one group (nlls_ctx.hpp, FoldTile): decides whether the group qualifies and builds the per-entry slot words, the per-tile
// record layout and every heavy row's list of records.  Leaves G.fold false (and everything else untouched) when it does not qualify.
//   NLLS_SWEEP_FOLD=0: never; =1: every group that qualifies; default: groups of three or more slots (BASELINE config 5: three evaluations per block become one;
//   for two-slot bundle adjustment the heavy rows already hide behind the light tiles of the fused launch -- A/B in DESIGN.md 4.1)
static int build_fold(nlls_ctx* c, Group& G, const ResDesc& d, const nlls_cost_group& in, const uint64_t* bi, std::vector<HostList>& hls, std::vector<SlotHost>& sh,
                      const std::vector<int64_t>& segs, const std::vector<int32_t>& row_nlists, std::vector<uint8_t>& row_zero, int32_t flags) {
    G.fold = false; G.fold_ls = -1; G.fold_nh = 0; G.nfrows = 0;
    const char* env = getenv("NLLS_SWEEP_FOLD"); const int mode = env ? atoi(env) : -1;
    if (mode == 0 || (mode < 0 && d.ndeps < 3) || d.ndeps < 2 || d.ndeps > FOLD_MAX_HEAVY + 1 || (flags & NLLS_FLAG_FORCE_ATOMIC)) return NLLS_OK;
    const int nd = d.ndeps; const int64_t nb = (int64_t)c->blocksizes.size();
    int ls = -1; for (int s = 0; s < nd; ++s) { const EntryList& E = G.lists[s]; if (E.n == 0) continue;
        if (E.nlight > 0 && E.nheavy == 0) { if (ls >= 0) return NLLS_OK; ls = s; } else if (!(E.nheavy > 0 && E.nlight == 0)) return NLLS_OK; }
    if (ls < 0 || G.lists[ls].n != G.ncost) return NLLS_OK;            // the light list must hold EVERY cost block of the group
    int nheavy = 0; for (int s = 0; s < nd; ++s) if (s != ls && G.lists[s].n > 0) ++nheavy;
    if (!nheavy) return NLLS_OK;
    auto dof = [&](int s) { return var_dof(d.sk[s], d.sd[s]); };
    FoldHeavy fh[FOLD_MAX_HEAVY] = {};
    std::vector<std::vector<int32_t>> rowidx(nd - 1);                 // block row -> index of the heavy row in its list
    std::vector<FoldRow> frows; std::vector<uint32_t> row_base(nd - 1, 0);
    for (int T = 0; T < nd; ++T) { if (T == ls) continue; const int h = T - (T > ls); FoldHeavy& F = fh[h];
        F.slot = T; F.ds = dof(T); F.nsym = F.ds * (F.ds + 1) / 2; F.cw = 0; F.xmask = 0; F.copies = 1; F.maxns = 0;
        for (int t = 0; t < 4; ++t) F.xdof[t] = t < nd ? dof(t) : 0;
        const EntryList& E = G.lists[T]; if (E.n == 0) continue;
        const HostList& L = hls[T]; const SlotHost& H = sh[T];
        for (const Tile& t : H.heavy) if (t.flags & TILE_DIRECT) return NLLS_OK;      // a row too long for an LDS image: its blocks have one writer each, nothing to fold
        rowidx[h].assign(nb, -1); row_base[h] = (uint32_t)frows.size();
        for (size_t r = 0; r < L.rows.size(); ++r) { const int64_t br = L.rows[r];
            if (row_nlists[br] != 1) return NLLS_OK;                                    // a row other lists add to as well
            FoldRow fr{}; fr.data_off = segs[br]; fr.diag_off = (uint32_t)(c->diag_off[br] - segs[br]); fr.b_off = (uint32_t)c->boffsets[br]; fr.h = (uint32_t)h;
            int xm = 0;
            for (int t = 0; t < 4; ++t) fr.xoff[t] = DEST_NONE;
            for (int t = 0; t < nd; ++t) { if (t == T) continue;
                const uint32_t first = H.dest[(size_t)L.rowptr[r] * nd + t];
                for (int64_t e = L.rowptr[r]; e < L.rowptr[r + 1]; ++e) if (H.dest[(size_t)e * nd + t] != first) return NLLS_OK;   // blocks with one writer each in a heavy row (points listed before cameras): not folded
                fr.xoff[t] = first; if (first != DEST_NONE) xm |= 1 << t; }
            if (r == 0) F.xmask = xm; else if (F.xmask != xm) return NLLS_OK;
            rowidx[h][br] = (int32_t)r; frows.push_back(fr); }
        F.cw = F.nsym + F.ds; for (int t = 0; t < nd; ++t) if (F.xmask >> t & 1) F.cw += F.ds * dof(t);
        if (F.cw > FOLD_MAX_CW) return NLLS_OK;
    }
    // slots: the distinct heavy rows every light tile touches, per heavy slot; an entry's rank among its tile's entries of the same heavy row picks its accumulator copy
    const HostList& LL = hls[ls]; const SlotHost& HL = sh[ls]; const int64_t n = G.lists[ls].n;
    std::vector<uint8_t> eslot((size_t)n * (nd - 1), (uint8_t)FOLD_SLOT_NONE), erank((size_t)n * (nd - 1), 0);
    std::vector<FoldTile> ftiles(HL.light.size());
    std::vector<std::vector<uint32_t>> cons(frows.size());
    uint64_t slab = 0; uint32_t max_img = 0;
    for (size_t ti = 0; ti < HL.light.size(); ++ti) { const Tile& t = HL.light[ti]; FoldTile& ft = ftiles[ti]; ft = FoldTile{};
        if (slab > 0xFFFF0000ull) return NLLS_OK;
        ft.slab_off = (uint32_t)slab;
        for (int h = 0; h < nd - 1; ++h) { const FoldHeavy& F = fh[h]; if (G.lists[F.slot].n == 0) continue;
            std::vector<int32_t> rows_here; std::vector<uint32_t> count;
            for (uint32_t e = t.e0; e < t.e1; ++e) { const int64_t k = LL.cost[e]; const uint64_t bv = bi[in.varind[k * nd + F.slot] - 1]; if (!bv) continue;
                const int32_t r = rowidx[h][bv - 1]; if (r < 0) return NLLS_OK;
                size_t j = 0; while (j < rows_here.size() && rows_here[j] != r) ++j;
                if (j == rows_here.size()) { if (rows_here.size() >= FOLD_SLOT_NONE - 1) return NLLS_OK; rows_here.push_back(r); count.push_back(0); }
                eslot[(size_t)e * (nd - 1) + h] = (uint8_t)j; erank[(size_t)e * (nd - 1) + h] = (uint8_t)(count[j]++ & 0xFF); }
            ft.ns[h] = (uint8_t)rows_here.size(); fh[h].maxns = std::max<int32_t>(fh[h].maxns, (int32_t)rows_here.size());
            for (size_t j = 0; j < rows_here.size(); ++j) cons[row_base[h] + rows_here[j]].push_back((uint32_t)(slab + (uint64_t)j * F.cw));
            slab += (uint64_t)rows_here.size() * F.cw; }
        }
    // accumulator copies: about a thousand doubles of LDS per heavy slot (conflicts of a wavefront's lanes on one address serialise: a slot every lane adds to gets 16)
    for (int h = 0; h < nd - 1; ++h) { FoldHeavy& F = fh[h]; if (!F.cw || !F.maxns) { F.copies = 1; continue; }
        int cp = 1; while (cp < 16 && (int64_t)2 * cp * F.maxns * F.cw <= 1024) cp *= 2; F.copies = cp; }   // (measured at config 5: 2048 doubles per heavy slot 73 us against 68 -- the zero fill and the sum of more copies cost more than the conflicts they spare)
    // the light rows' own off-diagonal blocks (ls, t): ONE writer each (registers -> HBM) or shared by all entries of the row (summed in the row's accumulator); anything in between: not folded
    uint32_t uniq = 0, shared = 0; std::vector<uint32_t> rowx(LL.rows.size() * 4, DEST_NONE);
    for (const Tile& t : HL.light) if (t.flags & TILE_PARTIAL) return NLLS_OK;      // rows other lists add to as well
    for (int t = 0; t < nd; ++t) { if (t == ls) continue; bool any = false, u = true, sh = true; std::vector<uint32_t> tmp;
        for (size_t rr = 0; rr < LL.rows.size(); ++rr) { tmp.clear();
            for (int64_t e = LL.rowptr[rr]; e < LL.rowptr[rr + 1]; ++e) if (HL.dest[(size_t)e * nd + t] != DEST_NONE) tmp.push_back(HL.dest[(size_t)e * nd + t]);
            if (tmp.empty()) continue;
            any = true;
            if (tmp.size() != (size_t)(LL.rowptr[rr + 1] - LL.rowptr[rr])) sh = false;
            for (size_t i = 1; i < tmp.size(); ++i) if (tmp[i] != tmp[0]) sh = false;
            rowx[rr * 4 + t] = tmp[0];
            std::sort(tmp.begin(), tmp.end()); for (size_t i = 1; i < tmp.size(); ++i) if (tmp[i] == tmp[i - 1]) { u = false; break; } }
        if (!any) continue;
        if (u) uniq |= 1u << t; else if (sh) shared |= 1u << t; else return NLLS_OK; }
    const uint32_t dsl = (uint32_t)dof(ls); uint32_t nw = dsl * (dsl + 1) / 2 + dsl; for (int t = 0; t < nd; ++t) if (shared >> t & 1) nw += dsl * (uint32_t)dof(t);
    for (size_t ti = 0; ti < HL.light.size(); ++ti) { const Tile& t = HL.light[ti];
        uint32_t need = t.nrows * 2u /* FOLD_ROW_COPIES */ * nw;
        for (int h = 0; h < nd - 1; ++h) need += (uint32_t)ftiles[ti].ns[h] * (uint32_t)fh[h].copies * (uint32_t)fh[h].cw;
        max_img = std::max(max_img, need); }
    if ((size_t)(max_img + 2) * sizeof(double) > 64 * 1024) return NLLS_OK;
    std::vector<uint32_t> fslot((size_t)n, 0);
    for (int64_t e = 0; e < n; ++e) { uint32_t w = 0;
        for (int h = 0; h < FOLD_MAX_HEAVY; ++h) { uint32_t sl = FOLD_SLOT_NONE, cp = 0;
            if (h < nd - 1) { sl = eslot[(size_t)e * (nd - 1) + h]; cp = erank[(size_t)e * (nd - 1) + h] & (uint32_t)(fh[h].copies - 1); }
            w |= (sl | cp << 6) << (10 * h); }
        fslot[(size_t)e] = w; }
    std::vector<uint32_t> fcons; fcons.reserve((size_t)(slab / 8 + 16));
    for (size_t r = 0; r < frows.size(); ++r) { frows[r].cbeg = (uint32_t)fcons.size(); fcons.insert(fcons.end(), cons[r].begin(), cons[r].end()); frows[r].cend = (uint32_t)fcons.size(); }
    EntryList& EL = G.lists[ls];
    HIPCHK(EL.fslot.upload(fslot)); HIPCHK(EL.ftiles.upload(ftiles)); HIPCHK(EL.frowx.upload(rowx));
    HIPCHK(G.frows.upload(frows)); HIPCHK(G.fcons.upload(fcons)); HIPCHK(G.fslab.alloc(std::max<uint64_t>(slab, 1)));
    G.nfrows = (int64_t)frows.size(); G.fold_lds = max_img; G.fold_unique = uniq; G.fold_shared = shared; G.fold_ls = ls; G.fold_nh = nd - 1;
    for (int h = 0; h < FOLD_MAX_HEAVY; ++h) G.fh[h] = fh[h];
    // the heavy rows are written whole by the gather launch: no zero fill in front of the sweep on their account (a sharded upload zeroes every reduced row anyway: build_structure)
    for (int T = 0; T < nd; ++T) if (T != ls && G.lists[T].n > 0) for (int64_t br : hls[T].rows) row_zero[br] = 0;
    G.fold = true;
    return NLLS_OK;
}

int build_structure(nlls_ctx* c, int64_t nvar, const int32_t* var_kind, const int32_t* var_dim, const uint64_t* bi,
                    int32_t ngroups, const nlls_cost_group* groups, int32_t flags) {
    c->rank = c->shard_rank; c->nranks = c->shard_nranks; c->replicated = false;      // what nlls_set_shard asked for (a problem that does not shard falls back to replicas below)
    c->presharded = (flags & NLLS_FLAG_PRESHARDED) != 0 && c->nranks > 1;
    c->ready = false; c->solved = false; c->have_grad = false; c->lambda = 0; c->reduced_summed = true; c->n_stage0 = 0; c->n_lazy_trials = 0;
    c->spec_pending = false; c->spec_stale = false; c->spec_armed = true; c->grad_phys = -1; c->grad_level = 0; c->sweeps_since_set = 0; c->dense_fin_pending = false; c->heavy_rows_zeroed = false; c->tail_zero_for_lookahead = false;   // (a re-upload starts from a clean look-ahead state)
    { std::vector<HotItem> v; hot_set(c, v); for (HotItem& it : v) if (!*it.owned) { *it.pp = nullptr; *it.owned = true; } }   // what lived in the previous upload's arena is gone with it
    c->arena.release(); c->arena_pre.release();     // ... so release it NOW: a re-upload would otherwise hold two arenas (and every buffer once more) at its peak
    c->groups.clear();
    // ---- variables ------------------------------------------------------------------------------
    c->var_kind.assign(var_kind, var_kind + nvar); c->var_dim.assign(var_dim, var_dim + nvar);
    c->var_off.assign(nvar + 1, 0);
    uint64_t off = 0;
    for (int64_t i = 0; i < nvar; ++i) {
        int st = var_storage(var_kind[i], var_dim[i]);
        if (st <= 0 || var_dof(var_kind[i], var_dim[i]) > (var_kind[i] == NLLS_VAR_DYNAMIC ? NLLS_MAX_DYN_DIM : NLLS_MAX_BLOCK_SZ)) return fail(c, NLLS_ERR_UNSUPPORTED, "unregistered variable kind / block larger than MAX_BLOCK_SZ");
        c->var_off[i] = (uint32_t)off; off += st;
        if (off > 0xFFFFFFF0ull) return fail(c, NLLS_ERR_UNSUPPORTED, "variable storage exceeds 32-bit offsets");
    }
    c->var_off[nvar] = (uint32_t)off;
    c->blockindices.assign(bi, bi + nvar);
    int64_t nb = 0; for (int64_t i = 0; i < nvar; ++i) nb = std::max<int64_t>(nb, (int64_t)bi[i]);
    c->blocksizes.assign(nb, 0);
    for (int64_t i = 0; i < nvar; ++i) if (bi[i]) c->blocksizes[bi[i] - 1] = var_dof(var_kind[i], var_dim[i]);   // linearsystem.jl:96-102
    c->boffsets.assign(nb + 1, 0);
    for (int64_t k = 0; k < nb; ++k) { if (c->blocksizes[k] <= 0) return fail(c, NLLS_ERR_INVALID_ARG, "blockindices are not a dense 1..nblocks numbering"); c->boffsets[k + 1] = c->boffsets[k] + c->blocksizes[k]; }
    const int64_t ndof = c->boffsets[nb];
    // ---- groups: validate ------------------------------------------------------------------------
    std::vector<ResDesc> desc(ngroups);
    int64_t ncost_total = 0; bool any_dyn = false;
    for (int g = 0; g < ngroups; ++g) {
        if (!res_desc(groups[g].res_kind, desc[g])) return fail(c, NLLS_ERR_UNSUPPORTED, "unregistered residual kind");
        int base = groups[g].robust_kind & 0xF;
        if (base > NLLS_ROBUST_GEMAN_MCCLURE || (groups[g].robust_kind & ~0x1F)) return fail(c, NLLS_ERR_UNSUPPORTED, "unregistered robust kernel");
        if (is_dyn_kind(groups[g].res_kind)) {       // n = the run-time length of the block's variable: the same for every block of the group
            any_dyn = true;
            if ((base != NLLS_ROBUST_NONE || (groups[g].robust_kind & NLLS_ROBUST_SCALED)) && groups[g].res_kind == NLLS_COST_DYN_LINEAR) return fail(c, NLLS_ERR_UNSUPPORTED, "a non-squared cost takes no robust kernel");
            int n = -1;
            for (int64_t k = 0; k < groups[g].ncost; ++k) { const int64_t v = groups[g].varind[k];
                if (v < 1 || v > nvar) return fail(c, NLLS_ERR_INVALID_ARG, "varind out of range");
                if (n < 0) n = var_dim[v - 1]; else if (var_dim[v - 1] != n) return fail(c, NLLS_ERR_UNSUPPORTED, "dynamic-size blocks of one group must share the variable length"); }
            if (n < 0) n = 0;
            if (groups[g].res_kind == NLLS_RES_DYN_LINEARSQ && n > 512) return fail(c, NLLS_ERR_UNSUPPORTED, "NLLS_RES_DYN_LINEARSQ: variable longer than 512");
            if (desc[g].ndata == -1) desc[g].ndata = 1 + n; else if (desc[g].ndata == -2) desc[g].ndata = n + n * n; else if (desc[g].ndata == -3) desc[g].ndata = n;
            if (desc[g].nres < 0) desc[g].nres = n;
        }
        const ResDesc& d = desc[g];
        for (int64_t k = 0; k < groups[g].ncost; ++k) for (int s = 0; s < d.ndeps; ++s) {
            int64_t v = groups[g].varind[k * d.ndeps + s];
            if (v < 1 || v > nvar) return fail(c, NLLS_ERR_INVALID_ARG, "varind out of range");
            if (var_kind[v - 1] != d.sk[s] || (d.sk[s] == NLLS_VAR_EUCLIDEAN && var_dim[v - 1] != d.sd[s]))
                return fail(c, NLLS_ERR_UNSUPPORTED, "variable kind does not match the residual's slot");
            for (int t = 0; t < s; ++t) if (groups[g].varind[k * d.ndeps + t] == v) return fail(c, NLLS_ERR_UNSUPPORTED, "a cost block lists the same variable twice");
        }
        ncost_total += groups[g].ncost;
    }
    // ---- sparsity: triu(V*V' .> 0) over the unfixed rows (linearsystem.jl:108-110) -------------------
    bool sparse = false;
    std::vector<uint64_t> keys;
    if ((ndof >= 40 || (flags & NLLS_FLAG_FORCE_SPARSE)) && nb > 0) {
        size_t cap = 0; for (int g = 0; g < ngroups; ++g) cap += (size_t)groups[g].ncost * (desc[g].ndeps * (desc[g].ndeps + 1) / 2);
        keys.reserve(cap);
        for (int g = 0; g < ngroups; ++g) { const int nd = desc[g].ndeps;
            for (int64_t k = 0; k < groups[g].ncost; ++k) for (int s = 0; s < nd; ++s) {
                uint64_t a = bi[groups[g].varind[k * nd + s] - 1]; if (!a) continue;
                for (int t = 0; t <= s; ++t) { uint64_t b = bi[groups[g].varind[k * nd + t] - 1]; if (!b) continue;
                    uint64_t hi = std::max(a, b), lo = std::min(a, b); keys.push_back((hi - 1) * (uint64_t)nb + (lo - 1)); } } }
        // (a rank's share of a pre-sharded problem: a reduced variable none of ITS cost blocks touches still owns its diagonal block -- the sum over
        //  ranks fills it, and the reduced system must have the same blocks on every rank)
        if ((flags & NLLS_FLAG_PRESHARDED) && c->nranks > 1) for (int64_t k = 0; k < nb; ++k) keys.push_back((uint64_t)k * (uint64_t)nb + (uint64_t)k);
        std::sort(keys.begin(), keys.end()); keys.erase(std::unique(keys.begin(), keys.end()), keys.end());
        int64_t bnnz = 0; for (uint64_t k : keys) bnnz += (int64_t)c->blocksizes[k / nb] * c->blocksizes[k % nb];   // utils.jl:110-120
        sparse = (flags & NLLS_FLAG_FORCE_SPARSE) || (bnnz * 64) < (25 * ndof * (ndof - 40));                        // utils.jl:108
    }
    // (dynamic-size blocks in a BLOCK-SPARSE system -- src/autodiff.jl:96-121 with src/linearsystem.jl:105-121: their variable appears in no other kind of block,
    //  so its block row holds its diagonal block only; the blocks accumulate into it directly, below)
    // (under sharding a dynamic-size block -- no eliminated variable in it -- is owned like any other such cost block: round robin; its variable's row is a reduced
    //  row, summed over ranks with the others)
    c->it_colptr.clear(); c->it_rowval.clear(); c->it_nzval.clear(); c->diag_off.assign(nb, -1);
    int64_t nnz_data = 0;
    if (sparse) {   // BlockSparseMatrix constructor, BlockSparseMatrix.jl:30-47 (0-based here)
        c->it_colptr.assign(nb + 1, 0); c->it_rowval.resize(keys.size()); c->it_nzval.resize(keys.size());
        size_t q = 0; int64_t start = 0;
        for (int64_t row = 0; row < nb; ++row) {
            c->it_colptr[row] = (int64_t)q;
            while (q < keys.size() && (int64_t)(keys[q] / nb) == row) {
                int64_t col = keys[q] % nb; c->it_rowval[q] = col; c->it_nzval[q] = start;
                if (col == row) c->diag_off[row] = start;
                start += (int64_t)c->blocksizes[row] * c->blocksizes[col]; ++q; }
        }
        c->it_colptr[nb] = (int64_t)q; nnz_data = start;
        if (nnz_data > 0xFFFFFFF0ll) return fail(c, NLLS_ERR_UNSUPPORTED, "A.data exceeds 32-bit offsets");
    } else {
        nnz_data = ndof * ndof;
        for (int64_t k = 0; k < nb; ++k) c->diag_off[k] = c->boffsets[k] + ndof * c->boffsets[k];
    }
    std::vector<uint64_t>().swap(keys);
    auto block_off = [&](int64_t row, int64_t col) -> int64_t {   // offset of block(A, row, col), row >= col
        const int64_t* b0 = c->it_rowval.data() + c->it_colptr[row]; const int64_t* b1 = c->it_rowval.data() + c->it_colptr[row + 1];
        const int64_t* it = std::lower_bound(b0, b1, col);
        return (it != b1 && *it == col) ? c->it_nzval[it - c->it_rowval.data()] : -1;
    };
    // segment [segs[row], segs[row+1]) of every block row: rows are laid out back to back
    std::vector<int64_t> segs(nb + 1, 0);
    if (sparse) { int64_t cur = 0; for (int64_t row = 0; row < nb; ++row) { segs[row] = cur;
            for (int64_t q = c->it_colptr[row]; q < c->it_colptr[row + 1]; ++q) cur += (int64_t)c->blocksizes[row] * c->blocksizes[c->it_rowval[q]]; }
        segs[nb] = nnz_data; }

    // ---- info ----------------------------------------------------------------------------------------
    nlls_info& I = c->info; memset(&I, 0, sizeof I);
    I.is_sparse = sparse; I.nvar = nvar; I.nblocks = nb; I.ndof = ndof; I.nnz_data = nnz_data; I.nblocks_stored = (int64_t)c->it_rowval.size();
    I.ncost = ncost_total; I.var_storage = c->var_off[nvar];

    // ---- device buffers ----------------------------------------------------------------------------------
    HIPCHK(hipSetDevice(c->device));
    if (!getenv("NLLS_NO_ARENA")) {
        // reserve the hot arena first (compact_hot_set fills it at the end of the upload).  Estimate: A, b, x, three variable sets, per (cost, slot)
        // incidence its list record, the cost-order arrays, per block the elimination's inverse + descriptors, and for the reduced system the
        // band / tile workspaces -- generous (the slack is never touched), and harmless when short
        size_t est = 8 * ((size_t)nnz_data + 2 * (size_t)ndof + 3 * (size_t)I.var_storage + 64);
        for (int g = 0; g < ngroups; ++g) { const ResDesc& d = desc[g]; const size_t nc = (size_t)groups[g].ncost, nd = (size_t)std::max(d.ndeps, 1), nda = (size_t)std::max(d.ndata, 1);
            est += nc * nd * (8 * nda + 8 * nd + 8) + nc * (8 * nda + 4 * nd); }
        est += (size_t)nb * 256 + (size_t)ndof * 64;
        est += est / 4 + ((size_t)64 << 20);
        (void)c->arena_pre.alloc(est);
    }
    for (int k = 0; k < 3; ++k) { HIPCHK(c->vars[k].alloc(std::max<size_t>(I.var_storage, 1))); HIPCHK(hipMemset(c->vars[k].p, 0, sizeof(double) * std::max<size_t>(I.var_storage, 1))); c->vars_slot[k] = k; }
    HIPCHK(c->A.alloc(std::max<int64_t>(nnz_data, 1))); HIPCHK(hipMemset(c->A.p, 0, sizeof(double) * std::max<int64_t>(nnz_data, 1)));
    HIPCHK(c->b.alloc(std::max<int64_t>(ndof, 1))); HIPCHK(hipMemset(c->b.p, 0, sizeof(double) * std::max<int64_t>(ndof, 1)));
    HIPCHK(c->x.alloc(std::max<int64_t>(ndof, 1))); HIPCHK(hipMemset(c->x.p, 0, sizeof(double) * std::max<int64_t>(ndof, 1)));
    HIPCHK(c->d_var_kind.upload(c->var_kind)); HIPCHK(c->d_var_dim.upload(c->var_dim)); HIPCHK(c->d_var_off.upload(c->var_off));
    { std::vector<uint32_t> vb(nvar); for (int64_t i = 0; i < nvar; ++i) vb[i] = bi[i] ? (uint32_t)c->boffsets[bi[i] - 1] : DEST_NONE; HIPCHK(c->d_var_boff.upload(vb)); }
    HIPCHK(c->d_diag_off.upload(c->diag_off)); HIPCHK(c->d_blocksizes.upload(c->blocksizes));
    if (!c->h_scalars) { HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&c->h_scalars), 64 * sizeof(double), hipHostMallocMapped | hipHostMallocCoherent)); memset(c->h_scalars, 0, 64 * sizeof(double));
        if (hipHostGetDevicePointer(reinterpret_cast<void**>(&c->h_scalars_dev), c->h_scalars, 0) != hipSuccess) c->h_scalars_dev = nullptr; }
    c->S_zeroed = false;
    HIPCHK(c->scalars.alloc(64));

    // ---- observation sharding: costs are owned by the rank that owns their eliminated block ----------
    c->elim_selected = false;
    int nranks = c->nranks, rank = c->rank;
    // A dense system, or a block-sparse one without an eliminated variable set to partition by: REPLICAS -- every rank runs the whole problem as rank 0 of 1 and no
    // collective is entered (comm_reduce is a no-op, nlls_lm_trial takes the single-GPU route); nlls_get_shard_info()[5] tells the host.  A pre-sharded upload holds
    // only this rank's share and cannot be replicated: refused as before.
    auto replicate = [&]() { c->replicated = true; c->rank = rank = 0; c->nranks = nranks = 1; };
    std::vector<int32_t> owner_of_block(nb, 0);           // eliminated blocks: owning rank; reduced blocks: 0
    std::vector<std::vector<uint8_t>> mine(ngroups);
    for (int g = 0; g < ngroups; ++g) mine[g].assign(groups[g].ncost, 1);
    c->local_ncost = ncost_total;
    if (nranks > 1) {
        if (!sparse) { if (flags & NLLS_FLAG_PRESHARDED) return fail(c, NLLS_ERR_UNSUPPORTED, "sharding needs the block-sparse path"); replicate(); }
    }
    if (nranks > 1) {
        c->info = I;
        int rc0 = select_elimination(c, flags); if (rc0 != NLLS_OK) return rc0;
        if (!c->nelim && (flags & NLLS_FLAG_PRESHARDED)) return fail(c, NLLS_ERR_UNSUPPORTED, "sharding needs an eliminated (Schur) variable set to partition by");
        if (!c->nelim) { replicate(); c->elim_selected = false; }
    }
    if (nranks > 1) {
        if (flags & NLLS_FLAG_PRESHARDED) {                   // the caller has partitioned: everything uploaded here is this rank's
            for (int64_t k = 0; k < nb; ++k) if (c->is_elim[k]) owner_of_block[k] = rank;
        } else {
        // weight of an eliminated block = number of cost blocks touching it; contiguous ranges of equal weight
        std::vector<int64_t> w(nb, 0);
        for (int g = 0; g < ngroups; ++g) { const int nd = desc[g].ndeps;
            for (int64_t k = 0; k < groups[g].ncost; ++k) for (int s = 0; s < nd; ++s) { uint64_t r = bi[groups[g].varind[k * nd + s] - 1]; if (r && c->is_elim[r - 1]) w[r - 1]++; } }
        int64_t total = 0; for (int64_t k = 0; k < nb; ++k) if (c->is_elim[k]) total += w[k];
        { int64_t cum = 0; for (int64_t k = 0; k < nb; ++k) if (c->is_elim[k]) { int r = (int)std::min<int64_t>(nranks - 1, total ? (cum * nranks) / total : 0); owner_of_block[k] = r; cum += w[k]; } }
        c->local_ncost = 0;
        for (int g = 0; g < ngroups; ++g) { const int nd = desc[g].ndeps;
            for (int64_t k = 0; k < groups[g].ncost; ++k) { int own = (int)(k % nranks);
                for (int s = 0; s < nd; ++s) { uint64_t r = bi[groups[g].varind[k * nd + s] - 1]; if (r && c->is_elim[r - 1]) { own = owner_of_block[r - 1]; break; } }
                mine[g][k] = own == rank; c->local_ncost += mine[g][k]; } }
        }
    }
    c->owner_of_block = owner_of_block;
    // ---- per-group lists ---------------------------------------------------------------------------------
    c->groups.resize(ngroups);
    std::vector<int32_t> row_nlists(nb, 0);         // how many entry lists touch a row
    std::vector<std::vector<HostList>> hl(ngroups);
    int64_t npartials = 0;
    for (int g = 0; g < ngroups; ++g) {
        Group& G = c->groups[g]; const ResDesc& d = desc[g]; const nlls_cost_group& in = groups[g];
        G.res_kind = in.res_kind; G.ndeps = d.ndeps; G.ndata = d.ndata; G.nres = d.nres; G.adaptive = d.adaptive; G.ncost = in.ncost;
        G.rk.kind = in.robust_kind; G.rk.p0 = in.robust_params[0]; G.rk.p1 = in.robust_params[1];
        // cost-order arrays
        std::vector<uint32_t> fixedcost;
        { std::vector<double> hd; std::vector<uint32_t> hv; hd.reserve((size_t)in.ncost * d.ndata); hv.reserve((size_t)in.ncost * d.ndeps);
          int64_t nl = 0;
          for (int64_t k = 0; k < in.ncost; ++k) { if (!mine[g][k]) continue;
              for (int q = 0; q < d.ndata; ++q) hd.push_back(in.data[k * d.ndata + q]);
              bool any = false;
              for (int s = 0; s < d.ndeps; ++s) { hv.push_back(c->var_off[in.varind[k * d.ndeps + s] - 1]); any |= bi[in.varind[k * d.ndeps + s] - 1] != 0; }
              if (!any) fixedcost.push_back((uint32_t)nl);
              ++nl; }
          G.ncost = nl;
          G.local_of.clear();
          if (nranks > 1 && !(flags & NLLS_FLAG_PRESHARDED)) { G.local_of.assign((size_t)in.ncost, -1); int32_t q = 0; for (int64_t k = 0; k < in.ncost; ++k) if (mine[g][k]) G.local_of[(size_t)k] = q++; }
          HIPCHK(G.data.upload(hd)); HIPCHK(G.voff.upload(hv)); }
        G.nfixedcost = (int64_t)fixedcost.size(); HIPCHK(G.fixedcost.upload(fixedcost));
        npartials += (G.nfixedcost + 255) / 256;
        hl[g].resize(d.ndeps);
        if (!sparse || is_dyn_kind(in.res_kind)) continue;      // (dynamic-size groups take no entry lists: see the DenseList built for them below)
        for (int s = 0; s < d.ndeps; ++s) {
            HostList& L = hl[g][s];
            // counting sort of the incidences (cost, s) by block row
            std::vector<int64_t> cnt(nb + 1, 0);
            for (int64_t k = 0; k < in.ncost; ++k) { if (!mine[g][k]) continue; uint64_t r = bi[in.varind[k * d.ndeps + s] - 1]; if (r) cnt[r]++; }
            for (int64_t r = 0; r < nb; ++r) if (cnt[r + 1]) { L.rows.push_back(r); row_nlists[r]++; }
            std::vector<int64_t> pos(nb + 2, 0); for (int64_t r = 1; r <= nb; ++r) pos[r + 1] = pos[r] + cnt[r];
            L.cost.resize(pos[nb + 1]);
            L.rowptr.resize(L.rows.size() + 1);
            for (size_t i = 0; i < L.rows.size(); ++i) L.rowptr[i] = pos[L.rows[i] + 1];
            L.rowptr[L.rows.size()] = pos[nb + 1];
            for (int64_t k = 0; k < in.ncost; ++k) { if (!mine[g][k]) continue; uint64_t r = bi[in.varind[k * d.ndeps + s] - 1]; if (r) L.cost[pos[r]++] = k; }
        }
    }
    // ---- tiles ------------------------------------------------------------------------------------------------
    std::vector<int64_t> zero_off; std::vector<uint32_t> zero_len, zero_b_off, zero_b_len;
    std::vector<uint8_t> row_zero(nb, 0);
    bool all_owner = true;
    if (sparse) for (int g = 0; g < ngroups; ++g) {
        Group& G = c->groups[g]; const ResDesc& d = desc[g]; const nlls_cost_group& in = groups[g];
        std::vector<SlotHost> sh(d.ndeps);
        for (int s = 0; s < d.ndeps; ++s) {
            HostList& L = hl[g][s]; EntryList& E = G.lists[s]; E.slot = s; E.n = (int64_t)L.cost.size();
            if (E.n == 0) continue;
            if (E.n > 0xFFFFFFF0ll) return fail(c, NLLS_ERR_UNSUPPORTED, "entry list exceeds 32-bit indices");
            const size_t nrows = L.rows.size();
            std::vector<RowInfo> rinfo(nrows);
            std::vector<Tile> light, heavy;
            std::vector<uint32_t> dest((size_t)E.n * d.ndeps, DEST_NONE);
            std::vector<int64_t> entry_base(E.n, 0);   // image base (A.data offset) of the entry's tile, -1 => DIRECT (absolute)
            const bool force_atomic = (flags & NLLS_FLAG_FORCE_ATOMIC) != 0;
            auto close_light = [&](size_t r0, size_t r1, bool partial) {
                Tile t{}; t.e0 = (uint32_t)L.rowptr[r0]; t.e1 = (uint32_t)L.rowptr[r1]; t.row0 = (uint32_t)r0; t.nrows = (uint32_t)(r1 - r0);
                int64_t br0 = L.rows[r0], br1 = L.rows[r1 - 1];
                t.data_off = segs[br0]; t.data_len = (uint32_t)(segs[br1 + 1] - segs[br0]);
                t.b_off = (uint32_t)c->boffsets[br0]; t.b_len = (uint32_t)(c->boffsets[br1 + 1] - c->boffsets[br0]);
                t.flags = partial ? TILE_PARTIAL : 0;
                for (size_t r = r0; r < r1; ++r) { int64_t br = L.rows[r];
                    rinfo[r].diag_off = (uint32_t)(c->diag_off[br] - t.data_off); rinfo[r].b_off = t.data_len + (uint32_t)(c->boffsets[br] - c->boffsets[br0]);
                    if (partial) row_zero[br] = 1;
                    for (int64_t e = L.rowptr[r]; e < L.rowptr[r + 1]; ++e) { entry_base[e] = t.data_off; dest[(size_t)e * d.ndeps + s] = (uint32_t)(r - r0); } }
                { const uint32_t dsz = (uint32_t)c->blocksizes[br0]; const uint32_t per_row = ACC_COPIES * (dsz * (dsz + 1) / 2 + dsz);
                  E.light_lds = std::max(E.light_lds, t.data_len + t.b_len + t.nrows * per_row); }
                light.push_back(t); if (partial) all_owner = false;
            };
            // (NLLS_HEAVY_MAX_ENTRIES: A/B knob -- shorter heavy tiles spread a few long rows over more workgroups, at the price of atomic flushes)
            const uint32_t HEAVY_MAX_ENTRIES = [] { const char* e = getenv("NLLS_HEAVY_MAX_ENTRIES"); const int v = e ? atoi(e) : 0; return v >= 128 ? (uint32_t)v : HEAVY_MAX_ENTRIES_DEFAULT; }();
            size_t r = 0;
            while (r < nrows) {
                int64_t br = L.rows[r]; int64_t ne = L.rowptr[r + 1] - L.rowptr[r]; int64_t seglen = segs[br + 1] - segs[br];
                bool is_heavy = ne > HEAVY_ROW_ENTRIES || seglen + c->blocksizes[br] > LIGHT_IMG_MAX;
                if (is_heavy) {
                    bool direct = seglen + c->blocksizes[br] > LIGHT_IMG_MAX;
                    bool split = ne > HEAVY_MAX_ENTRIES; bool shared = row_nlists[br] > 1 || force_atomic;
                    for (int64_t e0 = L.rowptr[r]; e0 < L.rowptr[r + 1]; e0 += HEAVY_MAX_ENTRIES) {
                        Tile t{}; t.e0 = (uint32_t)e0; t.e1 = (uint32_t)std::min<int64_t>(e0 + HEAVY_MAX_ENTRIES, L.rowptr[r + 1]); t.row0 = (uint32_t)r; t.nrows = 1;
                        t.data_off = segs[br]; t.data_len = direct ? 0u : (uint32_t)seglen; t.b_off = (uint32_t)c->boffsets[br]; t.b_len = (uint32_t)c->blocksizes[br];
                        t.flags = (direct ? TILE_DIRECT : 0) | ((split || shared || direct) ? TILE_PARTIAL : 0);
                        heavy.push_back(t);
                        E.heavy_lds = std::max(E.heavy_lds, t.data_len);
                        for (int64_t e = t.e0; e < t.e1; ++e) { entry_base[e] = direct ? 0 : t.data_off; dest[(size_t)e * d.ndeps + s] = 0; }
                        if (t.flags & TILE_PARTIAL) { row_zero[br] = 1; all_owner = false; }
                    }
                    rinfo[r].diag_off = (uint32_t)(c->diag_off[br] - segs[br]); rinfo[r].b_off = 0;
                    ++r; continue;
                }
                // light run: consecutive block rows present in this list
                size_t r1 = r; uint32_t ents = 0; int64_t img = 0; bool partial = force_atomic;
                while (r1 < nrows) {
                    int64_t brr = L.rows[r1]; int64_t nee = L.rowptr[r1 + 1] - L.rowptr[r1]; const int64_t dsz = c->blocksizes[brr];
                    int64_t sl = segs[brr + 1] - segs[brr] + dsz + ACC_COPIES * (dsz * (dsz + 1) / 2 + dsz);
                    if (r1 > r && brr != L.rows[r1 - 1] + 1) break;
                    if (nee > HEAVY_ROW_ENTRIES || sl > LIGHT_IMG_MAX) break;
                    if (r1 > r && (ents + nee > LIGHT_MAX_ENTRIES || img + sl > LIGHT_IMG_MAX || r1 - r >= 0xFFFF)) break;
                    bool sh = row_nlists[brr] > 1;
                    if (r1 > r && sh != partial && !force_atomic) break;      // keep exclusive and shared rows in separate tiles
                    partial = sh || force_atomic; ents += (uint32_t)nee; img += sl; ++r1;
                }
                close_light(r, r1, partial); r = r1;
            }
            // off-diagonal destinations + flags
            for (size_t rr = 0; rr < nrows; ++rr) { int64_t br = L.rows[rr];
                for (int64_t e = L.rowptr[rr]; e < L.rowptr[rr + 1]; ++e) { int64_t k = L.cost[e];
                    uint32_t own = dest[(size_t)e * d.ndeps + s] & OWN_ROW_MASK;
                    bool first_free = true; for (int t = 0; t < s; ++t) if (bi[in.varind[k * d.ndeps + t] - 1]) first_free = false;
                    if (first_free) own |= OWN_COST_OWNER;
                    own |= (uint32_t)((e - L.rowptr[rr]) & (ACC_COPIES - 1)) << OWN_COPY_SHIFT;
                    if (d.adaptive && bi[in.varind[k * d.ndeps] - 1]) own |= OWN_KERNEL_FREE;
                    dest[(size_t)e * d.ndeps + s] = own;
                    for (int t = 0; t < d.ndeps; ++t) { if (t == s) continue; uint64_t bt = bi[in.varind[k * d.ndeps + t] - 1];
                        if (!bt || (int64_t)bt - 1 > br) continue;
                        int64_t bo = block_off(br, (int64_t)bt - 1); if (bo < 0) return fail(c, NLLS_ERR_INVALID_ARG, "internal: block missing from the pattern");
                        dest[(size_t)e * d.ndeps + t] = (uint32_t)(bo - entry_base[e]); } } }
            // are the off-diagonal destinations unique (each stored block written by exactly one entry)?  Then the
            // kernel uses plain LDS stores for them and exclusive tiles need not zero their image.
            bool unique = true;
            { std::vector<uint32_t> tmp;
              for (size_t rr = 0; rr < nrows && unique; ++rr) { tmp.clear();
                  for (int64_t e = L.rowptr[rr]; e < L.rowptr[rr + 1]; ++e) for (int t = 0; t < d.ndeps; ++t) if (t != s && dest[(size_t)e * d.ndeps + t] != DEST_NONE) tmp.push_back(dest[(size_t)e * d.ndeps + t]);
                  std::sort(tmp.begin(), tmp.end()); for (size_t i = 1; i < tmp.size(); ++i) if (tmp[i] == tmp[i - 1]) { unique = false; break; } } }
            E.unique_dest = unique;
            if (unique) for (auto& tl : light) if (!(tl.flags & TILE_PARTIAL)) tl.flags |= TILE_NOZERO;
            // upload the list
            { std::vector<double> hd((size_t)E.n * d.ndata); std::vector<uint32_t> hv((size_t)E.n * d.ndeps);
              for (int64_t e = 0; e < E.n; ++e) { int64_t k = L.cost[e];
                  for (int q = 0; q < d.ndata; ++q) hd[(size_t)e * d.ndata + q] = in.data[k * d.ndata + q];
                  for (int q = 0; q < d.ndeps; ++q) hv[(size_t)e * d.ndeps + q] = c->var_off[in.varind[k * d.ndeps + q] - 1]; }
              HIPCHK(E.data.upload(hd)); HIPCHK(E.voff.upload(hv)); HIPCHK(E.rows.upload(rinfo));
              HIPCHK(E.dest.upload(dest));
              // compact form for the heavy pass (see EntryList::compact)
              E.compact = false; E.own_flags = 0;
              if (light.empty() && !heavy.empty() && d.ndeps >= 2 && E.n > 0) {
                  bool ok = true; const uint32_t fl = dest[(size_t)0 * d.ndeps + s] & (OWN_COST_OWNER | OWN_KERNEL_FREE);
                  for (int64_t e = 0; e < E.n && ok; ++e) { if ((dest[(size_t)e * d.ndeps + s] & (OWN_COST_OWNER | OWN_KERNEL_FREE)) != fl) ok = false;
                      for (int t = 0; t < d.ndeps; ++t) if (t != s && dest[(size_t)e * d.ndeps + t] != DEST_NONE) ok = false; }
                  if (ok) { std::vector<uint32_t> ho((size_t)E.n * (d.ndeps - 1));
                      for (int64_t e = 0; e < E.n; ++e) { int q2 = 0; for (int t = 0; t < d.ndeps; ++t) if (t != s) ho[(size_t)e * (d.ndeps - 1) + q2++] = hv[(size_t)e * d.ndeps + t]; }
                      HIPCHK(E.hvoff.upload(ho)); E.compact = true; E.own_flags = fl; } } }
            E.nlight = (int64_t)light.size(); E.nheavy = (int64_t)heavy.size();
            HIPCHK(E.light.upload(light)); HIPCHK(E.heavy.upload(heavy));
            npartials += E.nlight + E.nheavy;
            sh[s].dest = std::move(dest); sh[s].light = std::move(light); sh[s].heavy = std::move(heavy);
        }
        { const int rcf = build_fold(c, G, d, in, bi, hl[g], sh, segs, row_nlists, row_zero, flags); if (rcf != NLLS_OK) return rcf; }
        G.cost_list = -1;       // (see Group::cost_list)
        for (int s2 = 0; s2 < d.ndeps; ++s2) { const EntryList& E = G.lists[s2]; if (E.n == G.ncost && G.ncost > 0 && E.nlight > 0 && E.nheavy == 0) { G.cost_list = s2; break; } }
        for (int s2 = 0; s2 < d.ndeps; ++s2) std::vector<int64_t>().swap(hl[g][s2].cost);
    }
    std::vector<int64_t> red_off; std::vector<uint32_t> red_len, red_dst, red_which;   // stage-0 reduce ranges
    if (sparse && nranks > 1) {
        uint32_t dst = 1;   // slot 0 carries the cost
        for (int64_t r = 0; r < nb; ++r) if (!c->is_elim[r]) { row_zero[r] = 1;
            red_off.push_back(segs[r]); red_len.push_back((uint32_t)(segs[r + 1] - segs[r])); red_dst.push_back(dst); red_which.push_back(0); dst += (uint32_t)(segs[r + 1] - segs[r]);
            red_off.push_back(c->boffsets[r]); red_len.push_back((uint32_t)c->blocksizes[r]); red_dst.push_back(dst); red_which.push_back(1); dst += (uint32_t)c->blocksizes[r]; }
        c->nred_ranges = (int64_t)red_off.size(); c->redbuf_len = dst;
        HIPCHK(c->d_red_off.upload(red_off)); HIPCHK(c->d_red_len.upload(red_len)); HIPCHK(c->d_red_dst.upload(red_dst)); HIPCHK(c->d_red_which.upload(red_which));
        HIPCHK(c->redbuf.alloc(dst));
    } else { c->nred_ranges = 0; c->redbuf_len = 0; }
    if (sparse) for (int g = 0; g < ngroups; ++g) {
        // dynamic-size groups of a block-sparse system: one entry per cost block with a free variable, accumulated with atomics straight into the variable's
        // (full) diagonal block -- the row is zeroed in front of every sweep
        Group& G = c->groups[g]; const ResDesc& d = desc[g]; const nlls_cost_group& in = groups[g];
        if (!is_dyn_kind(in.res_kind)) continue;
        std::vector<double> hd; std::vector<uint32_t> hv, hb, ha;
        for (int64_t k = 0; k < in.ncost; ++k) { const int64_t v = in.varind[k] - 1; if (!bi[v]) continue;
            const int64_t blk = (int64_t)bi[v] - 1;
            row_zero[blk] = 1;                                   // (every rank zeroes the row: it is summed over ranks)
            if (!mine[g][k]) continue;
            for (int q = 0; q < d.ndata; ++q) hd.push_back(in.data[k * d.ndata + q]);
            hv.push_back(c->var_off[v]); hb.push_back((uint32_t)c->boffsets[blk]); ha.push_back((uint32_t)c->diag_off[blk]); row_zero[blk] = 1; }
        G.dense.n = (int64_t)hv.size();
        HIPCHK(G.dense.data.upload(hd)); HIPCHK(G.dense.voff.upload(hv)); HIPCHK(G.dense.brow.upload(hb)); HIPCHK(G.dense.aoff.upload(ha));
        npartials += (G.dense.n + 255) / 256 + 1;
        all_owner = false;
    }
    if (sparse) {
        for (int64_t r = 0; r < nb; ++r) if (row_zero[r]) { zero_off.push_back(segs[r]); zero_len.push_back((uint32_t)(segs[r + 1] - segs[r])); zero_b_off.push_back((uint32_t)c->boffsets[r]); zero_b_len.push_back((uint32_t)c->blocksizes[r]); }
    } else {
        // dense linear system: one entry per cost with at least one free variable
        for (int g = 0; g < ngroups; ++g) {
            Group& G = c->groups[g]; const ResDesc& d = desc[g]; const nlls_cost_group& in = groups[g];
            std::vector<double> hd; std::vector<uint32_t> hv, hb;
            for (int64_t k = 0; k < in.ncost; ++k) { bool any = false; for (int s = 0; s < d.ndeps; ++s) any |= bi[in.varind[k * d.ndeps + s] - 1] != 0; if (!any) continue;
                for (int q = 0; q < d.ndata; ++q) hd.push_back(in.data[k * d.ndata + q]);
                for (int s = 0; s < d.ndeps; ++s) { int64_t v = in.varind[k * d.ndeps + s] - 1; hv.push_back(c->var_off[v]); hb.push_back(bi[v] ? (uint32_t)c->boffsets[bi[v] - 1] : DEST_NONE); } }
            G.dense.n = (int64_t)(hv.size() / std::max(d.ndeps, 1));
            HIPCHK(G.dense.data.upload(hd)); HIPCHK(G.dense.voff.upload(hv)); HIPCHK(G.dense.brow.upload(hb));
            npartials += (G.dense.n + 255) / 256 + 1;
        }
        all_owner = false;
    }
    c->nzero = (int64_t)zero_off.size();
    HIPCHK(c->d_zero_off.upload(zero_off)); HIPCHK(c->d_zero_len.upload(zero_len)); HIPCHK(c->d_zero_b_off.upload(zero_b_off)); HIPCHK(c->d_zero_b_len.upload(zero_b_len));
    for (int g = 0; g < ngroups; ++g) npartials += (c->groups[g].ncost + 255) / 256 + 1;
    for (int g = 0; g < ngroups; ++g) if (is_dyn_kind(c->groups[g].res_kind)) npartials += 3 * c->groups[g].ncost + 3;   // dynamic-size blocks: one partial per block
    c->npartials = std::max<int64_t>(npartials + 16, 4096);
    // (an LM trial keeps the cost partials behind the post-solve partials, at TRIAL_COST_POFS: both are reduced by ONE finishing launch)
    HIPCHK(c->partials.alloc(TRIAL_COST_POFS + c->npartials));
    I.owner_path = all_owner ? 1 : 0;

    { int64_t lw = 0, ld = 0;
      for (int64_t r = 0; r < nb; ++r) if (nranks == 1 || !c->is_elim[r] || owner_of_block[r] == rank) { lw += sparse ? segs[r + 1] - segs[r] : 0; ld += c->blocksizes[r]; }
      c->local_nnz_data = sparse ? lw : nnz_data; c->local_ndof = ld; }
    int rc = build_schur(c, flags);
    if (rc != NLLS_OK) return rc;
    rc = build_mf(c, ngroups, groups, bi, flags);
    if (rc != NLLS_OK) return rc;
    rc = compact_hot_set(c);
    if (rc != NLLS_OK) return rc;
    c->ready = true;
    return NLLS_OK;
}


// ---- the matrix-free LM trial's block list (nlls_mf.hip) ------------------------------------------------------------------
// Eligible: one rank, ONE cost group of a two-slot residual kind, one slot's variables all eliminated on the fast path (Euclidean, at most three unknowns), the other
// slot's all reduced, no fixed variable, exactly one block per (eliminated block, neighbour) pair -- i.e. the member's blocks ARE the column blocks of its [E], one each.
// Leaves c->mf_ok false (and nothing else changed) whenever the problem does not qualify: nlls_lm_trial then takes the materialised path.
int build_mf(nlls_ctx* c, int32_t ngroups, const nlls_cost_group* groups, const uint64_t* bi, int32_t flags) {
    c->mf_ok = false; c->mf_group = -1; c->mf_ps = -1; c->mf_step = false; c->mf_use = false; c->mf_q.release(); c->d_mf_desc.release(); c->mf_nbig = 0;
    for (Group& G : c->groups) { G.mf_data.release(); G.mf_voff.release(); }
    { const char* e = getenv("NLLS_MATERIALIZE"); c->mf_on = !(e && e[0] == '1'); }
    // (the gather index of build_schur was built for this trial: without it -- or when the problem turns out not to qualify below -- it goes again, unless the flag asked for it)
    struct Drop { nlls_ctx* c; bool keep; ~Drop() { if (!c->mf_ok && !keep) { c->gather_ready = false; c->slab.release(); c->d_slab_off.release(); c->d_slab_groups.release(); c->d_gjobs.release(); c->d_gcons.release(); c->n_gjobs = 0; } } } drop{c, (flags & NLLS_FLAG_DETERMINISTIC) != 0};
    if ((flags & NLLS_FLAG_MATERIALIZE) || ngroups != 1 || c->nranks != 1 || !c->info.is_sparse || !c->gather_ready || c->h_slab_off.size() != c->h_elim_desc.size()) return NLLS_OK;
    const nlls_cost_group& in = groups[0]; ResDesc d;
    if (!res_desc(in.res_kind, d) || is_dyn_kind(in.res_kind) || d.ndeps != 2 || d.adaptive || d.nres <= 0) return NLLS_OK;
    if (c->n_fast_groups == 0 || c->n_slow_groups != 0 || c->n_fast_members != (int64_t)c->h_erow.size() || !c->fast_all_euclid) return NLLS_OK;
    if (c->solve_mode == SOLVE_SMALL || c->solve_mode == SOLVE_TSPARSE || c->h_elim_desc.size() != (size_t)c->n_fast_groups || in.ncost <= 0 || in.ncost > 0xFFFFFFF0ll) return NLLS_OK;
    Group& G = c->groups[0];
    if (G.nfixedcost != 0) return NLLS_OK;
    const int64_t nb = c->info.nblocks;
    int ps = -1;
    { const uint64_t b0 = bi[in.varind[0] - 1], b1 = bi[in.varind[1] - 1]; if (!b0 || !b1) return NLLS_OK; ps = c->is_elim[b0 - 1] ? 0 : 1; }
    const int cs = 1 - ps; const int dp = var_dof(d.sk[ps], d.sd[ps]), dc = var_dof(d.sk[cs], d.sd[cs]);
    if (dp != c->fast_dv || dp < 1 || dp > 3 || dc < 1) return NLLS_OK;
    // the blocks of every eliminated block row, counting sort by that row
    std::vector<int64_t> cnt(nb + 1, 0);
    for (int64_t k = 0; k < in.ncost; ++k) { const uint64_t bp = bi[in.varind[k * 2 + ps] - 1], bc = bi[in.varind[k * 2 + cs] - 1];
        if (!bp || !bc || !c->is_elim[bp - 1] || c->is_elim[bc - 1]) return NLLS_OK;
        cnt[bp]++; }
    std::vector<int64_t> pos(nb + 1, 0); for (int64_t r = 0; r < nb; ++r) pos[r + 1] = pos[r] + cnt[r + 1];
    std::vector<int64_t> byrow((size_t)in.ncost); { std::vector<int64_t> cur(pos.begin(), pos.end() - 1); for (int64_t k = 0; k < in.ncost; ++k) byrow[cur[bi[in.varind[k * 2 + ps] - 1] - 1]++] = k; }
    std::vector<double> hd; std::vector<uint32_t> hv; hd.reserve((size_t)in.ncost * d.ndata); hv.reserve((size_t)in.ncost * 2);
    // launch order: the supernodes of several batches (one workgroup each), then those of ONE batch (one wavefront each).  Members per batch B <= 64 / (blocks per member):
    // the value that gives the four wavefronts the shortest longest share -- rounds x (fixed work per batch ~ four members' + B)
    std::vector<MfDesc> desc; desc.reserve(c->h_elim_desc.size()); uint32_t ecap = 0, imgmax = 0; int64_t nobs = 0;
    const int bmax = mf_batch_max(), nw = mf_elim_waves();
    for (int pass = 0; pass < 2; ++pass) for (size_t ei = 0; ei < c->h_elim_desc.size(); ++ei) { const ElimDesc& e0 = c->h_elim_desc[ei];
        const int nd = (int)e0.nd; if (nd % dc || nd + 1 > 80 || nd / dc > 64 || nd / dc < 1 || e0.nmem > 128 || e0.nmem < 1) return NLLS_OK;
        // (members per batch: one lane per cost block, at most MF_BMAX, and -- four workgroups per CU -- a slab that leaves the wavefront's LDS region within 2446 doubles: with
        //  2514 the launch held three workgroups per CU and took 120 instead of 83 us at BASELINE config 4)
        const int ncb = nd / dc, TR0 = (nd + 1 + 15) / 16;
        int bfit = 1; while (bfit < bmax && mf_wave_doubles(mf_slab_doubles(bfit + 1, dp, TR0), dp) <= 2446u) ++bfit;
        const int bcap = std::max(1, std::min({64 / ncb, bmax, bfit}));
        const bool tiny = (int)e0.nmem <= bcap; if (tiny != (pass == 1)) continue;
        int B = tiny ? (int)e0.nmem : bcap;      // (a supernode of one batch: the batch is its members)
        if (!tiny) { int64_t best = -1; for (int b2 = 1; b2 <= bcap; ++b2) { const int64_t nbt = ((int64_t)e0.nmem + b2 - 1) / b2, rounds = (nbt + nw - 1) / nw, cost = rounds * (4 + b2); if (best < 0 || cost < best) { best = cost; B = b2; } } }
        MfDesc e{e0.v0, e0.nmem, e0.nd, e0.rc_off, e0.eb0, (uint32_t)nobs, (uint32_t)B, c->h_slab_off[ei]};
        imgmax = std::max<uint32_t>(imgmax, (uint32_t)(ncb * (ncb + 1) / 2 * dc * dc + nd + 4));
        const int TR = (nd + 1 + 15) / 16; ecap = std::max<uint32_t>(ecap, mf_slab_doubles(B, dp, TR));
        for (uint32_t m = 0; m < e.nmem; ++m) { const uint32_t v = e.v0 + m; const int64_t row = c->h_erow[v];
            const int64_t q0 = c->h_eptr[v], q1 = c->h_eptr[v + 1];
            if (q1 - q0 != ncb || pos[row + 1] - pos[row] != ncb) return NLLS_OK;
            for (int64_t q = q0; q < q1; ++q) { const int64_t nbk = c->h_enbr_block[q]; if (nbk < 0 || c->blocksizes[nbk] != dc) return NLLS_OK;
                int64_t found = -1;
                for (int64_t t = pos[row]; t < pos[row + 1]; ++t) { const int64_t k = byrow[t]; if ((int64_t)bi[in.varind[k * 2 + cs] - 1] - 1 == nbk) { if (found >= 0) return NLLS_OK; found = k; } }
                if (found < 0) return NLLS_OK;
                for (int q2 = 0; q2 < d.ndata; ++q2) hd.push_back(in.data[found * d.ndata + q2]);
                hv.push_back(c->var_off[in.varind[found * 2] - 1]); hv.push_back(c->var_off[in.varind[found * 2 + 1] - 1]); ++nobs; } }
        desc.push_back(e);
        if (pass == 0) c->mf_nbig = (int64_t)desc.size();
    }
    if (nobs != in.ncost) return NLLS_OK;
    HIPCHK(G.mf_data.upload(hd)); HIPCHK(G.mf_voff.upload(hv)); HIPCHK(c->d_mf_desc.upload(desc)); HIPCHK(c->mf_q.alloc(mf_part_doubles(c->n_fast_groups, (160 + (int64_t)c->var_kind.size() / 64 + 8) / 4 + 2)));
    c->mf_ecap = ecap; c->mf_wsz = std::max(mf_wave_doubles(ecap, dp), (imgmax + 1) & ~1u); c->mf_lds = sizeof(double) * (size_t)c->mf_wsz * mf_elim_waves();      // (a wavefront's region also stages its supernode's share in slab layout)
    if (c->mf_lds > (size_t)150 * 1024) return NLLS_OK;
    c->mf_ok = true; c->mf_group = 0; c->mf_ps = ps;
    return NLLS_OK;
}

// Choice of the eliminated (Schur) set: an independent set of blocks -- no two share a stored block.
int select_elimination(nlls_ctx* c, int32_t flags) {
    const nlls_info& I0 = c->info; const int64_t nb = I0.nblocks;
    c->is_elim.assign(nb, 0); c->nelim = 0; c->elim_selected = true;
    std::vector<int64_t> tptr(nb + 1, 0), trow;
    if (I0.is_sparse) {
        for (int64_t q = 0; q < (int64_t)c->it_rowval.size(); ++q) tptr[c->it_rowval[q] + 1]++;
        for (int64_t k = 0; k < nb; ++k) tptr[k + 1] += tptr[k];
        trow.resize(c->it_rowval.size());
        std::vector<int64_t> cur(tptr.begin(), tptr.end() - 1);
        for (int64_t row = 0; row < nb; ++row) for (int64_t q = c->it_colptr[row]; q < c->it_colptr[row + 1]; ++q) trow[cur[c->it_rowval[q]]++] = row;
    }
    if (I0.is_sparse && !(flags & NLLS_FLAG_NO_SCHUR) && nb > 1) {
        std::vector<int32_t> deg(nb, 0);
        for (int64_t row = 0; row < nb; ++row) for (int64_t q = c->it_colptr[row]; q < c->it_colptr[row + 1]; ++q) { int64_t col = c->it_rowval[q]; if (col != row) { deg[row]++; deg[col]++; } }
        // candidate class = the block size with the most blocks; greedy independent set inside it, lowest degree first
        std::unordered_map<int, int64_t> classcount; for (int64_t k = 0; k < nb; ++k) classcount[c->blocksizes[k]]++;
        int best = -1; int64_t bestn = 0; for (auto& kv : classcount) if (kv.second > bestn || (kv.second == bestn && kv.first < best)) { best = kv.first; bestn = kv.second; }
        std::vector<uint8_t> blocked(nb, 0);
        // A block whose neighbours do not fit the LDS-staged elimination (a landmark seen by more than ~330 six-dof cameras: 150 KB of a CU's 160) is NOT a
        // candidate: it stays in the reduced system -- as a border block when it couples to a quarter of it (build_schur), a hub of the tile-sparse solver, or a
        // plain block of the dense one -- instead of taking the whole problem off the Schur path (rounds 1-3: NLLS_SUB_SCHUR_SHAPE, retry without elimination).
        auto fits_lds = [&](int64_t v) { size_t nd = 0;
            for (int64_t q = c->it_colptr[v]; q < c->it_colptr[v + 1]; ++q) if (c->it_rowval[q] != v) nd += (size_t)c->blocksizes[c->it_rowval[q]];
            for (int64_t q = tptr[v]; q < tptr[v + 1]; ++q) if (trow[q] != v) nd += (size_t)c->blocksizes[trow[q]];
            const size_t dv = (size_t)best; return sizeof(double) * (dv * dv + dv * nd + dv * (nd + 1)) + 28 * nd + 16 <= (size_t)150 * 1024; };
        std::vector<int64_t> order; for (int64_t k = 0; k < nb; ++k) if (c->blocksizes[k] == best && best <= NLLS_MAX_BLOCK_SZ && fits_lds(k)) order.push_back(k);    // (a dynamic-size variable's block -- up to 4096 unknowns -- is never eliminated: the Schur kernels stage a block in LDS)
        std::stable_sort(order.begin(), order.end(), [&](int64_t a, int64_t b) { return deg[a] < deg[b]; });
        for (int64_t v : order) { if (blocked[v]) continue; c->is_elim[v] = 1; c->nelim++;
            for (int64_t q = c->it_colptr[v]; q < c->it_colptr[v + 1]; ++q) blocked[c->it_rowval[q]] = 1;
            for (int64_t q = tptr[v]; q < tptr[v + 1]; ++q) blocked[trow[q]] = 1; }
        if (c->nelim * 2 < nb) { std::fill(c->is_elim.begin(), c->is_elim.end(), 0); c->nelim = 0; }   // not worth it unless most blocks go
    }
    return NLLS_OK;
}

// ---- ordering of the reduced system ---------------------------------------------------------------------------
// The reference analyses the sparsity of the full system once with a fill-reducing ordering (ldl_analyze, src/linearsystem.jl:52,68) and is
// therefore insensitive to how the caller numbers the variables.  Here the reduced system is solved as a bordered BAND, so the numbering of
// the reduced blocks IS the fill: a reverse Cuthill-McKee ordering (George & Liu's pseudo-peripheral start node, neighbours by ascending
// degree, components one after the other) of the reduced blocks' graph replaces the caller's order whenever it gives the narrower band.
// adj: symmetric adjacency lists (sorted, no self loops).  Returns perm[new position] = node.
std::vector<int32_t> rcm_order(const std::vector<std::vector<int32_t>>& adj) {
    const int32_t n = (int32_t)adj.size();
    std::vector<int32_t> order; order.reserve(n);
    std::vector<int32_t> level(n, -1), q; q.reserve(n);
    std::vector<uint8_t> done(n, 0);
    auto deg = [&](int32_t v) { return (int32_t)adj[v].size(); };
    // breadth-first level structure of the component of r among nodes not yet ordered; returns the eccentricity, q = the nodes in visiting order
    auto bfs = [&](int32_t r) { q.clear(); q.push_back(r); level[r] = 0; int32_t ecc = 0;
        for (size_t h = 0; h < q.size(); ++h) { const int32_t u = q[h]; ecc = level[u];
            for (int32_t w : adj[u]) if (!done[w] && level[w] < 0) { level[w] = level[u] + 1; q.push_back(w); } }
        return ecc; };
    auto clear_levels = [&]() { for (int32_t u : q) level[u] = -1; };
    std::vector<int32_t> starts(n); std::iota(starts.begin(), starts.end(), 0);
    std::stable_sort(starts.begin(), starts.end(), [&](int32_t a, int32_t b) { return deg(a) < deg(b); });
    std::vector<int32_t> nb_sorted;
    for (int32_t s0 : starts) { if (done[s0]) continue;
        // pseudo-peripheral node: restart from a minimum-degree node of the last level while the eccentricity grows
        int32_t r = s0, ecc = bfs(r);
        for (int it = 0; it < 32; ++it) {
            int32_t cand = -1; for (int32_t u : q) if (level[u] == ecc && (cand < 0 || deg(u) < deg(cand))) cand = u;
            clear_levels();
            if (cand < 0 || cand == r) { bfs(r); break; }
            const int32_t e2 = bfs(cand);
            if (e2 > ecc) { r = cand; ecc = e2; } else { clear_levels(); bfs(r); break; }
        }
        clear_levels();
        // Cuthill-McKee from r
        const size_t first = order.size();
        order.push_back(r); done[r] = 1;
        for (size_t h = first; h < order.size(); ++h) { const int32_t u = order[h];
            nb_sorted.clear(); for (int32_t w : adj[u]) if (!done[w]) { nb_sorted.push_back(w); done[w] = 1; }
            std::stable_sort(nb_sorted.begin(), nb_sorted.end(), [&](int32_t a, int32_t b) { return deg(a) < deg(b); });
            order.insert(order.end(), nb_sorted.begin(), nb_sorted.end()); }
    }
    std::reverse(order.begin(), order.end());
    return order;
}

// ---- Schur structures --------------------------------------------------------------------------------------
// The reference factors the FULL sparse system with LDLFactorizations (src/linearsolver.jl:29); there is no
// Schur complement in it.  Here an independent set of blocks (no two share a stored block) is
// eliminated first -- the same fill-reducing choice a minimum-degree ordering makes for bundle adjustment --
// and the remaining blocks form the reduced system, ordered [banded part | border blocks | rhs row].
int build_schur(nlls_ctx* c, int32_t flags) {
    const nlls_info& I0 = c->info; const int64_t nb = I0.nblocks;
    c->nelim_groups = 0; c->max_elim_dim = 0; c->max_nbr_dof = 0;
    c->damped_floor = (flags & NLLS_FLAG_NO_PIVOT_FLOOR) ? 0.0 : 1e-11;
    // (a re-upload -- or the retry without Schur elimination after an unsupported shape -- must not see the previous
    // attempt's supernode lists: the solve dispatches on these counters)
    c->n_fast_groups = 0; c->n_slow_groups = 0; c->n_fast_members = 0; c->n_fast_narrow = 0; c->n_fast_n60 = 0; c->fast_dv = 0;
    c->h_elim_desc.clear(); c->h_erow.clear(); c->h_eptr.clear(); c->h_enbr_block.clear(); c->mf_ok = false; c->mf_step = false; c->mf_use = false;
    c->tE_valid = false; c->S_zeroed = false; c->status_known_zero = false; c->step_cached = false; c->bcr.release();
    c->elim_slab = false; c->gather_ready = false; c->h_slab_off.clear(); c->tiles_zeroed = false; c->slab.release(); c->d_slab_off.release(); c->d_slab_groups.release(); c->d_gjobs.release(); c->d_gcons.release(); c->n_gjobs = 0;
    // (the solve also dispatches on the SIZE of these lists: an upload without elimination must not inherit them)
    c->d_elim_ptr.release(); c->d_elim_nbr.release(); c->d_elim_diag.release(); c->d_elim_boff.release(); c->d_elim_dim.release(); c->d_elim_group.release();
    c->d_fast_groups.release(); c->d_elim_desc.release(); c->d_elim_rc.release(); c->d_slow_groups.release(); c->d_slow_blocks.release(); c->d_fast_members.release(); c->Cinv.release(); c->tE.release();
    std::vector<int64_t> red_of(nb, -1);     // dof offset in the reduced system
    std::vector<std::vector<int32_t>> red_adj; std::vector<int64_t> red_adj_blocks;     // graph of the reduced non-border blocks (kept for the tile-sparse solver's symbolic phase)
    c->tsp.release();
    // transposed block lists: for a block v, the rows w > v that store block (w, v)
    std::vector<int64_t> tptr(nb + 1, 0), trow, tq;
    if (I0.is_sparse) {
        for (int64_t q = 0; q < (int64_t)c->it_rowval.size(); ++q) tptr[c->it_rowval[q] + 1]++;
        for (int64_t k = 0; k < nb; ++k) tptr[k + 1] += tptr[k];
        trow.resize(c->it_rowval.size()); tq.resize(c->it_rowval.size());
        std::vector<int64_t> cur(tptr.begin(), tptr.end() - 1);
        for (int64_t row = 0; row < nb; ++row) for (int64_t q = c->it_colptr[row]; q < c->it_colptr[row + 1]; ++q) { int64_t p = cur[c->it_rowval[q]]++; trow[p] = row; tq[p] = q; }
    }
    if (!c->elim_selected) { int rc0 = select_elimination(c, flags); if (rc0 != NLLS_OK) return rc0; }
    // ---- reduced ordering: blocks coupled to a large share of the system go last (border) so that they cause no fill
    std::vector<uint8_t> is_border(nb, 0);
    {
        std::vector<int64_t> cnt(nb, 0); int64_t nR = 0;
        for (int64_t v = 0; v < nb; ++v) { if (!c->is_elim[v]) { nR++; continue; }
            for (int64_t q = c->it_colptr[v]; q < c->it_colptr[v + 1]; ++q) if (c->it_rowval[q] != v) cnt[c->it_rowval[q]]++;
            for (int64_t q = tptr[v]; q < tptr[v + 1]; ++q) if (trow[q] != v) cnt[trow[q]]++; }
        std::vector<int64_t> rcnt(nb, 0);
        if (I0.is_sparse) for (int64_t row = 0; row < nb; ++row) { if (c->is_elim[row]) continue;
            for (int64_t q = c->it_colptr[row]; q < c->it_colptr[row + 1]; ++q) { int64_t col = c->it_rowval[q]; if (col != row && !c->is_elim[col]) { rcnt[row]++; rcnt[col]++; } } }
        int64_t nelim_all = c->nelim;
        c->nelim_all = c->nelim;
        if ((flags & NLLS_FLAG_PRESHARDED) && c->nranks > 1) {
            // a rank's share alone would put different blocks into the border on different ranks: the rule is applied to the counts of the WHOLE
            // problem -- couplings to eliminated blocks summed over ranks (every rank eliminates its own), couplings among the reduced blocks as the
            // maximum (every rank sees a subset of the same blocks)
            if (!c->reduce_fn) return fail(c, NLLS_ERR_INVALID_ARG, "NLLS_FLAG_PRESHARDED: install the all-reduce (nlls_comm_init_rccl / nlls_set_allreduce) before nlls_upload_structure -- the reduced system's layout is agreed on collectively");
            std::vector<double> hs(1 + (size_t)nR), hm((size_t)nR);
            { size_t r = 0; hs[0] = (double)c->nelim; for (int64_t k = 0; k < nb; ++k) if (!c->is_elim[k]) { hs[1 + r] = (double)cnt[k]; hm[r] = (double)rcnt[k]; ++r; } }
            DevBuf<double> ds, dm;
            HIPCHK(ds.upload(hs)); HIPCHK(dm.upload(hm));
            { int rc = comm_reduce(c, ds.p, (int64_t)hs.size(), NLLS_REDUCE_SUM); if (rc == NLLS_OK && nR > 0) rc = comm_reduce(c, dm.p, (int64_t)hm.size(), NLLS_REDUCE_MAX); if (rc != NLLS_OK) return rc; }
            HIPCHK(hipStreamSynchronize(c->stream));
            HIPCHK(hipMemcpy(hs.data(), ds.p, sizeof(double) * hs.size(), hipMemcpyDeviceToHost)); if (nR > 0) HIPCHK(hipMemcpy(hm.data(), dm.p, sizeof(double) * hm.size(), hipMemcpyDeviceToHost));
            nelim_all = (int64_t)hs[0]; c->nelim_all = nelim_all;   // (what decides anything collective below: a rank whose share holds no eliminated block must still enter its peers' collectives)
            { size_t r = 0; for (int64_t k = 0; k < nb; ++k) if (!c->is_elim[k]) { cnt[k] = (int64_t)hs[1 + r]; rcnt[k] = (int64_t)hm[r]; ++r; } }
        }
        int64_t bd = 0;
        for (int64_t k = 0; k < nb; ++k) if (!c->is_elim[k]) {
            bool big = (nelim_all >= 64 && cnt[k] * 4 > nelim_all) || (nR >= 64 && rcnt[k] * 4 > nR);
            if (big && bd + c->blocksizes[k] <= 15) { is_border[k] = 1; bd += c->blocksizes[k]; } }
        // ---- order of the banded part: the caller's block order, or reverse Cuthill-McKee where that gives the narrower band ----------
        std::vector<int64_t> band_blocks;                    // reduced, non-border blocks in the order they take in S
        for (int64_t k = 0; k < nb; ++k) if (!c->is_elim[k] && !is_border[k]) band_blocks.push_back(k);
        c->red_reordered = 0; c->bw_caller = -1;
        if (I0.is_sparse && c->nelim_all > 0 && band_blocks.size() >= 3 && !(flags & NLLS_FLAG_NO_REORDER) && band_blocks.size() < ((size_t)1 << 30)) {
            const int32_t nRb = (int32_t)band_blocks.size();
            std::vector<int32_t> rid(nb, -1); for (int32_t i = 0; i < nRb; ++i) rid[band_blocks[i]] = i;
            std::vector<std::vector<int32_t>> adj(nRb); std::vector<uint32_t> cap(nRb, 64);
            auto add = [&](int32_t a, int32_t b2) { auto& l = adj[a]; l.push_back(b2);
                if (l.size() > cap[a]) { std::sort(l.begin(), l.end()); l.erase(std::unique(l.begin(), l.end()), l.end()); cap[a] = (uint32_t)std::max<size_t>(64, 2 * l.size()); } };
            std::vector<int32_t> nl, prevnl;
            for (int64_t v = 0; v < nb; ++v) if (c->is_elim[v]) {        // an eliminated block couples all its reduced neighbours pairwise
                nl.clear();
                for (int64_t q = c->it_colptr[v]; q < c->it_colptr[v + 1]; ++q) { const int64_t u = c->it_rowval[q]; if (u != v && rid[u] >= 0) nl.push_back(rid[u]); }
                for (int64_t q = tptr[v]; q < tptr[v + 1]; ++q) { const int64_t w = trow[q]; if (w != v && rid[w] >= 0) nl.push_back(rid[w]); }
                std::sort(nl.begin(), nl.end());
                if (nl == prevnl) continue;                                 // (supernodes: the same clique again)
                for (size_t a = 0; a < nl.size(); ++a) for (size_t b2 = 0; b2 < a; ++b2) { add(nl[a], nl[b2]); add(nl[b2], nl[a]); }
                prevnl = nl;
            }
            for (int64_t row = 0; row < nb; ++row) { if (rid[row] < 0) continue;       // stored reduced-reduced blocks
                for (int64_t q = c->it_colptr[row]; q < c->it_colptr[row + 1]; ++q) { const int64_t col = c->it_rowval[q]; if (col != row && rid[col] >= 0) { add(rid[row], rid[col]); add(rid[col], rid[row]); } } }
            bool graph_ok = true;
            if ((flags & NLLS_FLAG_PRESHARDED) && c->nranks > 1) {
                // every rank sees the couplings of ITS eliminated blocks only: the graph is the union over ranks.  The edges are GATHERED with the one primitive there is
                // (an all-reduce): the ranks' edge counts first (a sum over a [nranks] vector, each rank writing its own slot), then every rank writes its edges -- one double
                // per edge, a * nRb + b: exact up to 2^53 -- into its own segment of a buffer of the total length, zeros elsewhere; the sum is the concatenation.
                std::vector<double> hc((size_t)c->nranks, 0.0);
                { size_t ne = 0; for (int32_t a = 0; a < nRb; ++a) for (int32_t b2 : adj[a]) ne += b2 < a; hc[c->rank] = (double)ne; }
                DevBuf<double> dc; HIPCHK(dc.upload(hc));
                { const int rc = comm_reduce(c, dc.p, (int64_t)hc.size(), NLLS_REDUCE_SUM); if (rc != NLLS_OK) return rc; }
                HIPCHK(hipStreamSynchronize(c->stream)); HIPCHK(hipMemcpy(hc.data(), dc.p, sizeof(double) * hc.size(), hipMemcpyDeviceToHost));
                size_t total = 0, my0 = 0; for (int r = 0; r < c->nranks; ++r) { if (r == c->rank) my0 = total; total += (size_t)hc[r]; }
                if ((double)nRb * (double)nRb >= 9.0e15 || total > ((size_t)1 << 31)) graph_ok = false;       // (every rank decides alike: the counts are the same everywhere)
                else if (total > 0) {
                    std::vector<double> he(total, 0.0);
                    { size_t q = my0; for (int32_t a = 0; a < nRb; ++a) for (int32_t b2 : adj[a]) if (b2 < a) he[q++] = (double)a * (double)nRb + (double)b2; }
                    DevBuf<double> de; HIPCHK(de.upload(he));
                    { const int rc = comm_reduce(c, de.p, (int64_t)total, NLLS_REDUCE_SUM); if (rc != NLLS_OK) return rc; }
                    HIPCHK(hipStreamSynchronize(c->stream)); HIPCHK(hipMemcpy(he.data(), de.p, sizeof(double) * total, hipMemcpyDeviceToHost));
                    for (auto& l : adj) l.clear();
                    for (double e : he) { const int64_t k = (int64_t)e; const int32_t a = (int32_t)(k / nRb), b2 = (int32_t)(k % nRb); if (a > b2 && a < nRb) { adj[a].push_back(b2); adj[b2].push_back(a); } }
                }
            }
            if (graph_ok) {
                for (auto& l : adj) { std::sort(l.begin(), l.end()); l.erase(std::unique(l.begin(), l.end()), l.end()); }
                // half bandwidth (dof) of an order: the farthest coupled pair, first dof of the one to the last dof of the other
                auto bandwidth_of = [&](const std::vector<int32_t>& perm) { std::vector<int64_t> o(nRb + 1, 0), pos(nRb);
                    for (int32_t i = 0; i < nRb; ++i) { pos[perm[i]] = i; o[i + 1] = o[i] + c->blocksizes[band_blocks[perm[i]]]; }
                    int64_t w = 0; for (int32_t a = 0; a < nRb; ++a) { w = std::max<int64_t>(w, c->blocksizes[band_blocks[a]] - 1);
                        for (int32_t b2 : adj[a]) { const int64_t pa = pos[a], pb = pos[b2]; w = std::max(w, std::max(o[pa + 1], o[pb + 1]) - std::min(o[pa], o[pb]) - 1); } }
                    return w; };
                std::vector<int32_t> ident(nRb); std::iota(ident.begin(), ident.end(), 0);
                const int64_t bw_nat = bandwidth_of(ident);
                c->bw_caller = bw_nat;
                const std::vector<int32_t> perm = rcm_order(adj);
                red_adj_blocks = band_blocks;
                if ((int32_t)perm.size() == nRb && bandwidth_of(perm) < bw_nat) {
                    std::vector<int64_t> nbk(nRb); for (int32_t i = 0; i < nRb; ++i) nbk[i] = band_blocks[perm[i]];
                    band_blocks.swap(nbk); c->red_reordered = 1; }
                red_adj.swap(adj);
            }
        }
        int64_t ro = 0;
        for (int64_t k : band_blocks) { red_of[k] = ro; ro += c->blocksizes[k]; }
        c->n_band = ro;
        for (int64_t k = 0; k < nb; ++k) if (!c->is_elim[k] && is_border[k]) { red_of[k] = ro; ro += c->blocksizes[k]; }
        c->nred = ro; c->nbd = (int)(ro - c->n_band);
    }
    int64_t bw = 0;   // half bandwidth (in dof) of the non-border part of S
    // bt_bad[NT]: blocks of 16 NT unknowns do NOT make the band part of S block TRIDIAGONAL (some coupled pair of unknowns lies two blocks apart).  Blocks of at least bw
    // unknowns always do; a smaller block may as well -- cameras of 6 unknowns that share points with ten neighbours: bw = 65, and no coupling crosses two blocks of 64 --
    // and the block cyclic reduction's dependent chain is (tiles per block) x (levels): it takes the cheapest block size the STRUCTURE allows (choose_bcr_nt below)
    double bt_bad[6] = {0, 0, 0, 0, 0, 0};
    auto bt_note = [&](int64_t lo, int64_t hi) { for (int t = 1; t <= 5; ++t) if (hi / (16 * t) - lo / (16 * t) > 1) bt_bad[t] = 1.0; };
    if (c->nranks > 1) for (int64_t v = 0; v < nb; ++v) if (c->is_elim[v]) {   // all ranks must agree on the layout of S
        int64_t lo = -1, hi = -1;
        auto upd = [&](int64_t u) { const int64_t r0 = red_of[u]; if (r0 < c->n_band) { lo = lo < 0 ? r0 : std::min(lo, r0); hi = std::max(hi, r0 + c->blocksizes[u] - 1); } };
        for (int64_t q = c->it_colptr[v]; q < c->it_colptr[v + 1]; ++q) if (c->it_rowval[q] != v) upd(c->it_rowval[q]);
        for (int64_t q = tptr[v]; q < tptr[v + 1]; ++q) if (trow[q] != v) upd(trow[q]);
        if (lo >= 0) { bw = std::max(bw, hi - lo); bt_note(lo, hi); }
    }
    std::vector<int64_t> eptr; std::vector<SchurNbr> enbr; std::vector<int64_t> ediag; std::vector<uint32_t> eboff; std::vector<uint16_t> edim; std::vector<uint32_t> egroup;
    std::vector<int64_t> erow;                 // block row of each (local) eliminated member
    std::vector<uint32_t> fastg_all;           // fast supernodes in launch order (copy kept for the gather index)
    std::vector<uint8_t> row_fast(nb, 0);      // block rows whose members take the fast elimination path
    eptr.push_back(0);
    if (c->nelim) {
        std::vector<SchurNbr> prev; uint32_t glen = 0;
        constexpr uint32_t sn_cap = 128;            // members per supernode
        for (int64_t v = 0; v < nb; ++v) if (c->is_elim[v] && (c->nranks == 1 || c->owner_of_block[v] == c->rank)) {
            std::vector<SchurNbr> nl;
            for (int64_t q = c->it_colptr[v]; q < c->it_colptr[v + 1]; ++q) { int64_t u = c->it_rowval[q]; if (u == v) continue;
                nl.push_back(SchurNbr{c->it_nzval[q], (uint32_t)red_of[u], (uint16_t)c->blocksizes[u], 0}); }
            for (int64_t q = tptr[v]; q < tptr[v + 1]; ++q) { int64_t w = trow[q]; if (w == v) continue;
                nl.push_back(SchurNbr{c->it_nzval[tq[q]], (uint32_t)red_of[w], (uint16_t)c->blocksizes[w], 1}); }
            // column order of [E_v]: the order the blocks lie in memory when the whole row is stored with the member (the fast kernels
            // step through it with a constant stride; a border block -- ordered LAST in the reduced system -- may come FIRST here, so
            // nothing downstream may assume ascending reduced columns: every S(i, j) is addressed as (max, min)); else by reduced column
            { bool anytrans = false; for (auto& n : nl) anytrans |= n.trans != 0;
              if (!anytrans) std::sort(nl.begin(), nl.end(), [](const SchurNbr& a, const SchurNbr& b) { return a.off < b.off; });
              else std::sort(nl.begin(), nl.end(), [](const SchurNbr& a, const SchurNbr& b) { return a.rcol < b.rcol; }); }
            int nd = 0; int64_t lo = -1, hi = -1;
            for (auto& n : nl) { nd += n.dim; if ((int64_t)n.rcol < c->n_band) { lo = lo < 0 ? (int64_t)n.rcol : std::min<int64_t>(lo, n.rcol); hi = std::max<int64_t>(hi, (int64_t)n.rcol + n.dim - 1); } }   // (any order: the columns are in MEMORY order)
            if (lo >= 0) { bw = std::max(bw, hi - lo); bt_note(lo, hi); }
            c->max_nbr_dof = std::max(c->max_nbr_dof, nd); c->max_elim_dim = std::max(c->max_elim_dim, (int)c->blocksizes[v]);
            // supernode: same neighbour columns and own size as the previous eliminated block
            bool same = !edim.empty() && glen < sn_cap && prev.size() == nl.size() && edim.back() == (uint16_t)c->blocksizes[v];
            if (same) for (size_t i = 0; i < prev.size(); ++i) if (prev[i].rcol != nl[i].rcol || prev[i].dim != nl[i].dim) { same = false; break; }
            // ... and its block row follows the previous member's directly in A.data and b: the kernels then step through a
            // supernode with a constant stride instead of looking every member's offsets up (a dependent load per member)
            if (same) { const int64_t dv = c->blocksizes[v];
                if (c->diag_off[v] != ediag.back() + dv * dv + dv * nd || (int64_t)c->boffsets[v] != (int64_t)eboff.back() + dv) same = false; }
            if (!same) { egroup.push_back((uint32_t)ediag.size()); glen = 0; }
            ++glen; prev = nl;
            for (auto& n : nl) enbr.push_back(n);
            erow.push_back(v); eptr.push_back((int64_t)enbr.size()); ediag.push_back(c->diag_off[v]); eboff.push_back((uint32_t)c->boffsets[v]); edim.push_back((uint16_t)c->blocksizes[v]);
        }
        egroup.push_back((uint32_t)ediag.size());
        // Few eliminated blocks (BASELINE config 3: 10k points in ~100 runs of 99): one workgroup per run would leave most of the chip idle through the assembly and the
        // back-substitution.  Runs are cut into balanced pieces of about nelim / 384 members, never below 24 (every piece pays its own flush of the packed image: 1830 atomics at
        // ten cameras) -- measured at config 3: 5903 (99 members per workgroup), 6346 (50), 6478 (25), 6330 (16), 5706 (8) LM iterations/s; configs 4 and 5 (991 / 500 runs)
        // lose with ANY cut (2848 -> 2667 / 2505 -> 2332 at 64) and are not cut.  NLLS_SUPERNODE_PIECE=n: the piece size by hand (A/B).
        { static const int piece_env = [] { const char* e = getenv("NLLS_SUPERNODE_PIECE"); return e ? atoi(e) : 0; }();
          const int64_t total = (int64_t)ediag.size();
          const int64_t piece = piece_env > 0 ? piece_env : std::min<int64_t>(128, std::max<int64_t>(24, (total + 383) / 384));
          if (piece < 128 && c->nranks == 1) {
              std::vector<uint32_t> cut; cut.reserve(egroup.size() * 2);
              for (size_t gi = 0; gi + 1 < egroup.size(); ++gi) {
                  const uint32_t a0 = egroup[gi], m = egroup[gi + 1] - a0;
                  const uint32_t np = (uint32_t)std::max<int64_t>(1, (2 * (int64_t)m + piece) / (2 * piece));        // round(m / piece)
                  for (uint32_t k = 0; k < np; ++k) cut.push_back(a0 + (uint32_t)(((uint64_t)m * k) / np));
              }
              cut.push_back(egroup.back()); egroup.swap(cut);
          } }
        c->nelim_groups = (int64_t)egroup.size() - 1;
        // fast-path eligibility: small compile-time block size, few neighbour dof, and the member's off-diagonal
        // blocks stored back to back right before its diagonal block, in reduced-column order
        std::vector<uint32_t> fastg60, fastg, fastw, slowg; int fast_dv = 0; int fast_maxk = 0;
        { std::unordered_map<int, int64_t> dvcount; for (auto d : edim) dvcount[d]++;
          int64_t bestc = 0; for (auto& kv : dvcount) if (kv.first <= 3 && kv.second > bestc) { bestc = kv.second; fast_dv = kv.first; } }
        for (size_t gi = 0; gi + 1 < egroup.size(); ++gi) {
            bool ok = fast_dv > 0;
            int ndg = 0;
            for (uint32_t v = egroup[gi]; ok && v < egroup[gi + 1]; ++v) {
                if (edim[v] != fast_dv) { ok = false; break; }
                int nd = 0; int64_t expect = -1;
                for (int64_t p = eptr[v]; p < eptr[v + 1]; ++p) { const SchurNbr& n = enbr[p];
                    if (n.trans) { ok = false; break; }
                    if (expect >= 0 && n.off != expect) { ok = false; break; }
                    expect = n.off + (int64_t)fast_dv * n.dim; nd += n.dim; }
                if (ok && eptr[v + 1] > eptr[v] && expect != ediag[v]) ok = false;
                if (nd + 1 > 71) ok = false;
                ndg = nd;
            }
            if (ok) { (ndg <= 60 ? fastg60 : ndg + 1 <= 64 ? fastg : fastw).push_back((uint32_t)gi); if (ndg + 1 <= 64) fast_maxk = std::max(fast_maxk, (ndg * (ndg + 1) / 2 + 63) / 64); } else slowg.push_back((uint32_t)gi);
        }
        c->n_fast_n60 = (int64_t)fastg60.size();
        fastg.insert(fastg.begin(), fastg60.begin(), fastg60.end());
        c->n_fast_narrow = (int64_t)fastg.size(); c->fast_maxk_narrow = fast_maxk;
        fastg.insert(fastg.end(), fastw.begin(), fastw.end());
        // the generic kernel stages [C | E | Y] of a member (and, where they fit, the supernode's pair accumulators) in LDS: a supernode is limited
        // by ITS OWN width -- a point seen by 40 cameras costs its own workgroup more LDS and more atomics, not the whole problem its Schur
        // path -- and by the 160 KB of a gfx950 CU (150 KB budget: 328 six-dof neighbours of a three-dof block)
        constexpr size_t ELIM_LDS_BUDGET = 150 * 1024;
        auto lds_base = [](size_t nd, size_t dv) { return sizeof(double) * (dv * dv + dv * nd + dv * (nd + 1)) + 16 * nd + 12 * nd + 16; };
        auto lds_acc = [](size_t nd) { return sizeof(double) * (nd * (nd + 1) / 2 + nd); };
        { std::vector<uint32_t> sacc, snoacc; c->slow_nd_acc = c->slow_nd_noacc = 0;
          const size_t dvm = (size_t)c->max_elim_dim;
          for (uint32_t gi : slowg) { int nd = 0; for (int64_t p = eptr[egroup[gi]]; p < eptr[egroup[gi] + 1]; ++p) nd += enbr[p].dim;
              if (lds_base((size_t)nd, dvm) > ELIM_LDS_BUDGET) { c->err_sub = NLLS_SUB_SCHUR_SHAPE; return fail(c, NLLS_ERR_UNSUPPORTED, "eliminated block with too many neighbours for the LDS-staged Schur kernel (retry with NLLS_FLAG_NO_SCHUR)"); }
              if (lds_base((size_t)nd, dvm) + lds_acc((size_t)nd) <= ELIM_LDS_BUDGET) { sacc.push_back(gi); c->slow_nd_acc = std::max(c->slow_nd_acc, nd); }
              else { snoacc.push_back(gi); c->slow_nd_noacc = std::max(c->slow_nd_noacc, nd); } }
          c->n_slow_acc = (int64_t)sacc.size();
          slowg = sacc; slowg.insert(slowg.end(), snoacc.begin(), snoacc.end());
          c->elim_lds_acc = lds_base((size_t)c->slow_nd_acc, dvm) + lds_acc((size_t)c->slow_nd_acc); c->elim_lds_noacc = lds_base((size_t)c->slow_nd_noacc, dvm); }
        c->n_fast_groups = (int64_t)fastg.size(); c->n_slow_groups = (int64_t)slowg.size(); c->fast_dv = fast_dv; c->fast_maxk = fast_maxk;
        fastg_all = fastg;
        std::vector<uint32_t> slowb; for (uint32_t gi : slowg) for (uint32_t v = egroup[gi]; v < egroup[gi + 1]; ++v) slowb.push_back(v);
        std::vector<uint32_t> fastb; for (uint32_t gi : fastg) for (uint32_t v = egroup[gi]; v < egroup[gi + 1]; ++v) { fastb.push_back(v); row_fast[erow[v]] = 1; }
        c->n_fast_members = (int64_t)fastb.size();
        if (hipSuccess != c->d_fast_members.upload(fastb) || hipSuccess != c->tE.alloc(std::max<size_t>(1, ediag.size() * (size_t)std::max(1, fast_dv)))) return fail(c, NLLS_ERR_HIP, "fast member upload");
        { std::vector<ElimDesc> desc; std::vector<uint32_t> rcflat;
          for (uint32_t gi : fastg) { const uint32_t v0 = egroup[gi];
              ElimDesc d{v0, egroup[gi + 1] - v0, 0, (uint32_t)rcflat.size(), ediag[v0], eboff[v0], 0};
              for (int64_t p = eptr[v0]; p < eptr[v0 + 1]; ++p) { for (int q = 0; q < enbr[p].dim; ++q) rcflat.push_back(enbr[p].rcol + q); d.nd += enbr[p].dim; }
              // ... followed by the list columns sorted by reduced column (the flush walks S column by column whatever order the columns have in memory)
              { std::vector<uint32_t> idx(d.nd); std::iota(idx.begin(), idx.end(), 0u); const uint32_t* r0 = rcflat.data() + d.rc_off;
                std::sort(idx.begin(), idx.end(), [r0](uint32_t a, uint32_t b2) { return r0[a] < r0[b2]; });
                rcflat.insert(rcflat.end(), idx.begin(), idx.end()); }
              desc.push_back(d); }
          if (rcflat.empty()) rcflat.push_back(0);
          // (rounds 3-4 could fold the tiny supernodes at every step of the visibility window into a large neighbour -- ElimPre, NLLS_ELIM_FOLD: parity-green, 316 against 294 us
          //  per solve: the tiny workgroups were filling slots the large ones leave idle.  Out of the library since round 5; last in the tree at commit 6e015b8.)
          if (hipSuccess != c->d_elim_desc.upload(desc) || hipSuccess != c->d_elim_rc.upload(rcflat)) return fail(c, NLLS_ERR_HIP, "supernode descriptor upload");
          // (host copies for build_mf: the matrix-free trial's block list follows the supernodes' launch order)
          c->h_elim_desc = desc; c->h_erow = erow; c->h_eptr = eptr;
          { std::vector<int64_t> blk_of_red((size_t)c->nred, -1); for (int64_t k = 0; k < nb; ++k) if (!c->is_elim[k] && red_of[k] >= 0) blk_of_red[(size_t)red_of[k]] = k;
            c->h_enbr_block.resize(enbr.size()); for (size_t i = 0; i < enbr.size(); ++i) c->h_enbr_block[i] = enbr[i].rcol < blk_of_red.size() ? blk_of_red[enbr[i].rcol] : -1; } }
        if (hipSuccess != c->d_fast_groups.upload(fastg) || hipSuccess != c->d_slow_groups.upload(slowg) || hipSuccess != c->d_slow_blocks.upload(slowb) ||
            hipSuccess != c->Cinv.alloc(std::max<size_t>(1, ediag.size() * (size_t)std::max(1, fast_dv * fast_dv)))) return fail(c, NLLS_ERR_HIP, "group list upload");
        if (hipSuccess != c->d_elim_ptr.upload(eptr) || hipSuccess != c->d_elim_nbr.upload(enbr) || hipSuccess != c->d_elim_diag.upload(ediag) ||
            hipSuccess != c->d_elim_boff.upload(eboff) || hipSuccess != c->d_elim_dim.upload(edim) || hipSuccess != c->d_elim_group.upload(egroup)) return fail(c, NLLS_ERR_HIP, "schur upload");
    }
    // reduced-reduced blocks to copy into S
    std::vector<SchurCopy> copies; std::vector<uint32_t> red_boff(c->nred);
    for (int64_t k = 0; k < nb; ++k) if (!c->is_elim[k]) for (int i = 0; i < c->blocksizes[k]; ++i) red_boff[red_of[k] + i] = (uint32_t)(c->boffsets[k] + i);
    if (I0.is_sparse) for (int64_t row = 0; row < nb; ++row) { if (c->is_elim[row]) continue;
        for (int64_t q = c->it_colptr[row]; q < c->it_colptr[row + 1]; ++q) { int64_t col = c->it_rowval[q]; if (c->is_elim[col]) continue;
            // S is addressed by its lower triangle in REDUCED order: transpose the block if the border reordering flipped it
            SchurCopy cp{c->it_nzval[q], (uint32_t)red_of[row], (uint32_t)red_of[col], (uint16_t)c->blocksizes[row], (uint16_t)c->blocksizes[col]};
            copies.push_back(cp);
            if (red_of[row] < c->n_band && red_of[col] < c->n_band && row != col) { bw = std::max<int64_t>(bw, std::llabs(red_of[row] - red_of[col]) + std::max(c->blocksizes[row], c->blocksizes[col]) - 1);
                bt_note(std::min(red_of[row], red_of[col]), std::max(red_of[row] + c->blocksizes[row], red_of[col] + c->blocksizes[col]) - 1); }
            if (row == col) { bw = std::max<int64_t>(bw, c->blocksizes[row] - 1); if (red_of[row] < c->n_band) bt_note(red_of[row], red_of[row] + c->blocksizes[row] - 1); }
        } }
    c->ncopy = (int64_t)copies.size();
    if ((flags & NLLS_FLAG_PRESHARDED) && c->nranks > 1) {
        // every rank has built the reduced system from ITS cost blocks only: the layout of [S | s] -- summed element by element over ranks --
        // must be one layout.  The bandwidth is the maximum over ranks; the reduced order itself (banded part, border, size) has to agree.
        if (!c->reduce_fn) return fail(c, NLLS_ERR_INVALID_ARG, "NLLS_FLAG_PRESHARDED: install the all-reduce (nlls_comm_init_rccl / nlls_set_allreduce) before nlls_upload_structure -- the reduced system's layout is agreed on collectively");
        // ... and so must everything that is summed element by element: the stage-0 buffer ([cost | reduced rows of A.data | reduced part of b]: its length
        // and the lengths of its segments -- they differ when a rank's reduced rows hold E blocks another rank's do not, i.e. when an eliminated block
        // precedes a reduced one in block order) and the list of reduced-reduced blocks.  +x / -x pairs under MAX: equal on all ranks or refused.
        double seghash = 0.0;      // order-sensitive checksum of the reduced rows' segment lengths and of the reduced order itself (exact in a double)
        { uint64_t hsh = 1469598103934665603ull; auto mix = [&](uint64_t v) { hsh ^= v; hsh *= 1099511628211ull; };
          for (int64_t k = 0; k < nb; ++k) if (!c->is_elim[k]) { mix((uint64_t)red_of[k]); mix((uint64_t)c->blocksizes[k]);
              int64_t len = 0; if (I0.is_sparse) for (int64_t q = c->it_colptr[k]; q < c->it_colptr[k + 1]; ++q) len += (int64_t)c->blocksizes[k] * c->blocksizes[c->it_rowval[q]];
              mix((uint64_t)len); }
          seghash = (double)(hsh >> 12); }
        double h[20] = {(double)bw, (double)c->nbd, (double)c->n_band, (double)c->nred, -(double)c->nbd, -(double)c->n_band, -(double)c->nred, 0.0,
                        (double)c->redbuf_len, -(double)c->redbuf_len, (double)c->ncopy, -(double)c->ncopy, seghash, -seghash, 0.0, 0.0, bt_bad[1], bt_bad[2], bt_bad[3], bt_bad[4]};   // (bt_bad: a block size one rank's couplings rule out is ruled out)
        DevBuf<double> dh; HIPCHK(dh.alloc(20));
        HIPCHK(hipMemcpyAsync(dh.p, h, sizeof h, hipMemcpyHostToDevice, c->stream));
        { const int rc = comm_reduce(c, dh.p, 20, NLLS_REDUCE_MAX); if (rc != NLLS_OK) return rc; }
        HIPCHK(hipMemcpyAsync(h, dh.p, sizeof h, hipMemcpyDeviceToHost, c->stream)); HIPCHK(hipStreamSynchronize(c->stream));
        if (h[1] != -h[4] || h[2] != -h[5] || h[3] != -h[6]) return fail(c, NLLS_ERR_INVALID_ARG, "NLLS_FLAG_PRESHARDED: the ranks' reduced systems differ in size or border (the reduced variables must be the same on every rank)");
        if (h[8] != -h[9] || h[10] != -h[11] || h[12] != -h[13]) return fail(c, NLLS_ERR_INVALID_ARG, "NLLS_FLAG_PRESHARDED: the ranks' reduced rows differ in layout (stage-0 buffer length, reduced-reduced block list or segment lengths): "
                                                                                      "every rank must hold the same reduced variables in the same order, each coupled to the same stored blocks -- list the reduced variables before the eliminated ones");
        bw = (int64_t)h[0]; for (int t = 1; t <= 4; ++t) bt_bad[t] = h[15 + t];
    }
    { std::vector<SchurCopy> blks;
      if (I0.is_sparse) for (int64_t row = 0; row < nb; ++row) for (int64_t q = c->it_colptr[row]; q < c->it_colptr[row + 1]; ++q) { int64_t col = c->it_rowval[q];
          blks.push_back(SchurCopy{c->it_nzval[q], (uint32_t)c->boffsets[row], (uint32_t)c->boffsets[col], (uint16_t)c->blocksizes[row], (uint16_t)c->blocksizes[col]}); }
      c->nblk = (int64_t)blks.size(); if (hipSuccess != c->d_blk.upload(blks)) return fail(c, NLLS_ERR_HIP, "block list upload");
      // ownership masks for global reductions under sharding: eliminated rows count on their owner, reduced rows on rank 0
      std::vector<uint8_t> rowmask(nb, 1), blkmask; std::vector<double> dofmask(I0.ndof, 1.0);
      if (c->nranks > 1) {
          for (int64_t r = 0; r < nb; ++r) { const bool own = c->is_elim[r] ? c->owner_of_block[r] == c->rank : c->rank == 0; rowmask[r] = own;
              for (int i = 0; i < c->blocksizes[r]; ++i) dofmask[c->boffsets[r] + i] = own ? 1.0 : 0.0; }
          for (int64_t row = 0; row < nb; ++row) for (int64_t q = c->it_colptr[row]; q < c->it_colptr[row + 1]; ++q) blkmask.push_back(rowmask[row]);
      }
      // blocks that quadform_points_kernel covers (rows of fast-path members) are skipped by the block kernel when E x is at hand
      std::vector<uint8_t> slowmask;
      if (I0.is_sparse && c->n_fast_members > 0) for (int64_t row = 0; row < nb; ++row) for (int64_t q = c->it_colptr[row]; q < c->it_colptr[row + 1]; ++q) slowmask.push_back((uint8_t)(rowmask[row] && !row_fast[row]));
      // ... as a compact list (bundle adjustment: the camera blocks, ~1e3 of ~1e6), so that the kernel does not walk a mask over all blocks
      std::vector<SchurCopy> slowblks;
      for (size_t q = 0; q < slowmask.size(); ++q) if (slowmask[q]) slowblks.push_back(blks[q]);
      c->nblk_slow = (int64_t)slowblks.size();
      if (c->nranks > 1) {      // lazy stage 0: the reduced rows hold this rank's share only -- they count on EVERY rank
          std::vector<uint8_t> blkmask2; std::vector<double> dofmask2(dofmask); std::vector<SchurCopy> slow2; size_t q = 0;
          for (int64_t row = 0; row < nb; ++row) { const bool own2 = c->is_elim[row] ? rowmask[row] != 0 : true;
              if (!c->is_elim[row]) for (int i = 0; i < c->blocksizes[row]; ++i) dofmask2[c->boffsets[row] + i] = 1.0;
              for (int64_t qq = c->it_colptr[row]; qq < c->it_colptr[row + 1]; ++qq, ++q) { blkmask2.push_back(own2);
                  if (!slowmask.empty() && own2 && !row_fast[row]) slow2.push_back(blks[q]); } }
          c->nblk_slow_lazy = (int64_t)slow2.size();
          if (hipSuccess != c->d_blk_mask_lazy.upload(blkmask2) || hipSuccess != c->d_dof_mask_lazy.upload(dofmask2) || hipSuccess != c->d_blk_slow_lazy.upload(slow2)) return fail(c, NLLS_ERR_HIP, "mask upload");
      }
      if (hipSuccess != c->d_row_mask.upload(rowmask) || hipSuccess != c->d_blk_mask.upload(blkmask) || hipSuccess != c->d_dof_mask.upload(dofmask) ||
          hipSuccess != c->d_blk_slowmask.upload(slowmask) || hipSuccess != c->d_blk_slow.upload(slowblks)) return fail(c, NLLS_ERR_HIP, "mask upload"); }
    if (hipSuccess != c->d_copy.upload(copies) || hipSuccess != c->d_red_boff.upload(red_boff)) return fail(c, NLLS_ERR_HIP, "schur upload");
    {   // the retraction of an LM trial inside the back-substitution launch: eliminated member -> where its variable is stored; every other variable -> where its step
        // starts in the reduced solution (fixed variables: -1, they are copied)
        const int64_t nvar = (int64_t)c->var_kind.size();
        std::vector<int64_t> var_of_block(nb, -1); for (int64_t i = 0; i < nvar; ++i) if (c->blockindices[i]) var_of_block[c->blockindices[i] - 1] = i;
        std::vector<uint32_t> fast_var(erow.size(), 0), rest_var; std::vector<int32_t> rest_red; std::vector<uint8_t> is_member(nvar, 0);
        bool euclid = c->n_fast_members > 0 && c->n_fast_members == (int64_t)erow.size();
        for (size_t v = 0; v < erow.size(); ++v) { const int64_t i = var_of_block[erow[v]]; if (i < 0) { euclid = false; continue; }
            fast_var[v] = c->var_off[i]; is_member[i] = 1;
            if (c->var_kind[i] != NLLS_VAR_EUCLIDEAN || c->var_dim[i] != c->fast_dv || c->blocksizes[erow[v]] != c->fast_dv) euclid = false; }
        for (int64_t i = 0; i < nvar; ++i) { if (is_member[i]) continue;
            const int64_t k = (int64_t)c->blockindices[i] - 1; rest_var.push_back((uint32_t)i);
            rest_red.push_back(k < 0 ? -1 : (c->is_elim[k] ? -2 : (int32_t)red_of[k])); if (k >= 0 && c->is_elim[k]) euclid = false; }
        c->fast_all_euclid = euclid && c->nranks == 1;
        if (hipSuccess != c->d_fast_voff.upload(fast_var) || hipSuccess != c->d_rest_var.upload(rest_var) || hipSuccess != c->d_rest_red.upload(rest_red)) return fail(c, NLLS_ERR_HIP, "schur upload");
    }
    // ---- choose the reduced-system solver ----------------------------------------------------------------------
    const int64_t n = c->nred;
    c->bw = (int)bw; c->solve_mode = SOLVE_DENSE; c->band_twisted = !(flags & NLLS_FLAG_NO_TWIST); c->dense_window = false; c->dense_pad128 = false;
    if (n < 64) c->solve_mode = SOLVE_SMALL;
    else if (!I0.is_sparse) c->solve_mode = SOLVE_DENSE;
    else if (c->n_band >= 128 && !(flags & NLLS_FLAG_NO_BAND)) {
        // bordered-band LDL' in one persistent workgroup: needs the LDS ring + prefetch registers to fit
        const int H = (int)bw + 1 + c->nbd + 1;
        // register window: (bw+1) columns x ceil((bw+1)/SEG) segments, NSLOT segments per lane of a 256-lane workgroup
        int SEG = 0, NSEG = 0;
        for (auto cfg : {std::pair<int, int>{8, 1}, {10, 2}, {12, 2}, {12, 4}, {16, 4}}) {
            const int64_t nsc = (bw + 1 + cfg.first - 1) / cfg.first;
            if ((bw + 1) * nsc <= 256 * cfg.second) { SEG = cfg.first; NSEG = cfg.second; break; }
        }
        if (SEG && bw <= 127) for (int CH : {32, 16, 8}) {
            const int PFC = ((int)bw + 1 + CH - 1) / CH + 1, RC = (PFC + 1) * CH;
            const int nbr = c->nbd + 1, NSC = ((int)bw + 1 + SEG - 1) / SEG;
            const size_t lds = sizeof(double) * ((size_t)RC * H + 2 * (size_t)(2 * NSC * SEG + 2 * SEG) + (size_t)(bw + 2) * nbr + (size_t)nbr * nbr + nbr + 8);
            if (H <= 255 && bw >= 1 && lds <= 150 * 1024 && CH * H <= 12 * 256) {
                c->solve_mode = SOLVE_BAND; c->band_CH = CH; c->band_H = H; c->band_SEG = SEG; c->band_NSEG = NSEG; break; }
        }
    }
    if (c->solve_mode == SOLVE_BAND) {
        const size_t sz = (size_t)c->band_H * c->n_band + (size_t)(c->nbd + 1) * (c->nbd + 1);
        c->s_elems = sz;
        // the blocked factor kernel exports tiles: (NBW + 1) 16x16 tiles + the border/rhs rows per 16-column block
        const size_t nbw = ((size_t)bw + 15) / 16, tsz = (((size_t)c->n_band + 15) / 16) * ((nbw + 1) * 256 + (size_t)(c->nbd + 1) * 16) + 256 + 2 * (size_t)(c->nbd + 1) * (c->nbd + 1) + 2 * (256 * nbw * nbw + 16 * nbw)
                                                    + (16 * nbw) * (16 * nbw + 1) + 8 + (nbw + 1) * ((nbw + 1) * 256 + 16) + 256;   // separator: band system + its factor tiles
        if (hipSuccess != c->S.alloc(sz + (size_t)n + 64) || hipSuccess != c->Lwork.alloc(std::max(sz, tsz)) || hipSuccess != c->d_status.alloc(96)) return fail(c, NLLS_ERR_HIP, "band system alloc");
        // block cyclic reduction (nlls_bcr.hip) is the band solver whenever it supports the shape; the chain kernels stay as fallbacks
        if (!(flags & NLLS_FLAG_NO_BCR) && BcrSolver::supports(c->n_band, (int)bw, c->nbd)) {
            // tiles per block: the cheapest dependent chain, NT x levels (levels = ceil(log2(N + 1)) for N blocks), among the block sizes the structure allows
            // (NLLS_BCR_NT_FULL=1: always ceil(bw / 16), rounds 2-5)
            const int nt_full = std::max(1, ((int)bw + 15) / 16); int nt_best = nt_full;
            if (!getenv("NLLS_BCR_NT_FULL")) {
                auto chain = [&](int t) { const int64_t N = (c->n_band + 16 * t - 1) / (16 * t); int lv = 0; while (((int64_t)1 << lv) < N + 1) ++lv; return (int64_t)t * lv; };
                for (int t = nt_full - 1; t >= 1; --t) if (bt_bad[t] == 0.0 && chain(t) < chain(nt_best)) nt_best = t;
            }
            std::string e; const int rc = c->bcr.build(c->n_band, (int)bw, c->nbd, c->band_H, &e, nt_best);
            if (rc != NLLS_OK) return fail(c, rc, e.c_str());
        }
        // ---- slab + gather assembly (NLLS_FLAG_DETERMINISTIC: no atomics, x is bit-reproducible; 15 % slower than the atomic flush at
        //      config 4): single rank, every eliminated block on the fast path
        // (round 6: the matrix-free LM trial assembles the tiles this way BY DEFAULT -- its supernodes leave their shares in the slabs with plain stores, the memory side took
        //  45 us per trial for the 4.1 M atomics of the flush at BASELINE config 4 -- so the index is built whenever that trial may apply; elim_slab, i.e. the MATERIALISED
        //  elimination through slabs, stays the flag's)
        c->gather_ready = false; c->h_slab_off.clear();
        if (c->bcr.ready && c->nranks == 1 && ((flags & NLLS_FLAG_DETERMINISTIC) || !(flags & NLLS_FLAG_MATERIALIZE)) && c->n_fast_groups > 0 && c->n_slow_groups == 0 && I0.is_sparse) {
            struct Key { uint32_t r, c; bool operator<(const Key& o) const { return r != o.r ? r < o.r : c < o.c; } };
            std::map<Key, std::vector<GatherCon>> pairs; std::map<uint32_t, std::vector<GatherCon>> rhs;
            std::vector<uint32_t> slab_off, slab_groups; uint64_t off = 0; bool ok = (uint64_t)I0.nnz_data < ((uint64_t)1 << 32) && (uint64_t)I0.ndof < ((uint64_t)1 << 32);
            int64_t cls_count[3] = {0, 0, 0};
            // (shares formed INSIDE the gather, member by member, for the one- and two-member supernodes: measured twice -- rounds 4 and 6, notes/r06.md -- and slower; out of the library)
            for (size_t pos = 0; ok && pos < fastg_all.size(); ++pos) {
                const uint32_t gi = fastg_all[pos];
                const uint32_t v0 = egroup[gi], v1 = egroup[gi + 1];
                std::vector<SchurNbr> nl(enbr.begin() + eptr[v0], enbr.begin() + eptr[v0 + 1]);
                if (nl.size() > 16) { ok = false; break; }
                // (the columns of a supernode are in MEMORY order, the reduced order may be any permutation of it: a share whose list pair (a >= b) lies
                //  above the diagonal of S is flagged and read transposed by the gather)
                const int cls = (int64_t)pos < c->n_fast_n60 ? 0 : ((int64_t)pos < c->n_fast_narrow ? 1 : 2);
                cls_count[cls]++; slab_groups.push_back(gi);
                slab_off.push_back((uint32_t)off);
                uint64_t po = off;
                for (size_t a = 0; a < nl.size(); ++a)
                    for (size_t b2 = 0; b2 <= a; ++b2) { const bool tr = nl[a].rcol < nl[b2].rcol;
                        pairs[tr ? Key{nl[b2].rcol, nl[a].rcol} : Key{nl[a].rcol, nl[b2].rcol}].push_back(GatherCon{(uint32_t)po, (uint32_t)nl[a].dim, tr ? 1u : 0u, 0}); po += (uint64_t)nl[a].dim * nl[b2].dim; }
                uint64_t ro = po;
                for (size_t a = 0; a < nl.size(); ++a) { rhs[nl[a].rcol].push_back(GatherCon{(uint32_t)ro, 1, 0, 0}); ro += nl[a].dim; }
                off = (ro + 3) & ~(uint64_t)3;
                if (off >= ((uint64_t)1 << 32)) { ok = false; break; }
            }
            c->n_slab60 = cls_count[0]; c->n_slabnar = cls_count[1]; c->n_slabwide = cls_count[2];
            if (ok) {
                std::vector<GatherJob> jobs; std::vector<GatherCon> cons;
                std::map<Key, const SchurCopy*> base;
                for (const SchurCopy& cp : copies) { if (cp.r >= cp.c) base[Key{cp.r, cp.c}] = &cp; else base[Key{cp.c, cp.r}] = &cp; }
                std::vector<int32_t> dim_of(c->nred + 1, 0);      // size of the reduced block that starts at a reduced dof
                std::vector<uint32_t> boff_of(c->nred + 1, 0);
                for (int64_t k = 0; k < nb; ++k) if (!c->is_elim[k]) { dim_of[red_of[k]] = c->blocksizes[k]; boff_of[red_of[k]] = (uint32_t)c->boffsets[k]; }
                // every block pair that holds a share or a stored block
                std::map<Key, int> all; for (auto& kv : pairs) all[kv.first] = 1; for (auto& kv : base) all[kv.first] = 1;
                for (auto& kv : all) {
                    GatherJob j{}; j.copy_off = -1; j.r0 = kv.first.r; j.c0 = kv.first.c; j.rows = (uint16_t)dim_of[j.r0]; j.cols = (uint16_t)dim_of[j.c0]; j.kind = 0;
                    auto bi = base.find(kv.first);
                    if (bi != base.end()) { j.copy_off = bi->second->off; j.copy_trans = bi->second->r < bi->second->c ? 1 : 0; }
                    j.cbeg = (uint32_t)cons.size();
                    auto pi = pairs.find(kv.first); if (pi != pairs.end()) cons.insert(cons.end(), pi->second.begin(), pi->second.end());
                    j.cend = (uint32_t)cons.size();
                    jobs.push_back(j);
                }
                for (int64_t k = 0; k < nb; ++k) if (!c->is_elim[k]) {
                    GatherJob j{}; j.copy_off = -1; j.r0 = (uint32_t)red_of[k]; j.rows = (uint16_t)c->blocksizes[k]; j.cols = 1; j.kind = 1; j.boff = (uint32_t)c->boffsets[k];
                    j.cbeg = (uint32_t)cons.size(); auto ri = rhs.find(j.r0); if (ri != rhs.end()) cons.insert(cons.end(), ri->second.begin(), ri->second.end()); j.cend = (uint32_t)cons.size();
                    jobs.push_back(j);
                }
                { GatherJob j{}; j.copy_off = -1; j.kind = 2; jobs.push_back(j); }
                if (hipSuccess != c->slab.alloc((size_t)std::max<uint64_t>(off, 4)) || hipSuccess != c->d_slab_off.upload(slab_off) || hipSuccess != c->d_slab_groups.upload(slab_groups) || hipSuccess != c->d_gjobs.upload(jobs) ||
                    hipSuccess != c->d_gcons.upload(cons)) return fail(c, NLLS_ERR_HIP, "gather index upload");
                c->n_gjobs = (int64_t)jobs.size(); c->elim_slab = (flags & NLLS_FLAG_DETERMINISTIC) != 0; c->gather_ready = true; c->h_slab_off = slab_off;
            }
        }
    } else {
        // dense storage, column-major, the rhs riding along as row n.  When the (re-ordered) reduced system is a WIDE band -- too wide for the band
        // kernels, much narrower than the system: a 2-D camera grid, a loop closure -- the blocked LDL' is restricted to the band and the border strip
        // (enqueue_reduced_solve, `dense_window`): O(n w^2) work instead of n^3 / 3.  The reference's LDL' takes any sparsity (src/linearsolver.jl:28-32).
        c->dense_window = I0.is_sparse && c->nelim_all > 0 && c->n_band >= 1024 && !(flags & NLLS_FLAG_NO_BAND) && 2 * (bw + 256) < c->n_band && !getenv("NLLS_NO_DENSE_WINDOW");
        c->dense_pad128 = c->dense_window;
        const int64_t npad = c->dense_pad128 ? ((n + 1 + 127) / 128) * 128 : ((n + 1 + 63) / 64) * 64;   // +1: the rhs rides along as an extra row
        // TILE-SPARSE: nested dissection of the reduced blocks' graph, the factorisation level by level of its elimination tree (nlls_tsp.hip).  Taken when its
        // dependent chain (levels of the tree) and its tile products come out clearly below what the dense / windowed factorisation of the same system costs
        // (rough launch + matrix-core times, in us).
        if (I0.is_sparse && c->nelim_all > 0 && n >= 512 && !red_adj_blocks.empty() && !(flags & (NLLS_FLAG_NO_BAND | NLLS_FLAG_NO_TILE_SPARSE)) && !getenv("NLLS_NO_TSPARSE")) {
            std::vector<int32_t> ndof, noff;
            for (int64_t k : red_adj_blocks) { ndof.push_back((int32_t)c->blocksizes[k]); noff.push_back((int32_t)red_of[k]); }
            int nbdn = 0; for (int64_t k = 0; k < nb; ++k) if (!c->is_elim[k] && red_of[k] >= c->n_band) { ndof.push_back((int32_t)c->blocksizes[k]); noff.push_back((int32_t)red_of[k]); ++nbdn; }
            TspSym sym;
            if (tsp_symbolic(red_adj, ndof, nbdn, sym) && sym.nt > 0 && sym.nt <= 4096) {
                const double NB128 = (double)(npad / 128 + (npad % 128 ? 1 : 0));
                // (calibrated on camera grids of 24 x 24 ... 100 x 100: a level = panel 30 + update 12 + backward 10 us; a 128^3 product of an update 0.09 us of the chip
                //  with the empty chunks skipped, of a panel 0.07; the dense trailing update 0.12 us per tile product)
                const double t_tsp = 52.0 * sym.nlevels + 0.09 * (double)sym.nupd_products + 0.07 * (double)(sym.ntiles_lower - sym.nt);
                double t_dense = 32.0 * NB128 + 0.12 * NB128 * NB128 * NB128 / 6.0;
                if (c->dense_window) { const double wt = (double)((bw + 127) / 128 + 1 + (c->nbd + 1 + 127) / 128); t_dense = std::min(t_dense, NB128 * (40.0 + 0.17 * wt * (wt + 1) / 2.0)); }
                const bool force = getenv("NLLS_FORCE_TSPARSE") != nullptr;
                if (t_tsp < 0.8 * t_dense || force) {
                    // (what the device holds decides: the tiles twice -- S and W -- + slots per tile; a system whose tiles do not fit falls back to the dense / windowed solver,
                    //  which then declines by its own size check)
                    size_t mfree = 0, mtotal = 0; const size_t want = sizeof(double) * (2 * ((size_t)sym.ntiles_lower * TSP_TE + (size_t)sym.nt * TSP_STRIP) + 3 * (size_t)sym.nt * TSP_TE + (size_t)n + 4096);
                    // (under sharding every rank must arrive at the SAME solver -- the layout of the summed [S | s] depends on it --, so nothing rank-local may decide:
                    //  no look at this device's free memory, and an allocation that fails is an error of the upload, not a quiet change of solver)
                    const bool fits = c->nranks > 1 || hipMemGetInfo(&mfree, &mtotal) != hipSuccess || want + ((size_t)2 << 30) <= mfree;
                    std::string e; const int rc = fits ? c->tsp.build(sym, noff, ndof, (int)n, &e, &red_adj, nbdn) : NLLS_ERR_UNSUPPORTED;
                    if (rc == NLLS_ERR_HIP && c->nranks == 1) { (void)hipGetLastError(); c->tsp.release(); }        // an allocation that failed after all: the other solvers
                    else if (rc != NLLS_OK && rc != NLLS_ERR_UNSUPPORTED) return fail(c, rc, e.c_str());
                }
            }
        }
        if (c->tsp.ready) {
            c->solve_mode = SOLVE_TSPARSE; c->dense_window = false; c->dense_pad128 = false;
            c->s_elems = c->tsp.s_elems();
            if (hipSuccess != c->S.alloc(c->s_elems + (size_t)n + 64) || hipSuccess != c->Lwork.alloc(64) || hipSuccess != c->d_status.alloc(96)) return fail(c, NLLS_ERR_HIP, "tile-sparse reduced system alloc");
        } else {
        // npad^2 doubles: what the DEVICE holds decides (288 GB on an MI355X: ~150 000 reduced dof), not a constant; the shim keeps the CPU system when it does not fit
        const size_t lw = (size_t)std::max<int64_t>(npad * 128 + npad + (npad / 16) * 256 + 256 + (npad / 64 + 1) * 128 * 128 + (npad / 128 + 1) * 128 * 128, 1);   // ... | inverses of the diagonal blocks (look-ahead)
        { size_t mfree = 0, mtotal = 0; const size_t want = sizeof(double) * ((size_t)npad * npad + (size_t)npad + 64 + lw);
          if (hipMemGetInfo(&mfree, &mtotal) == hipSuccess && want + ((size_t)2 << 30) > mfree)
              return fail(c, NLLS_ERR_UNSUPPORTED, "reduced system too large for the dense solver on this device (" + std::to_string(n) + " dof need " + std::to_string(want >> 20) + " MiB, " + std::to_string(mfree >> 20) + " MiB free)"); }
        if (npad >= ((int64_t)1 << 30)) return fail(c, NLLS_ERR_UNSUPPORTED, "reduced system too large for 32-bit row indices");
        c->s_elems = (size_t)std::max<int64_t>(npad * npad, 1);
        if (hipSuccess != c->S.alloc(c->s_elems + (size_t)npad + 64) ||
            hipSuccess != c->Lwork.alloc(lw)   /* W of a 128-column panel (or of two 64-column ones) | acc | inverted diagonal tiles | factored diagonal blocks (a slot per 64-block) */ || hipSuccess != c->d_status.alloc(96)) return fail(c, NLLS_ERR_HIP, "reduced system alloc");
        }
    }
    if ((flags & NLLS_FLAG_PRESHARDED) && c->nranks > 1 && c->reduce_fn) {
        // the solver of the reduced system and the size of [S | s] are part of the layout that is summed element by element: +x / -x pairs under MAX, equal on every rank or refused
        double h[4] = {(double)c->solve_mode, -(double)c->solve_mode, (double)c->s_elems, -(double)c->s_elems};
        DevBuf<double> dh; HIPCHK(dh.alloc(4));
        HIPCHK(hipMemcpyAsync(dh.p, h, sizeof h, hipMemcpyHostToDevice, c->stream));
        { const int rc = comm_reduce(c, dh.p, 4, NLLS_REDUCE_MAX); if (rc != NLLS_OK) return rc; }
        HIPCHK(hipMemcpyAsync(h, dh.p, sizeof h, hipMemcpyDeviceToHost, c->stream)); HIPCHK(hipStreamSynchronize(c->stream));
        if (h[0] != -h[1] || h[2] != -h[3]) return fail(c, NLLS_ERR_INVALID_ARG, "NLLS_FLAG_PRESHARDED: the ranks chose different solvers or sizes for the reduced system");
    }
    c->info.has_schur = c->nelim > 0; c->info.nschur_blocks = c->nelim; c->info.nreduced_dof = c->nred;
    c->info.solve_mode = c->solve_mode; c->info.bandwidth = c->bw; c->info.nborder_dof = c->nbd;
    // the small dense system's own route (nlls_ctx::tiny_dense): one image of [A | b] per sweep workgroup, summed by one gathering launch
    c->tiny_dense = false; c->dense_slab_wgs = 0;
    if (c->tiny_dense_on && !c->info.is_sparse && c->solve_mode == SOLVE_SMALL && c->nranks == 1 && c->nelim == 0 && c->info.ndof > 0 && c->nred == c->info.ndof) {
        bool ok = true; int64_t wgs = 0;
        for (const Group& G : c->groups) { if (is_dyn_kind(G.res_kind)) ok = false; if (G.dense.n > 0) wgs += std::min<int64_t>((G.dense.n + 255) / 256, TINY_DENSE_MAX_WGS); }
        if (ok && wgs > 0) {
            if (hipSuccess != c->dense_slab.alloc((size_t)wgs * (size_t)(c->info.ndof * c->info.ndof + c->info.ndof))) return fail(c, NLLS_ERR_HIP, "dense slab alloc");
            c->tiny_dense = true; c->dense_slab_wgs = wgs;
        }
    }
    return NLLS_OK;
}

}  // namespace nlls
