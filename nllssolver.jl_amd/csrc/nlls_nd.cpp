// nlls_nd.cpp -- symbolic phase of the tile-sparse reduced solver (host only): nested dissection of the reduced blocks' graph, tiles, the
// elimination tree and the fill of the TILE graph.
//
// The reference hands the whole system to a sparse LDL' that analyses the pattern once with a fill-reducing ordering and takes any
// numbering (src/linearsystem.jl:52,68 ldl_analyze; src/linearsolver.jl:28-32).  Here the points are eliminated first (nlls_structure.cpp);
// what remains couples cameras that see a common point.  When that graph is neither a narrow band (block cyclic reduction, nlls_bcr.hip) nor
// small, a left-to-right factorisation is a chain of n / 128 dependent panel steps whatever the flops: what shortens the chain is an ordering
// whose elimination tree is SHALLOW -- nested dissection -- with every level of the tree one launch over all its tiles (nlls_tsp.hip).
#include <algorithm>
#include <cstdlib>
#include <numeric>

#include "nlls_tsp.hpp"

namespace nlls {
namespace {

struct Dissector {
    const std::vector<std::vector<int32_t>>& adj; const std::vector<int32_t>& dof;
    std::vector<int32_t> stamp, lev, q;
    int32_t cur = 0;
    int64_t leaf_dof = [] { const char* e = getenv("NLLS_TSP_LEAF"); const int v = e ? atoi(e) : 0; return (int64_t)(v > 0 ? v : 2 * TSP_TR); }();   // parts up to here are not cut further (A/B: NLLS_TSP_LEAF=<unknowns>)
    std::vector<std::vector<int32_t>> supernodes;      // in elimination order: parts before their separator
    std::vector<int32_t> parent;                       // the separator a part hangs under (-1: none)
    int32_t emit(std::vector<int32_t>&& nodes) { supernodes.push_back(std::move(nodes)); parent.push_back(-1); return (int32_t)supernodes.size() - 1; }
    Dissector(const std::vector<std::vector<int32_t>>& a, const std::vector<int32_t>& d) : adj(a), dof(d), stamp(a.size(), 0), lev(a.size(), -1) {}
    int64_t dofsum(const std::vector<int32_t>& nodes) const { int64_t s = 0; for (int32_t v : nodes) s += dof[v]; return s; }
    // breadth-first level structure from r among the nodes stamped id; q = visiting order; returns the eccentricity
    int32_t bfs(int32_t r, int32_t id) {
        q.clear(); q.push_back(r); lev[r] = 0; int32_t ecc = 0;
        for (size_t h = 0; h < q.size(); ++h) { const int32_t u = q[h]; ecc = lev[u];
            for (int32_t w : adj[u]) if (stamp[w] == id && lev[w] < 0) { lev[w] = lev[u] + 1; q.push_back(w); } }
        return ecc;
    }
    void clear() { for (int32_t u : q) lev[u] = -1; }
    // returns the supernodes at the top of what it emitted (they get the caller's separator as parent)
    std::vector<int32_t> run(std::vector<int32_t> nodes, int depth) {
        if (nodes.empty()) return {};
        const int64_t total = dofsum(nodes);
        if (total <= leaf_dof || depth >= 64) return {emit(std::move(nodes))};     // one or two tiles: a chain of at most two steps either way
        const int32_t id = ++cur; for (int32_t v : nodes) stamp[v] = id;
        // connected components: independent subtrees, no separator between them
        {
            std::vector<std::vector<int32_t>> comps;
            for (int32_t v : nodes) if (lev[v] < 0) { bfs(v, id); comps.emplace_back(q); }
            for (int32_t v : nodes) lev[v] = -1;
            if (comps.size() > 1) {
                std::vector<int32_t> tops, bucket; int64_t bdof = 0;         // small components share tiles
                for (auto& cmp : comps) { const int64_t d = dofsum(cmp);
                    if (d > TSP_TR) { const auto r = run(std::move(cmp), depth + 1); tops.insert(tops.end(), r.begin(), r.end()); continue; }
                    if (bdof + d > TSP_TR) { tops.push_back(emit(std::move(bucket))); bucket.clear(); bdof = 0; }
                    bucket.insert(bucket.end(), cmp.begin(), cmp.end()); bdof += d; }
                if (!bucket.empty()) tops.push_back(emit(std::move(bucket)));
                return tops;
            }
        }
        // pseudo-peripheral root (George & Liu): restart from a minimum-degree node of the last level while the eccentricity grows
        int32_t r = nodes[0]; for (int32_t v : nodes) if (adj[v].size() < adj[r].size()) r = v;
        int32_t ecc = bfs(r, id);
        for (int it = 0; it < 16; ++it) {
            int32_t cand = -1; for (int32_t u : q) if (lev[u] == ecc && (cand < 0 || adj[u].size() < adj[cand].size())) cand = u;
            clear();
            if (cand < 0 || cand == r) { bfs(r, id); break; }
            const int32_t e2 = bfs(cand, id);
            if (e2 > ecc) { r = cand; ecc = e2; } else { clear(); bfs(r, id); break; }
        }
        if (ecc < 2) { clear(); return {emit(std::move(nodes))}; }      // no interior level to cut at: a dense front
        std::vector<int64_t> ldof(ecc + 1, 0); for (int32_t u : q) ldof[lev[u]] += dof[u];
        std::vector<int64_t> cum(ecc + 2, 0); for (int32_t j = 0; j <= ecc; ++j) cum[j + 1] = cum[j] + ldof[j];
        int32_t best = -1; int64_t bsep = 0, bmin = -1;
        for (int32_t j = 1; j < ecc; ++j) { const int64_t a = cum[j], b = total - cum[j + 1], mn = std::min(a, b);
            if (5 * mn < total) continue;                                      // balanced enough: the smaller side holds a fifth
            if (best < 0 || ldof[j] < bsep || (ldof[j] == bsep && mn > bmin)) { best = j; bsep = ldof[j]; bmin = mn; } }
        if (best < 0) for (int32_t j = 1; j < ecc; ++j) { const int64_t mn = std::min(cum[j], total - cum[j + 1]); if (mn > bmin) { bmin = mn; best = j; } }
        std::vector<int32_t> A, B, sep;
        for (int32_t u : q) {
            if (lev[u] < best) A.push_back(u);
            else if (lev[u] > best) B.push_back(u);
            else { bool touches = false; for (int32_t w : adj[u]) if (stamp[w] == id && lev[w] == best + 1) { touches = true; break; }
                   (touches ? sep : A).push_back(u); }                         // a node of the cut level without a neighbour behind it separates nothing
        }
        clear();
        const auto ra = run(std::move(A), depth + 1), rb = run(std::move(B), depth + 1);
        const int32_t si = emit(std::move(sep));
        for (int32_t r : ra) parent[r] = si;
        for (int32_t r : rb) parent[r] = si;
        return {si};
    }
};

}  // namespace

bool tsp_symbolic(const std::vector<std::vector<int32_t>>& adj, const std::vector<int32_t>& dof, int nborder, TspSym& out) {
    const int32_t n = (int32_t)adj.size(); const int32_t nall = n + nborder;
    if ((int32_t)dof.size() != nall) return false;
    for (int32_t d : dof) if (d < 1 || d > TSP_TR) return false;
    out = TspSym{};
    out.tile_of.assign(nall, -1); out.row_in_tile.assign(nall, 0);
    // HUBS -- nodes coupled to an eighth of all nodes or more (an overview image that sees half the scene): no level structure has an interior level while they are
    // in the graph (everything is within two steps of everything).  They are taken out before the dissection and ordered behind it, next to the border nodes:
    // their tiles count as neighbours of every tile (the root front of the factorisation).
    std::vector<uint8_t> hub(n, 0); std::vector<int32_t> hubs; int64_t hubdof = 0;
    { const size_t thr = std::max<size_t>(48, (size_t)n / 8);
      for (int32_t v = 0; v < n; ++v) if (adj[v].size() >= thr && hubdof + dof[v] <= 16 * TSP_TR) { hub[v] = 1; hubs.push_back(v); hubdof += dof[v]; } }
    std::vector<std::vector<int32_t>> adj_nohub;
    if (!hubs.empty()) { adj_nohub.resize(n); for (int32_t v = 0; v < n; ++v) if (!hub[v]) for (int32_t w : adj[v]) if (!hub[w]) adj_nohub[v].push_back(w); }
    const std::vector<std::vector<int32_t>>& G = hubs.empty() ? adj : adj_nohub;
    Dissector D(G, dof);
    { std::vector<int32_t> all; all.reserve(n); for (int32_t v = 0; v < n; ++v) if (!hub[v]) all.push_back(v); D.run(std::move(all), 0); }
    // supernodes -> tiles.  A supernode starts a tile of its own (a tile that mixed two sibling parts would chain their subtrees) -- but the nodes of its LAST,
    // poorly filled tile move up into the first tile of the separator it hangs under (they are eliminated with that front instead: any order is a valid
    // one, and a tile shared by a separator and the tails of its own parts chains nothing that was not chained already).  Without this 30 % of all tile rows
    // are padding -- 2.9 x the tile products of full tiles.
    const int carry_max = [] { const char* e = getenv("NLLS_TSP_CARRY"); return e ? atoi(e) : 80; }();      // rows of a last tile up to which it is carried up (0: never)
    std::vector<std::vector<int32_t>> carried(D.supernodes.size());
    for (size_t si = 0; si < D.supernodes.size(); ++si) {
        std::vector<int32_t> sn = std::move(carried[si]); sn.insert(sn.end(), D.supernodes[si].begin(), D.supernodes[si].end());
        if (sn.empty()) continue;
        // the tail that a sequential packing would leave in the last tile
        size_t tail0 = 0; { int f = 0; for (size_t q = 0; q < sn.size(); ++q) { if (f + dof[sn[q]] > TSP_TR) { f = 0; tail0 = q; } f += dof[sn[q]]; }
                            if (D.parent[si] >= 0 && f <= carry_max) { auto& up = carried[D.parent[si]]; up.insert(up.end(), sn.begin() + tail0, sn.end()); sn.resize(tail0); } }
        if (sn.empty()) continue;
        out.fill.push_back(0);
        for (int32_t v : sn) { if (out.fill.back() + dof[v] > TSP_TR) out.fill.push_back(0);
            out.tile_of[v] = (int32_t)out.fill.size() - 1; out.row_in_tile[v] = out.fill.back(); out.fill.back() += dof[v]; }
    }
    bool dense_last = false; int32_t first_border_tile = -1;
    if (nborder > 0 || !hubs.empty()) {            // hubs and border nodes couple to everything: behind all the others, every tile from the first of theirs on a neighbour of every tile
        std::vector<int32_t> last = hubs; for (int32_t v = n; v < nall; ++v) last.push_back(v);
        // (they start in the last tile there is -- the tail of the last root front, which becomes a neighbour of everything with them: it was the last step anyway)
        for (int32_t v : last) { if (out.fill.empty() || out.fill.back() + dof[v] > TSP_TR) out.fill.push_back(0);
            if (first_border_tile < 0) first_border_tile = (int32_t)out.fill.size() - 1;
            out.tile_of[v] = (int32_t)out.fill.size() - 1; out.row_in_tile[v] = out.fill.back(); out.fill.back() += dof[v]; }
        dense_last = true;
    }
    const int nt = out.nt = (int)out.fill.size();
    if (nt == 0) return true;
    if (first_border_tile < 0) first_border_tile = nt;
    std::vector<std::vector<int32_t>> tadj(nt);
    for (int32_t v = 0; v < n; ++v) { const int32_t tv = out.tile_of[v]; for (int32_t w : G[v]) { const int32_t tw = out.tile_of[w]; if (tw != tv) tadj[tv].push_back(tw); } }      // (hubs: covered by dense_last)
    for (auto& l : tadj) { std::sort(l.begin(), l.end()); l.erase(std::unique(l.begin(), l.end()), l.end()); }
    out.cstruct.assign(nt, {}); out.parent.assign(nt, -1); out.level.assign(nt, 0);
    std::vector<std::vector<int32_t>> children(nt);
    for (int32_t k = 0; k < nt; ++k) {
        auto& s = out.cstruct[k];
        for (int32_t t : tadj[k]) if (t > k) s.push_back(t);
        for (int32_t ch : children[k]) for (int32_t t : out.cstruct[ch]) if (t != k) s.push_back(t);
        if (dense_last) for (int32_t t = std::max(first_border_tile, k + 1); t < nt; ++t) s.push_back(t);
        std::sort(s.begin(), s.end()); s.erase(std::unique(s.begin(), s.end()), s.end());
        if (!s.empty()) { out.parent[k] = s[0]; children[s[0]].push_back(k); }
        for (int32_t ch : children[k]) out.level[k] = std::max(out.level[k], out.level[ch] + 1);
        out.nlevels = std::max(out.nlevels, out.level[k] + 1);
        out.ntiles_lower += 1 + (int64_t)s.size();
        out.nupd_products += (int64_t)s.size() * ((int64_t)s.size() + 1) / 2;
    }
    return true;
}

}  // namespace nlls
