"""Host-side logic that needs no GPU: the ordering / independent-set rule of optimizesingles!, the sharding partition rule."""
import numpy as np
import pytest

import nllssolver_jl_amd as N
from nllssolver_jl_amd import kinds as K
from nllssolver_jl_amd import synthetic
from nllssolver_jl_amd.dist import partition_by_weight


def test_singles_levels_reproduce_the_sequential_order():
    """src/optimize.jl:183-205 relaxes the listed variables one after the other.  singles_levels must put a variable strictly after
    every EARLIER listed variable it shares a cost block with, and never separate variables that share nothing."""
    p = synthetic.create_ba_problem(5, 40, 0.7, seed=2)
    ncam = 5
    pts = np.nonzero((p.var_kind == K.VAR_EUCLIDEAN) & (p.var_dim == 3))[0] + 1
    cams = np.arange(1, ncam + 1)
    assert p.singles_levels(pts).max() == 0                               # points share no block: one launch
    assert p.singles_levels(cams).max() == 0
    order = np.concatenate([pts, cams])                                   # the reference's order: by variable size, stable
    lv = p.singles_levels(order)
    assert np.all(lv[:pts.size] == 0) and np.all(lv[pts.size:] == 1)
    lv2 = p.singles_levels(np.concatenate([cams[:2], pts, cams[2:]]))     # an arbitrary listed order is respected as given
    assert np.all(lv2[:2] == 0) and lv2[2:2 + pts.size].max() == 1 and lv2[2 + pts.size:].max() == 2
    # exhaustive check of the rule on the last order: every pair of co-listed variables is separated in the right direction
    seq = np.concatenate([cams[:2], pts, cams[2:]])
    (g,) = p.costs.values(); vi, _ = g.arrays()
    pos = {int(v): i for i, v in enumerate(seq)}
    for a, b in vi:
        ia, ib = pos[int(a)], pos[int(b)]
        assert (lv2[ia] < lv2[ib]) == (ia < ib)


def test_partition_by_weight_is_contiguous_and_balanced():
    w = np.random.default_rng(0).integers(1, 30, size=1000)
    b = partition_by_weight(w, 4)
    assert b[0] == 0 and b[-1] == 1000 and np.all(np.diff(b) >= 0)
    parts = [w[b[i]:b[i + 1]].sum() for i in range(4)]
    assert max(parts) - min(parts) <= 2 * w.max()


def _camera_graph(ncam, npts, prop, seed):
    """CSR graph of the reduced camera system of the reference's BA generator (test/optimizeba.jl:22-23) with the cameras' labels permuted:
    two cameras are coupled when one point sees both."""
    cam, lm = synthetic.ba_visibility(ncam, npts, prop)
    perm = np.random.default_rng(seed).permutation(ncam) if seed is not None else np.arange(ncam)
    inv = np.empty(ncam, np.int64); inv[perm] = np.arange(ncam)
    c2 = inv[cam - 1]; order = np.lexsort((c2, lm)); c2, l2 = c2[order], lm[order]
    starts = np.r_[0, np.nonzero(np.diff(l2))[0] + 1, len(l2)]
    pairs = set()
    for s in {tuple(c2[starts[i]:starts[i + 1]]) for i in range(len(starts) - 1)}:
        pairs.update((a, b) for a in s for b in s if a != b)
    pa = np.array(sorted(pairs), dtype=np.int64)
    ptr = np.zeros(ncam + 1, np.int64); np.add.at(ptr, pa[:, 0] + 1, 1)
    return np.cumsum(ptr), pa[:, 1].astype(np.int32).copy(), pa


def _rcm(ptr, adj):
    from nllssolver_jl_amd import _capi
    n = len(ptr) - 1; out = np.zeros(n, np.int32)
    assert _capi.lib().nlls_rcm_order(n, _capi._p(ptr), _capi._p(adj), _capi._p(out)) == 0
    assert sorted(out.tolist()) == list(range(n))                         # a permutation
    return out


def test_rcm_order_recovers_the_band_of_relabelled_cameras():
    """The reference's factorisation orders itself (ldl_analyze, src/linearsystem.jl:52,68); here the ordering of the reduced camera system is
    reverse Cuthill-McKee (host side, nlls_rcm_order -- the routine nlls_upload_structure applies).  On the reference generator's visibility
    (test/optimizeba.jl:22-23) with shuffled camera labels it must give back the band of the unshuffled numbering."""
    for ncam, npts, prop, seed in [(40, 1500, 0.12, 31), (150, 6000, 0.04, 32), (100, 10000, 0.1, 5), (1000, 100000, 0.01, 7)]:
        ptr0, adj0, pa0 = _camera_graph(ncam, npts, prop, None)
        bw0 = int(np.abs(pa0[:, 0] - pa0[:, 1]).max())
        ptr, adj, pa = _camera_graph(ncam, npts, prop, seed)
        assert int(np.abs(pa[:, 0] - pa[:, 1]).max()) > 4 * bw0            # the shuffled labels have no band
        out = _rcm(ptr, adj)
        pos = np.empty(ncam, np.int64); pos[out] = np.arange(ncam)
        assert int(np.abs(pos[pa[:, 0]] - pos[pa[:, 1]]).max()) <= int(1.25 * bw0)


def test_rcm_order_components_grid_and_bad_input():
    from nllssolver_jl_amd import _capi
    # two components (a path of 5 and a triangle) + an isolated node: every node is placed, components stay contiguous
    edges = [(0, 1), (1, 2), (2, 3), (3, 4), (5, 6), (6, 7), (5, 7)]
    n = 9; nb = [[] for _ in range(n)]
    for a, b in edges:
        nb[a].append(b); nb[b].append(a)
    ptr = np.cumsum([0] + [len(x) for x in nb]).astype(np.int64); adj = np.array([y for x in nb for y in x], np.int32)
    out = _rcm(ptr, adj); pos = np.empty(n, np.int64); pos[out] = np.arange(n)
    assert max(abs(pos[a] - pos[b]) for a, b in edges) <= 2
    for comp in ([0, 1, 2, 3, 4], [5, 6, 7]):
        p = sorted(pos[comp]); assert p[-1] - p[0] == len(comp) - 1
    # a 2-D grid (k x k, 4-neighbourhood): the bandwidth of any level-set ordering is about k
    k = 12; idx = lambda i, j: i * k + j; nb = [[] for _ in range(k * k)]
    for i in range(k):
        for j in range(k):
            for di, dj in ((1, 0), (0, 1)):
                if i + di < k and j + dj < k:
                    nb[idx(i, j)].append(idx(i + di, j + dj)); nb[idx(i + di, j + dj)].append(idx(i, j))
    perm = np.random.default_rng(0).permutation(k * k); inv = np.empty(k * k, np.int64); inv[perm] = np.arange(k * k)
    nb2 = [[] for _ in range(k * k)]
    for a in range(k * k):
        nb2[inv[a]] = sorted(int(inv[b]) for b in nb[a])
    ptr = np.cumsum([0] + [len(x) for x in nb2]).astype(np.int64); adj = np.array([y for x in nb2 for y in x], np.int32)
    out = _rcm(ptr, adj); pos = np.empty(k * k, np.int64); pos[out] = np.arange(k * k)
    assert max(abs(pos[a] - pos[b]) for a in range(k * k) for b in nb2[a]) <= k + 2
    # a self loop / an index out of range is an argument error, not a crash
    bad = np.array([0], np.int32); out1 = np.zeros(1, np.int32)
    assert _capi.lib().nlls_rcm_order(1, _capi._p(np.array([0, 1], np.int64)), _capi._p(bad), _capi._p(out1)) == _capi.ERR_INVALID_ARG


def test_contaminated_gaussian_em_recovers_the_mixture():
    """optimize(kernel::ContaminatedGaussian, squarederrors) (src/robustadaptive.jl:48-73), the host-side EM step of the reference's EM callback
    (test/adaptivecost.jl:15-25): on squared errors drawn from 0.8 N(0, 1) + 0.2 N(0, 10^2) it must recover (1, 10, 0.8) to the tolerance the reference's
    own test uses (rtol 0.1, test/adaptivecost.jl:57), from the reference's starting kernel (0.5, 5, 0.6)."""
    from nllssolver_jl_amd.variables import contaminated_gaussian, contaminated_gaussian_em, contaminated_gaussian_params
    rng = np.random.default_rng(1)
    x = np.concatenate([rng.standard_normal(8000), rng.standard_normal(2000) * 10.0])
    k = contaminated_gaussian(0.5, 5.0, 0.6)
    for _ in range(20):
        k = contaminated_gaussian_em(k, x * x)
    assert np.allclose(contaminated_gaussian_params(k), [1.0, 10.0, 0.8], rtol=0.1), contaminated_gaussian_params(k)
    assert k[0] >= k[1]                                                       # the narrowest Gaussian first (src/robustadaptive.jl:14)
    # one step from the truth stays at the truth
    k2 = contaminated_gaussian_em(contaminated_gaussian(1.0, 10.0, 0.8), x * x, maxiters=1)
    assert np.allclose(contaminated_gaussian_params(k2), [1.0, 10.0, 0.8], rtol=0.1)


def _grid_graph(gw, gh, reach=2):
    """cameras on a gw x gh grid, coupled when they lie within `reach` of each other in both directions (what a point seen by a 3 x 3 neighbourhood leaves)"""
    idx = lambda x, y: y * gw + x
    nbrs = [[] for _ in range(gw * gh)]
    for y in range(gh):
        for x in range(gw):
            for dy in range(-reach, reach + 1):
                for dx in range(-reach, reach + 1):
                    xx, yy = x + dx, y + dy
                    if (dx or dy) and 0 <= xx < gw and 0 <= yy < gh:
                        nbrs[idx(x, y)].append(idx(xx, yy))
    ptr = np.zeros(gw * gh + 1, np.int64); ptr[1:] = np.cumsum([len(l) for l in nbrs])
    return ptr, np.array([w for l in nbrs for w in l], np.int32)


def _nd_tiles(ptr, adj, dof, nborder=0):
    from nllssolver_jl_amd import _capi
    n = len(ptr) - 1; nall = n + nborder
    tile_of = np.zeros(nall, np.int32); row = np.zeros(nall, np.int32); mt = nall + 1
    parent = np.zeros(mt, np.int32); level = np.zeros(mt, np.int32); colptr = np.zeros(mt + 1, np.int64); rows = np.zeros(mt * mt // 2 + 1, np.int32)
    nt = _capi.lib().nlls_nd_tiles(n, nborder, _capi._p(ptr), _capi._p(adj), _capi._p(dof), _capi._p(tile_of), _capi._p(row), mt, _capi._p(parent), _capi._p(level),
                                   _capi._p(colptr), len(rows), _capi._p(rows))
    assert nt > 0, nt
    return nt, tile_of, row, parent[:nt], level[:nt], colptr[:nt + 1], rows


@pytest.mark.parametrize("gw,gh,nborder,seed", [(12, 12, 0, 0), (17, 9, 0, 1), (14, 14, 2, 2)])
def test_nd_tiles_symbolic_phase(gw, gh, nborder, seed):
    """The symbolic phase of the tile-sparse reduced solver (nlls_nd_tiles -- what nlls_upload_structure runs; the reference's counterpart is ldl_analyze,
    src/linearsystem.jl:52,68): every node in exactly one tile of at most 128 rows; the NUMERIC Cholesky factor of a random matrix with the graph's pattern,
    permuted into the tile order, has no entry outside the predicted tile pattern; tiles of one level do not touch; the tree is much shallower than the
    number of tile columns (the dependent chain of the factorisation)."""
    rng = np.random.default_rng(seed)
    ptr, adj = _grid_graph(gw, gh)
    if seed == 1:                                              # relabelled cameras: the result must not depend on the numbering
        n = gw * gh; p = rng.permutation(n); inv = np.argsort(p)
        lists = [sorted(int(inv[w]) for w in adj[ptr[p[i]]:ptr[p[i] + 1]]) for i in range(n)]
        ptr = np.zeros(n + 1, np.int64); ptr[1:] = np.cumsum([len(l) for l in lists]); adj = np.array([w for l in lists for w in l], np.int32)
    n = len(ptr) - 1; nall = n + nborder
    dof = np.full(nall, 6, np.int32); dof[n:] = 3
    nt, tile_of, row, parent, level, colptr, rows = _nd_tiles(ptr, adj, dof, nborder)
    pos = tile_of.astype(np.int64) * 128 + row
    used = np.zeros(nt * 128, bool)
    for v in range(nall):
        assert row[v] + dof[v] <= 128 and not used[pos[v]:pos[v] + dof[v]].any(); used[pos[v]:pos[v] + dof[v]] = True
    # a random symmetric positive definite matrix with the graph's block pattern (+ border rows coupled to everything), in tile order, identity on the padding
    N = nt * 128; M = np.zeros((N, N))
    for v in range(n):
        for w in adj[ptr[v]:ptr[v + 1]]:
            if w < v:
                B = rng.normal(size=(dof[v], dof[w])); M[pos[v]:pos[v] + dof[v], pos[w]:pos[w] + dof[w]] = B; M[pos[w]:pos[w] + dof[w], pos[v]:pos[v] + dof[v]] = B.T
    for b in range(n, nall):
        B = rng.normal(size=(dof[b], N)) * used; M[pos[b]:pos[b] + dof[b], :] = B; M[:, pos[b]:pos[b] + dof[b]] = B.T
    M[np.arange(N), np.arange(N)] = np.where(used, 2.0 * np.abs(M).sum(axis=1).max() + 1.0, 1.0)
    Lf = np.linalg.cholesky(M)
    struct = [set(rows[colptr[k]:colptr[k + 1]].tolist()) for k in range(nt)]
    for k in range(nt):
        for i in range(k + 1, nt):
            if np.abs(Lf[128 * i:128 * i + 128, 128 * k:128 * k + 128]).max() > 1e-13:
                assert i in struct[k], (i, k)
        assert all(level[i] > level[k] for i in struct[k])              # what a tile updates is factored in a later launch
        assert parent[k] == (min(struct[k]) if struct[k] else -1)
    assert level.max() + 1 <= 0.7 * nt, (level.max() + 1, nt)


def test_nd_tiles_rejects_bad_input():
    from nllssolver_jl_amd import _capi
    z = np.zeros(8, np.int32); z64 = np.zeros(8, np.int64)
    bad = np.array([5], np.int32)
    assert _capi.lib().nlls_nd_tiles(1, 0, _capi._p(np.array([0, 1], np.int64)), _capi._p(bad), _capi._p(np.array([6], np.int32)), _capi._p(z), _capi._p(z), 4, _capi._p(z), _capi._p(z), _capi._p(z64), 4, _capi._p(z)) == _capi.ERR_INVALID_ARG
    assert _capi.lib().nlls_nd_tiles(1, 0, _capi._p(np.array([0, 0], np.int64)), _capi._p(bad), _capi._p(np.array([129], np.int32)), _capi._p(z), _capi._p(z), 4, _capi._p(z), _capi._p(z), _capi._p(z64), 4, _capi._p(z)) == _capi.ERR_INVALID_ARG


def test_nd_tiles_takes_hub_cameras_out_of_the_dissection():
    """Three overview cameras coupled to 40 % of a 20 x 20 camera grid: with them in the graph everything is within two steps of everything and no breadth-first
    level structure has an interior level to cut at (one dense front: as many levels as tiles).  The symbolic phase orders such hubs last -- the root front, a
    neighbour of every tile -- and dissects the rest: the tree stays shallow, and the numeric factor of a random matrix with the pattern stays inside the
    predicted tile pattern."""
    rng = np.random.default_rng(4)
    gw = gh = 20; ptr, adj = _grid_graph(gw, gh); n0 = gw * gh; nh = 3
    lists = [adj[ptr[i]:ptr[i + 1]].tolist() for i in range(n0)] + [[] for _ in range(nh)]
    for h in range(nh):
        for v in rng.choice(n0, size=int(0.4 * n0), replace=False):
            lists[n0 + h].append(int(v)); lists[int(v)].append(n0 + h)
    n = n0 + nh
    ptr2 = np.zeros(n + 1, np.int64); ptr2[1:] = np.cumsum([len(l) for l in lists]); adj2 = np.array([w for l in lists for w in l], np.int32)
    dof = np.full(n, 6, np.int32)
    nt, tile_of, row, parent, level, colptr, rows = _nd_tiles(ptr2, adj2, dof)
    assert level.max() + 1 <= 0.6 * nt, (level.max() + 1, nt)
    assert all(tile_of[n0 + h] >= nt - 2 for h in range(nh))                      # the hubs: at the very end (they start in the last front's tail tile)
    pos = tile_of.astype(np.int64) * 128 + row
    N = nt * 128; M = np.zeros((N, N)); used = np.zeros(N, bool)
    for v in range(n):
        used[pos[v]:pos[v] + 6] = True
        for w in lists[v]:
            if w < v:
                B = rng.normal(size=(6, 6)); M[pos[v]:pos[v] + 6, pos[w]:pos[w] + 6] = B; M[pos[w]:pos[w] + 6, pos[v]:pos[v] + 6] = B.T
    M[np.arange(N), np.arange(N)] = np.where(used, 2.0 * np.abs(M).sum(axis=1).max() + 1.0, 1.0)
    Lf = np.linalg.cholesky(M)
    struct = [set(rows[colptr[k]:colptr[k + 1]].tolist()) for k in range(nt)]
    for k in range(nt):
        for i in range(k + 1, nt):
            if np.abs(Lf[128 * i:128 * i + 128, 128 * k:128 * k + 128]).max() > 1e-13:
                assert i in struct[k], (i, k)


def test_nd_tiles_on_degenerate_graphs():
    """The symbolic phase on graphs a dissection has nothing to cut in, or too much: no edges at all, a path, a star, one clique, several components of very
    different sizes, two nodes, nodes of 128 unknowns each, a random sparse graph with isolated nodes -- every node placed exactly once, the tree a tree, the
    pattern closed under elimination (struct[k] without its first entry lies inside struct of that entry: the defining property of a symbolic factorisation)."""
    rng = np.random.default_rng(0)
    def csr(n, edges):
        lists = [set() for _ in range(n)]
        for a, b in edges:
            if a != b: lists[a].add(b); lists[b].add(a)
        ptr = np.zeros(n + 1, np.int64); ptr[1:] = np.cumsum([len(l) for l in lists])
        return ptr, np.array([w for l in lists for w in sorted(l)], np.int32)
    cases = []
    cases.append((300, [], 6))                                                        # no edges
    cases.append((500, [(i, i + 1) for i in range(499)], 6))                          # a path
    cases.append((400, [(0, i) for i in range(1, 400)], 6))                           # a star (its centre is a hub)
    cases.append((90, [(i, j) for i in range(90) for j in range(i)], 6))              # one clique: a dense front
    comp = [(i, i + 1) for i in range(0, 700)] + [(800 + i, 800 + j) for i in range(40) for j in range(i)] + [(900, 901)]
    cases.append((1000, comp, 6))                                                     # components: a long path, a clique, a pair, isolated nodes
    cases.append((2, [(0, 1)], 6))
    cases.append((40, [(i, (i + 1) % 40) for i in range(40)], 128))                   # a ring of nodes that fill a tile each
    e = rng.integers(0, 2000, size=(5000, 2)); cases.append((2000, [tuple(x) for x in e.tolist()], 3))
    for n, edges, d in cases:
        ptr, adj = csr(n, edges); dof = np.full(n, d, np.int32)
        nt, tile_of, row, parent, level, colptr, rows = _nd_tiles(ptr, adj, dof)
        pos = tile_of.astype(np.int64) * 128 + row
        used = np.zeros(nt * 128, bool)
        for v in range(n):
            assert 0 <= tile_of[v] < nt and row[v] + d <= 128 and not used[pos[v]:pos[v] + d].any(); used[pos[v]:pos[v] + d] = True
        struct = [rows[colptr[k]:colptr[k + 1]].tolist() for k in range(nt)]
        for k in range(nt):
            assert struct[k] == sorted(set(struct[k])) and all(i > k for i in struct[k])
            assert parent[k] == (struct[k][0] if struct[k] else -1)
            if struct[k]:
                assert set(struct[k][1:]) <= set(struct[struct[k][0]]), (n, k)          # closed under elimination
                assert level[struct[k][0]] > level[k]
        # every edge of the graph lies inside the pattern
        for v in range(n):
            for w in adj[ptr[v]:ptr[v + 1]]:
                a, b = sorted((int(tile_of[v]), int(tile_of[w])))
                assert a == b or b in struct[a], (n, v, w)
