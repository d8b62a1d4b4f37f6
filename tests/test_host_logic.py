"""Host-side logic that needs no GPU: the ordering / independent-set rule of optimizesingles!, the sharding partition rule."""
import numpy as np

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
