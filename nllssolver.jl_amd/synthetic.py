"""Synthetic problem generators restating the reference's own test generators.

create_ba_problem / perturb_ba_problem follow test/optimizeba.jl:6-47 so that the STRUCTURE
(variable order, banded visibility, camera-major cost order, noiseless measurements) is exactly
the reference's; random values come from numpy's PCG64 (Julia's RNG stream is not reproducible
outside Julia), which is what "values match in distribution" means in SURVEY 8d.
"""
import numpy as np

from . import kinds as K
from .problem import NLLSProblem


def ba_visibility(ncameras, nlandmarks, propvisible):
    """visibility[c, l] = |c - t_l| <= K-th smallest, t = LinRange(2, ncameras-1, nlandmarks)
    (test/optimizeba.jl:22-23).  Returns (camind, landmark) 1-based pairs in camera-major order."""
    t = np.linspace(2.0, ncameras - 1.0, nlandmarks) if nlandmarks > 1 else np.array([2.0])
    total = ncameras * nlandmarks
    kth = int(np.ceil(total * propvisible))
    if total <= 40_000_000:
        vis = np.abs(np.arange(1, ncameras + 1, dtype=np.float64)[:, None] - t[None, :])
        thr = np.partition(vis.ravel(), kth - 1)[kth - 1]
        cam, lm = np.nonzero(vis <= thr)               # row-major nonzero == camera-major order
        return cam + 1, lm + 1
    # large case: bisection on the threshold using closed-form per-landmark counts
    def count(tau):
        lo = np.maximum(np.ceil(t - tau), 1); hi = np.minimum(np.floor(t + tau), ncameras)
        return int(np.maximum(hi - lo + 1, 0).sum())
    a, b = 0.0, float(ncameras)
    for _ in range(200):
        m = 0.5 * (a + b)
        if count(m) >= kth:
            b = m
        else:
            a = m
        if b - a <= np.spacing(b):
            break
    thr = b
    lo = np.maximum(np.ceil(t - thr), 1).astype(np.int64); hi = np.minimum(np.floor(t + thr), ncameras).astype(np.int64)
    cnt = np.maximum(hi - lo + 1, 0)
    lm = np.repeat(np.arange(1, nlandmarks + 1), cnt)
    start = np.repeat(lo, cnt); pos = np.arange(cnt.sum()) - np.repeat(np.cumsum(cnt) - cnt, cnt)
    cam = start + pos
    order = np.lexsort((lm, cam))                      # camera-major (test/optimizeba.jl:24-31)
    return cam[order], lm[order]


def create_ba_problem(ncameras, nlandmarks, propvisible, seed=1, robust=None, outlier_frac=0.0, outlier_sigma=0.0):
    """test/optimizeba.jl:6-35: affine 6-dof cameras, 3-dof landmarks, 2-dim reprojection residuals."""
    rng = np.random.default_rng(seed)
    problem = NLLSProblem()
    cams = rng.standard_normal((ncameras, 6)) + np.array([1.0, 0, 0, 0, 1.0, 0])      # :10-13
    pts = rng.random((nlandmarks, 3)) + np.array([-0.5, -0.5, 10.0])                  # :16-19
    problem.addvariables(cams)
    problem.addvariables(pts)
    cam, lm = ba_visibility(ncameras, nlandmarks, propvisible)
    c, X = cams[cam - 1], pts[lm - 1]
    meas = np.stack([(c[:, 0:3] * X).sum(1), (c[:, 3:6] * X).sum(1)], axis=1)          # generatemeasurement :4
    if outlier_frac > 0:
        bad = rng.random(meas.shape[0]) < outlier_frac
        meas[bad] += rng.standard_normal((int(bad.sum()), 2)) * outlier_sigma
    varind = np.stack([cam, lm + ncameras], axis=1)
    problem.addcosts(K.RES_BA_AFFINE, varind, meas, robust)
    return problem


def create_ba_problem_shard(ncameras, nlandmarks, propvisible, rank, world, seed=1, robust=None, outlier_frac=0.0, outlier_sigma=0.0, pointnoise=0.0, posenoise=0.0):
    """Rank `rank`'s share of create_ba_problem(ncameras, nlandmarks, ...) WITHOUT building the whole: all cameras (the same values on every rank) and
    the contiguous range of landmarks [l0, l1) this rank owns, with exactly the cost blocks of the whole problem that touch them (same visibility
    threshold, camera-major order).  For NLLS_FLAG_PRESHARDED (ShardedLS(..., presharded=True)): each of N processes generates and uploads 1/N
    of an N-times larger job.  Landmark coordinates and outliers come from per-rank random streams (a workload, not a parity target; the
    structure is the reference generator's, test/optimizeba.jl:6-35)."""
    rng = np.random.default_rng(seed)
    cams = rng.standard_normal((ncameras, 6)) + np.array([1.0, 0, 0, 0, 1.0, 0])
    l0, l1 = (nlandmarks * rank) // world, (nlandmarks * (rank + 1)) // world
    t = np.linspace(2.0, ncameras - 1.0, nlandmarks) if nlandmarks > 1 else np.array([2.0])
    kth = int(np.ceil(ncameras * nlandmarks * propvisible))
    def count(tau):
        lo = np.maximum(np.ceil(t - tau), 1); hi = np.minimum(np.floor(t + tau), ncameras)
        return int(np.maximum(hi - lo + 1, 0).sum())
    a, b = 0.0, float(ncameras)
    for _ in range(200):
        m = 0.5 * (a + b)
        if count(m) >= kth:
            b = m
        else:
            a = m
        if b - a <= np.spacing(b):
            break
    thr = b
    tl = t[l0:l1]
    lo = np.maximum(np.ceil(tl - thr), 1).astype(np.int64); hi = np.minimum(np.floor(tl + thr), ncameras).astype(np.int64)
    cnt = np.maximum(hi - lo + 1, 0)
    lm = np.repeat(np.arange(1, l1 - l0 + 1), cnt)                 # local landmark numbers, 1-based
    start = np.repeat(lo, cnt); pos = np.arange(cnt.sum()) - np.repeat(np.cumsum(cnt) - cnt, cnt)
    cam = start + pos
    order = np.lexsort((lm, cam)); cam, lm = cam[order], lm[order]
    rng_l = np.random.default_rng([seed, 7919, rank])
    pts = rng_l.random((l1 - l0, 3)) + np.array([-0.5, -0.5, 10.0])
    problem = NLLSProblem()
    # starting point (perturb_ba_problem's role): the cameras' noise from the SHARED stream (every rank must start from the same cameras)
    problem.addvariables(cams + rng.standard_normal(cams.shape) * posenoise); problem.addvariables(pts + rng_l.standard_normal(pts.shape) * pointnoise)
    c, X = cams[cam - 1], pts[lm - 1]
    meas = np.stack([(c[:, 0:3] * X).sum(1), (c[:, 3:6] * X).sum(1)], axis=1)
    if outlier_frac > 0:
        bad = rng_l.random(meas.shape[0]) < outlier_frac
        meas[bad] += rng_l.standard_normal((int(bad.sum()), 2)) * outlier_sigma
    problem.addcosts(K.RES_BA_AFFINE, np.stack([cam, lm + ncameras], axis=1), meas, robust)
    return problem


def shard_of_problem(problem, ncameras, rank, world):
    """The share of an EXISTING two-slot (camera, landmark) problem that create_ba_problem_shard would hand to `rank`: all cameras, a contiguous
    range of landmarks, the cost blocks that touch them (order kept).  Lets a test hold a pre-sharded run against the unsharded one."""
    g = next(iter(problem.costs.values())); vi, da = g.arrays()
    npts = problem.nvariables - ncameras
    l0, l1 = (npts * rank) // world, (npts * (rank + 1)) // world
    keep = (vi[:, 1] > ncameras + l0) & (vi[:, 1] <= ncameras + l1)
    off = problem.var_offsets
    q = NLLSProblem()
    q.addvariables(problem.variables[: 6 * ncameras].reshape(ncameras, 6)); q.addvariables(problem.variables[off[ncameras + l0]: off[ncameras + l0] + 3 * (l1 - l0)].reshape(l1 - l0, 3))
    vi2 = vi[keep].copy(); vi2[:, 1] -= l0
    q.addcosts(g.res_kind, vi2, da[keep], g.robust)
    return q


def widen_visibility(problem, ncameras, wide):
    """Real bundle-adjustment graphs have a tail of landmarks seen by many cameras (test/optimizeba.jl:22-23 leaves visibility a free parameter).
    `wide` = {landmark (1-based among the landmarks): number of cameras}: each listed landmark gets noiseless measurements from a window of that
    many consecutive cameras around the ones that already see it (all cameras when the number reaches ncameras).  Call BEFORE perturbing: the
    variables are then still the ground truth the measurements are generated from."""
    g = next(iter(problem.costs.values())); vi, da = g.arrays()
    off = problem.var_offsets; v = problem.variables
    have = {}
    for c, l in vi:
        have.setdefault(int(l) - ncameras, set()).add(int(c))
    add_vi, add_da = [], []
    for lm, k in wide.items():
        seen = have.get(lm, set()); k = min(int(k), ncameras)
        mid = int(round(np.mean(sorted(seen)))) if seen else ncameras // 2
        lo = max(1, min(mid - k // 2, ncameras - k + 1))
        X = v[off[ncameras + lm - 1]: off[ncameras + lm - 1] + 3]
        for c in range(lo, lo + k):
            if c in seen:
                continue
            cam = v[off[c - 1]: off[c - 1] + 6]
            add_vi.append((c, ncameras + lm)); add_da.append((cam[0:3] @ X, cam[3:6] @ X))
    if add_vi:
        g.set_arrays(np.concatenate([vi, np.array(add_vi, dtype=vi.dtype)]), np.concatenate([da, np.array(add_da)]))
    return problem


def grid_visibility(gw, gh, pts_per_cell):
    """(camera, landmark) pairs, 1-based and camera-major, of gw x gh cameras on a grid with pts_per_cell landmarks per cell, each seen by the 3 x 3 block of
    cameras around its cell (clipped at the border); the number of landmarks."""
    cx, cy = np.meshgrid(np.arange(gw), np.arange(gh), indexing="xy")
    cell = np.repeat(np.stack([cx.ravel(), cy.ravel()], axis=1), pts_per_cell, axis=0)            # the cell of every landmark
    cam_l, lm_l = [], []
    for dx in (-1, 0, 1):
        for dy in (-1, 0, 1):
            x, y = cell[:, 0] + dx, cell[:, 1] + dy
            ok = (x >= 0) & (x < gw) & (y >= 0) & (y < gh)
            cam_l.append((y[ok] * gw + x[ok]) + 1); lm_l.append(np.nonzero(ok)[0] + 1)
    cam, lm = np.concatenate(cam_l), np.concatenate(lm_l)
    order = np.lexsort((lm, cam))
    return cam[order], lm[order], cell.shape[0]


def scattered_visibility(ncameras, nlandmarks, nviews, seed, loop=False, overview=0, overview_frac=0.3):
    """(camera, landmark) pairs, 1-based and camera-major, of cameras SCATTERED over the unit square (loop=True: along a closed ring -- a loop closure), every
    landmark at a random place seen by the `nviews` cameras nearest to it: an unstructured camera graph (no grid, no numbering to exploit)."""
    rng = np.random.default_rng(seed)
    if loop:
        t = np.sort(rng.random(ncameras)) * 2 * np.pi
        cpos = np.stack([np.cos(t), np.sin(t)], axis=1) * (1.0 + 0.02 * rng.standard_normal((ncameras, 1)))
        tl = rng.random(nlandmarks) * 2 * np.pi
        lpos = np.stack([np.cos(tl), np.sin(tl)], axis=1) * (1.0 + 0.05 * rng.standard_normal((nlandmarks, 1)))
    else:
        cpos = rng.random((ncameras, 2)); lpos = rng.random((nlandmarks, 2))
    from scipy.spatial import cKDTree
    _, nn = cKDTree(cpos).query(lpos, k=nviews)
    cam = nn.ravel() + 1; lm = np.repeat(np.arange(nlandmarks), nviews) + 1
    # (a camera that is among the nearest of NO landmark would be a variable without a cost block -- a zero diagonal block: it sees its own three nearest landmarks)
    lonely = np.flatnonzero(np.bincount(cam - 1, minlength=ncameras) < 3)
    if lonely.size:
        _, nl = cKDTree(lpos).query(cpos[lonely], k=3)
        have = set(zip(cam.tolist(), lm.tolist()))
        extra = [(int(c) + 1, int(l) + 1) for c, row in zip(lonely, np.atleast_2d(nl)) for l in row if (int(c) + 1, int(l) + 1) not in have]
        if extra:
            cam = np.concatenate([cam, [e[0] for e in extra]]); lm = np.concatenate([lm, [e[1] for e in extra]])
    for h in range(overview):                      # OVERVIEW cameras (the last `overview` ones): each also sees a random `overview_frac` of all landmarks
        seen = np.flatnonzero(rng.random(nlandmarks) < overview_frac) + 1
        c = ncameras - overview + h + 1
        seen = seen[~np.isin(seen, lm[cam == c])]
        cam = np.concatenate([cam, np.full(seen.size, c)]); lm = np.concatenate([lm, seen])
    order = np.lexsort((lm, cam))
    return cam[order], lm[order]


def create_scattered_ba_problem(ncameras, nlandmarks, nviews=6, seed=1, robust=None, outlier_frac=0.0, outlier_sigma=0.0, noise=0.0, loop=False, overview=0):
    """The affine-camera bundle adjustment of test/optimizeba.jl:4-35 over scattered_visibility (visibility is that generator's free parameter, :22-23)."""
    rng = np.random.default_rng(seed + 1000)
    cams = rng.standard_normal((ncameras, 6)) + np.array([1.0, 0, 0, 0, 1.0, 0])
    pts = rng.random((nlandmarks, 3)) + np.array([-0.5, -0.5, 10.0])
    cam, lm = scattered_visibility(ncameras, nlandmarks, nviews, seed, loop, overview)
    problem = NLLSProblem(); problem.addvariables(cams); problem.addvariables(pts)
    c, X = cams[cam - 1], pts[lm - 1]
    meas = np.stack([(c[:, 0:3] * X).sum(1), (c[:, 3:6] * X).sum(1)], axis=1)
    if noise > 0:
        meas += rng.standard_normal(meas.shape) * noise
    if outlier_frac > 0:
        bad = rng.random(meas.shape[0]) < outlier_frac
        meas[bad] += rng.standard_normal((int(bad.sum()), 2)) * outlier_sigma
    problem.addcosts(K.RES_BA_AFFINE, np.stack([cam, lm + ncameras], axis=1), meas, robust)
    return problem


def create_grid_ba_problem(gw, gh, pts_per_cell=6, seed=1, robust=None, outlier_frac=0.0, outlier_sigma=0.0, noise=0.0):
    """A bundle adjustment whose camera graph is a 2-D GRID (an aerial survey: gw x gh cameras in rows, every landmark seen by the 3 x 3 block of cameras
    around its cell, clipped at the border) -- the reduced camera system is then neither a narrow band nor small: row-major numbering gives a half bandwidth of
    about (2 gw + 2) cameras, no ordering gives less than about gw.  test/optimizeba.jl:22-23 leaves visibility a free parameter; residual model, variable
    order (all cameras, then all landmarks) and camera-major cost order are the reference generator's (test/optimizeba.jl:4-35)."""
    rng = np.random.default_rng(seed)
    ncam = gw * gh
    cams = rng.standard_normal((ncam, 6)) + np.array([1.0, 0, 0, 0, 1.0, 0])
    cam, lm, npts = grid_visibility(gw, gh, pts_per_cell)
    pts = rng.random((npts, 3)) + np.array([-0.5, -0.5, 10.0])
    problem = NLLSProblem(); problem.addvariables(cams); problem.addvariables(pts)
    c, X = cams[cam - 1], pts[lm - 1]
    meas = np.stack([(c[:, 0:3] * X).sum(1), (c[:, 3:6] * X).sum(1)], axis=1)
    if noise > 0:
        meas += rng.standard_normal(meas.shape) * noise
    if outlier_frac > 0:
        bad = rng.random(meas.shape[0]) < outlier_frac
        meas[bad] += rng.standard_normal((int(bad.sum()), 2)) * outlier_sigma
    problem.addcosts(K.RES_BA_AFFINE, np.stack([cam, lm + ncam], axis=1), meas, robust)
    return problem


def shuffle_camera_labels(problem, ncameras, seed, first=1):
    """The same problem with the cameras' LABELS permuted (camera c is afterwards variable perm^-1[c]; its pose moves with the label, cost blocks are
    re-listed camera-major in the new labels, as test/optimizeba.jl:24-31 adds them).  test/optimizeba.jl:22 numbers neighbouring cameras consecutively --
    a property of that generator, not of bundle adjustment: the reference's solve does not depend on it (ldl_analyze orders the factorisation itself,
    src/linearsystem.jl:52,68), and neither may this path.  `first`: 1-based variable index of the first camera (2 behind an adaptive kernel variable).
    Works for the affine (6 doubles per camera) and the SO(3) (12 doubles) kinds, two- and three-slot residuals."""
    perm = np.random.default_rng(seed).permutation(ncameras)
    inv = np.empty(ncameras, np.int64); inv[perm] = np.arange(ncameras)
    off = problem.var_offsets; v = problem.variables
    st = int(off[first] - off[first - 1])
    a0 = int(off[first - 1])
    cams = v[a0: a0 + st * ncameras].reshape(ncameras, st).copy()
    v[a0: a0 + st * ncameras] = cams[perm].ravel()                                   # new camera j holds old camera perm[j]
    for g in problem.costs.values():
        vi, da = g.arrays()
        col = next(c for c in range(vi.shape[1]) if vi.shape[0] and first <= vi[0, c] < first + ncameras)
        vi2 = vi.copy(); vi2[:, col] = inv[vi[:, col] - first] + first
        order = np.lexsort(tuple(vi2[:, c] for c in range(vi2.shape[1] - 1, -1, -1) if c != col) + (vi2[:, col],))
        g.set_arrays(np.ascontiguousarray(vi2[order]), np.ascontiguousarray(da[order]))
    problem._gpu = None
    return problem


def perturb_ba_problem(problem, pointnoise, posenoise, seed=2):
    """test/optimizeba.jl:38-47."""
    rng = np.random.default_rng(seed)
    kk, dd, off = problem.var_kind, problem.var_dim, problem.var_offsets
    v = problem.variables
    is_pt = (kk == K.VAR_EUCLIDEAN) & (dd == 3)
    is_cam = (kk == K.VAR_EUCLIDEAN) & (dd == 6)
    for mask, dim, noise in ((is_pt, 3, pointnoise), (is_cam, 6, posenoise)):
        idx = np.nonzero(mask)[0]
        if idx.size and noise != 0:
            pos = (off[idx][:, None] + np.arange(dim)[None, :]).ravel()
            v[pos] += rng.standard_normal(pos.size) * noise
    is_pose = kk == K.VAR_POSE_SO3
    idx = np.nonzero(is_pose)[0]
    if idx.size and posenoise != 0:
        from .variables import so3_exp
        for i in idx:
            d = rng.standard_normal(6) * posenoise
            st = v[off[i]:off[i] + 12]
            R = st[:9].reshape(3, 3, order="F") @ so3_exp(d[:3])
            st[:9] = R.ravel(order="F"); st[9:] += d[3:]
    problem._gpu = None
    return problem


def create_curvefit_problem(n=10_000, seed=1, noise=0.01):
    """BASELINE config 2: n 1-dim residuals a*exp(b*t)+c*t+d - y over 4 scalar variables
    (4 blocks of dof 1 -> MultiVariateLSdense, src/linearsystem.jl:105-106,123)."""
    rng = np.random.default_rng(seed)
    truth = np.array([2.0, -1.5, 0.7, 0.3])
    t = rng.random(n) * 2.0
    y = truth[0] * np.exp(truth[1] * t) + truth[2] * t + truth[3] + rng.standard_normal(n) * noise
    problem = NLLSProblem()
    for v in (1.5, -1.0, 0.0, 0.0):
        problem.addvariable(v)
    problem.addcosts(K.RES_CURVE_EXP4, np.tile(np.array([1, 2, 3, 4]), (n, 1)), np.stack([t, y], axis=1))
    return problem, truth


def create_so3_ba_problem(ncameras, nlandmarks, propvisible, seed=1, adaptive=True, outlier_frac=0.1,
                          noise=1e-3, outlier_sigma=0.1, robust=None, visibility=None):
    """BASELINE config 5: pinhole cameras with SO(3) rotations (new kind), optionally with
    a ContaminatedGaussian adaptive kernel as variable #1 (src/residual.jl:46-47).  visibility: (camera, landmark) pairs
    to use instead of the reference generator's windows (e.g. grid_visibility)."""
    from .variables import so3_exp
    rng = np.random.default_rng(seed)
    problem = NLLSProblem()
    first = 1
    if adaptive:
        from .variables import contaminated_gaussian
        problem.addvariable(contaminated_gaussian(2 * noise, 0.5 * outlier_sigma, 0.5), K.VAR_CONTAMINATED_GAUSSIAN)
        first = 2
    poses = np.zeros((ncameras, 12))
    ang = np.linspace(-0.3, 0.3, ncameras)
    for i in range(ncameras):
        R = so3_exp(np.array([0.05 * rng.standard_normal(), ang[i], 0.05 * rng.standard_normal()]))
        poses[i, :9] = R.ravel(order="F")
        poses[i, 9:] = np.array([0.2 * rng.standard_normal(), 0.2 * rng.standard_normal(), 0.1 * rng.standard_normal()])
    pts = rng.random((nlandmarks, 3)) + np.array([-0.5, -0.5, 5.0])
    problem.addvariables(poses, K.VAR_POSE_SO3)
    problem.addvariables(pts)
    cam, lm = ba_visibility(ncameras, nlandmarks, propvisible) if visibility is None else visibility
    R = poses[cam - 1, :9].reshape(-1, 3, 3).transpose(0, 2, 1)      # col-major storage -> R[n, r, c]
    Y = np.einsum("nrc,nc->nr", R, pts[lm - 1]) + poses[cam - 1, 9:]
    meas = Y[:, :2] / Y[:, 2:3] + rng.standard_normal((cam.size, 2)) * noise
    bad = rng.random(cam.size) < outlier_frac
    meas[bad] += rng.standard_normal((int(bad.sum()), 2)) * outlier_sigma
    camv = cam + (first - 1)
    lmv = lm + ncameras + (first - 1)
    if adaptive:
        varind = np.stack([np.ones_like(camv), camv, lmv], axis=1)
        problem.addcosts(K.RES_BA_SO3_ADAPTIVE, varind, meas)
    else:
        problem.addcosts(K.RES_BA_SO3, np.stack([camv, lmv], axis=1), meas, robust)
    return problem
