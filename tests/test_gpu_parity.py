"""GPU parity tests proper: the HIP path, called through the C ABI (ctypes), against the CPU oracle on the
same seeded inputs.  Tolerances (fp64): the GPU sums in a different order than the sequential reference
(SURVEY.md "fp64 reduction order"), so values agree to ~1e-12 relative, not bit for bit; index arrays
(the BlockSparseMatrix layout) must agree exactly."""
import os

import numpy as np
import pytest

import nllssolver_jl_amd as N
from nllssolver_jl_amd import kinds as K
from nllssolver_jl_amd import synthetic, _capi
from nllssolver_jl_amd.variables import contaminated_gaussian
from oracle import oracle as O
from tests.helpers import oracle_problem, blockindices

pytestmark = pytest.mark.gpu

RTOL = 1e-11      # accumulate / cost sweeps
RTOL_X = 1e-7     # solve: conditioning of the damped normal equations enters


def rel(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300)


def check_problem(problem, unfixed=None, flags=0, lam_scale=1e-6, expect_sparse=None, expect_schur=None):
    bi = blockindices(problem, unfixed)
    op = oracle_problem(problem)
    ols = op.linear_system(bi, flags & _capi.FLAG_FORCE_SPARSE)
    ctx = _capi.Context()
    info = ctx.upload(problem.var_kind, problem.var_dim, bi, problem.groups(), flags)
    # ---- structure: exact
    assert info.is_sparse == ols.info.is_sparse and info.ndof == ols.info.ndof and info.nnz_data == ols.info.nnz_data
    if expect_sparse is not None:
        assert info.is_sparse == expect_sparse
    if expect_schur is not None:
        assert info.has_schur == expect_schur
    if info.is_sparse:
        for g, o in zip(ctx.bsm_index(), ols.bsm_index()):
            assert np.array_equal(g, o)
    # ---- sweeps
    ctx.set_variables(problem.variables)
    c_gpu = ctx.sweep_gradhess()
    c_ora = ols.costgradhess()
    assert np.isclose(c_gpu, c_ora, rtol=RTOL, atol=1e-300)
    A_gpu, b_gpu = ctx.get_bsm_data(), ctx.get_grad()
    A_ora = ols.data.copy()
    if not info.is_sparse:   # the device mirrors the lower triangle at the end of the sweep (gethessian)
        n = info.ndof; M = A_ora.reshape(n, n).T; M = np.tril(M) + np.tril(M, -1).T; A_ora = M.T.ravel()
    assert rel(A_gpu, A_ora) < RTOL, "A.data mismatch"
    assert rel(b_gpu, ols.b) < RTOL, "b mismatch"
    assert np.isclose(ctx.sweep_cost(), op.cost(), rtol=RTOL, atol=1e-300)
    assert ctx.sweep_cost() == ctx.sweep_cost()                      # run-to-run deterministic (fixed reduction tree)
    assert np.isclose(ctx.max_abs_diag(), ols.max_abs_diag(), rtol=1e-13)
    # ---- damped solve, quadratic form, retraction
    lam = ols.max_abs_diag() * lam_scale
    ctx.damp(lam)
    x_gpu = ctx.solve(want_x=True)
    assert ols.solve(lam) == 0
    assert rel(x_gpu, ols.x) < RTOL_X, f"x mismatch {rel(x_gpu, ols.x)}"
    xHx, gx = ctx.quadform()
    assert np.isclose(xHx, ols.quadform(x_gpu, lam), rtol=1e-9)
    assert np.isclose(gx, float(ols.b @ x_gpu), rtol=1e-9)
    assert np.isclose(ctx.step_maxabs(), np.max(np.abs(x_gpu)), rtol=1e-14)
    assert np.isclose(ctx.step_norm(), np.linalg.norm(x_gpu), rtol=1e-12)
    ctx.retract(_capi.VARS_NEXT, _capi.VARS_CURRENT)
    op.set_variables(problem.variables, O.VARS_NEXT)     # varnext starts as a deepcopy (src/optimize.jl:80-82)
    op.update(ols, O.VARS_NEXT, O.VARS_CURRENT, step=x_gpu)
    v_gpu, v_ora = ctx.get_variables(_capi.VARS_NEXT), op.get_variables(O.VARS_NEXT)
    assert rel(v_gpu, v_ora) < 1e-13
    # ---- the same trial in ONE call (nlls_lm_trial: damp, solve, retract, cost, step statistics -- where the structure allows it without a launch of
    # their own for the retraction and the statistics): against the separate entry points above.  (Atomics: the step agrees to rounding, not bits.)
    c_next = ctx.sweep_cost(_capi.VARS_NEXT)
    ctx.set_variables(np.zeros_like(v_gpu), _capi.VARS_NEXT)                     # (whatever the trial leaves there must be its own work)
    c_trial = ctx.lm_trial(0.0)
    assert np.isclose(c_trial, c_next, rtol=1e-9, atol=1e-300), (c_trial, c_next)
    assert rel(ctx.get_variables(_capi.VARS_NEXT), v_gpu) < 1e-9
    xHx2, gx2 = ctx.quadform()
    assert np.isclose(xHx2, xHx, rtol=1e-8) and np.isclose(gx2, gx, rtol=1e-8), (xHx2, xHx, gx2, gx)
    assert np.isclose(ctx.step_maxabs(), np.max(np.abs(x_gpu)), rtol=1e-8)
    # ---- BOTH paths of the trial (round 6).  Where nlls_upload_structure found the matrix-free trial applicable, the call above WAS matrix-free (the cost blocks evaluated inside the
    # elimination and the back-substitution, A.data's eliminated rows never read): the same trial once more with NLLS_OPT_MATERIALIZE -- the round-5 kernels on the materialised A.data --
    # must give the same step, point, cost and statistics; and against the oracle's x the matrix-free step holds the tolerance of the materialised one.
    st = ctx.solve_stats(); x_trial = ctx.get_step()
    assert rel(x_trial, ols.x) < RTOL_X, f"trial step vs oracle {rel(x_trial, ols.x)}"
    if st["mf_trials"] > 0:
        v_mf = ctx.get_variables(_capi.VARS_NEXT)
        ctx.set_option(_capi.OPT_MATERIALIZE, 1)
        ctx.set_variables(np.zeros_like(v_gpu), _capi.VARS_NEXT)
        c_mat = ctx.lm_trial(0.0)
        assert ctx.solve_stats()["mf_trials"] == st["mf_trials"]                 # (this one was not matrix-free)
        assert np.isclose(c_trial, c_mat, rtol=1e-9, atol=1e-13 * abs(c_gpu)), (c_trial, c_mat)      # (a noise-free problem's trial cost is a cancellation: the tolerance of the comparison above)
        assert rel(x_trial, ctx.get_step()) < 1e-9 and rel(v_mf, ctx.get_variables(_capi.VARS_NEXT)) < 1e-11
        xHx3, gx3 = ctx.quadform()
        assert np.isclose(xHx3, xHx2, rtol=1e-9) and np.isclose(gx3, gx2, rtol=1e-9)
        ctx.set_option(_capi.OPT_MATERIALIZE, 0)
        # A.data and b on demand behind a matrix-free trial: what the first sweep left (the linearisation point has not moved)
        assert rel(ctx.get_bsm_data(), A_ora) < RTOL and rel(ctx.get_grad(), ols.b) < RTOL
    ctx.close()
    return info


@pytest.mark.parametrize("shared,robust", [(1, None), (3, None), (40, N.HuberKernel(0.05))])
def test_standalone_bounded_scalar_variables(shared, robust):
    """NLLS_RES_SCALE_MIX: ZeroToInfScalar and ZeroToOneScalar as variables of their own (src/variable.jl:18-32) -- dual seeding through
    update(), retraction, sweep, solve against the oracle; then the whole LM loop against the oracle's."""
    from tests.test_oracle_pins import scale_mix_problem
    p = scale_mix_problem(5 + shared, n=64, noise=1e-3, shared=shared, robust=robust)
    check_problem(p, lam_scale=1e-4)
    q = scale_mix_problem(5 + shared, n=64, noise=1e-3, shared=shared, robust=robust)
    op = oracle_problem(q); ro = op.optimize(iterator=1)
    rg = N.optimize(q)
    assert np.isclose(rg.bestcost, ro.bestcost, rtol=1e-9) and np.allclose(q.variables, op.get_variables(), rtol=1e-6)
    assert np.all(q.variables[0::2] > 0) and np.all((q.variables[1::2] > 0) & (q.variables[1::2] < 1))


def test_wide_visibility_keeps_the_schur_path():
    """Real BA visibility at BASELINE config 3's size (100 cameras x 10k points): 1 % of the points are seen by 40 cameras and one by all 100.
    The reference's solve takes any sparsity (src/linearsolver.jl:28-32, src/linearsystem.jl:91-124); here a wide point must cost ITS OWN
    supernode more (LDS-staged generic kernel, up to the 160 KB of a CU; no pair accumulators beyond 29 cameras) and nobody else anything:
    the problem stays on the Schur path (no NLLS_SUB_SCHUR_SHAPE retry to the full dense system), the reduced camera system -- no longer a
    narrow band -- is solved by the dense MFMA LDL', x matches the oracle and LM converges to the oracle's cost."""
    ncam, npts = 100, 10_000
    p = synthetic.create_ba_problem(ncam, npts, 0.1, seed=11, robust=N.HuberKernel(0.05), outlier_frac=0.02, outlier_sigma=0.05)
    rng = np.random.default_rng(4)
    wide = {int(l): 40 for l in rng.choice(np.arange(1, npts + 1), size=npts // 100, replace=False)}
    wide[int(rng.integers(1, npts + 1))] = ncam
    p = synthetic.perturb_ba_problem(synthetic.widen_visibility(p, ncam, wide), 1e-3, 1e-3)
    info = check_problem(p, expect_sparse=1, expect_schur=1, lam_scale=1e-4)
    assert info.nreduced_dof == 6 * ncam and info.solve_mode == 1                 # the camera system, dense
    q = synthetic.perturb_ba_problem(synthetic.widen_visibility(synthetic.create_ba_problem(ncam, npts, 0.1, seed=11, robust=N.HuberKernel(0.05), outlier_frac=0.02, outlier_sigma=0.05), ncam, wide), 1e-3, 1e-3)
    op = oracle_problem(q); ro = op.optimize(iterator=1, maxiters=8)
    rg = N.optimize(q, N.NLLSOptions(maxiters=8))
    assert np.isclose(rg.bestcost, ro.bestcost, rtol=1e-7), (rg.bestcost, ro.bestcost)


@pytest.mark.parametrize("nwide,kwide", [(2, 420), (1, 600)])
def test_landmarks_seen_by_hundreds_of_cameras_stay_out_of_the_eliminated_set(nwide, kwide):
    """A landmark seen by more cameras than the LDS-staged elimination can stage (~330 six-dof neighbours: 150 KB of a CU's LDS) used to take the WHOLE problem off
    the Schur path (NLLS_SUB_SCHUR_SHAPE: retry without elimination, i.e. the full system densely -- a decline at any real size).  Now such a block is simply not
    eliminated: it stays in the reduced system (a border block when it couples to a quarter of the cameras, else a hub of the tile-sparse solver / a block of the
    dense one), every other landmark is eliminated as before.  600 cameras, 4000 landmarks, one or two of them seen by 420 / by all cameras: structure, sweep,
    damped solve, retraction and five LM iterations against the oracle (whose LDL' takes any sparsity: src/linearsolver.jl:28-32)."""
    ncam, npts = 600, 4000
    def mk():
        p = synthetic.create_ba_problem(ncam, npts, 8.0 / ncam, seed=17, robust=N.HuberKernel(0.05), outlier_frac=0.02, outlier_sigma=0.05)
        wide = {int(l): kwide for l in np.random.default_rng(3).choice(np.arange(1, npts + 1), size=nwide, replace=False)}
        return synthetic.perturb_ba_problem(synthetic.widen_visibility(p, ncam, wide), 1e-3, 1e-3)
    p = mk()
    info = check_problem(p, expect_sparse=1, expect_schur=1, lam_scale=1e-4)
    assert info.nschur_blocks == npts - nwide and info.nreduced_dof == 6 * ncam + 3 * nwide, (info.nschur_blocks, info.nreduced_dof)
    op = oracle_problem(mk()); ro = op.optimize(iterator=1, maxiters=5)
    rg = N.optimize(p, N.NLLSOptions(maxiters=5))
    assert np.isclose(rg.bestcost, ro.bestcost, rtol=1e-7), (rg.bestcost, ro.bestcost)


def _upload_info(p, flags=0):
    ctx = _capi.Context()
    info = ctx.upload(p.var_kind, p.var_dim, np.arange(1, p.nvariables + 1, dtype=np.uint64), p.groups(), flags)
    st = ctx.solve_stats(); ctx.close()
    return info, st


@pytest.mark.parametrize("ncam,npts,prop,seed", [(40, 1500, 0.12, 31), (150, 6000, 0.04, 32), (300, 9000, 0.03, 33)])
@pytest.mark.parametrize("reorder", [1, 0])
def test_shuffled_camera_labels(ncam, npts, prop, seed, reorder):
    """The generator of test/optimizeba.jl gives every point a window of NEIGHBOURING cameras (a banded reduced system); real image collections do
    not number their cameras that way, and the reference does not care: ldl_analyze orders the factorisation itself (src/linearsystem.jl:52,68).
    The same problems with the cameras' labels shuffled.  Default: the reduced camera system is put in reverse Cuthill-McKee order at upload and
    stays on the band path, with a band no wider than 1.25 x the unshuffled problem's.  NLLS_FLAG_NO_REORDER: the caller's order -- no band, the
    dense reduced system (what round 3 did with every such problem).  Both against the oracle: sweep, solve, retraction, then the LM loop."""
    mk = lambda: synthetic.create_ba_problem(ncam, npts, prop, seed=seed, robust=N.HuberKernel(0.05), outlier_frac=0.05, outlier_sigma=0.05)
    info0, st0 = _upload_info(mk())
    assert info0.solve_mode == 2 and st0["reordered"] == 0                               # the generator's own numbering is already the narrowest
    p = synthetic.perturb_ba_problem(synthetic.shuffle_camera_labels(mk(), ncam, seed), 1e-3, 1e-3)
    flags = 0 if reorder else _capi.FLAG_NO_REORDER
    info = check_problem(p, expect_sparse=1, expect_schur=1, lam_scale=1e-4, flags=flags)
    _, st = _upload_info(p, flags)
    if reorder:
        assert info.solve_mode == 2 and info.bandwidth <= 1.25 * info0.bandwidth, (info.bandwidth, info0.bandwidth)
        assert st["reordered"] == 1 and st["bandwidth_caller_order"] > 3 * info0.bandwidth
    else:
        assert st["reordered"] == 0 and (info.solve_mode in (0, 1) or info.bandwidth > 3 * info0.bandwidth)
    if reorder:
        q_vars = p.variables.copy()
        op = oracle_problem(p); ro = op.optimize(iterator=1, maxiters=6)
        p.variables[:] = q_vars
        rg = N.optimize(p, N.NLLSOptions(maxiters=6))
        assert np.isclose(rg.bestcost, ro.bestcost, rtol=1e-7), (rg.bestcost, ro.bestcost)


@pytest.mark.parametrize("ncam,npts,k,seed", [(60, 2000, 4, 41), (200, 5000, 3, 42)])
def test_random_visibility_has_no_band(ncam, npts, k, seed):
    """Every point seen by k cameras drawn at random: the reduced camera graph is an expander -- no ordering gives a band.  Reverse Cuthill-McKee
    is tried and whatever is narrower is kept; the dense MFMA LDL' solves the reduced system; against the oracle."""
    rng = np.random.default_rng(seed)
    p = N.NLLSProblem()
    cams = rng.standard_normal((ncam, 6)) + np.array([1.0, 0, 0, 0, 1.0, 0]); pts = rng.random((npts, 3)) + np.array([-0.5, -0.5, 10.0])
    p.addvariables(cams); p.addvariables(pts)
    cam = np.concatenate([rng.choice(ncam, size=k, replace=False) for _ in range(npts)]) + 1; lm = np.repeat(np.arange(1, npts + 1), k)
    order = np.lexsort((lm, cam)); cam, lm = cam[order], lm[order]
    c, X = cams[cam - 1], pts[lm - 1]
    meas = np.stack([(c[:, 0:3] * X).sum(1), (c[:, 3:6] * X).sum(1)], axis=1) + rng.standard_normal((cam.size, 2)) * 1e-3
    p.addcosts(K.RES_BA_AFFINE, np.stack([cam, lm + ncam], axis=1), meas, N.HuberKernel(0.05))
    p = synthetic.perturb_ba_problem(p, 1e-3, 1e-3)
    info = check_problem(p, expect_sparse=1, expect_schur=1, lam_scale=1e-4)
    assert info.nreduced_dof == 6 * ncam and info.solve_mode == 1


@pytest.mark.parametrize("gw,gh,shuffle", [(16, 12, None), (24, 24, 5)])
def test_grid_camera_graph_windowed_dense_solve(gw, gh, shuffle):
    """A 2-D grid of cameras (every landmark seen by a 3 x 3 block): the reduced camera system is a WIDE band -- too wide for the band kernels (> 80 columns),
    far narrower than the system.  With the tile-sparse solver switched off (NLLS_FLAG_NO_TILE_SPARSE) the dense blocked LDL' works only inside the band of the
    re-ordered system and the border strip (`dense_window`); with shuffled camera labels the reverse Cuthill-McKee ordering of the upload has to find the band
    first.  Sweep, solve, retraction and LM against the oracle, whose sparse LDL' takes any structure (as the reference's does: src/linearsolver.jl:28-32)."""
    mk = lambda: synthetic.create_grid_ba_problem(gw, gh, 4, seed=3, robust=N.HuberKernel(0.05), outlier_frac=0.05, outlier_sigma=0.05, noise=1e-3)
    p = mk()
    if shuffle is not None:
        p = synthetic.shuffle_camera_labels(p, gw * gh, shuffle)
    p = synthetic.perturb_ba_problem(p, 1e-3, 1e-3)
    NTS = _capi.FLAG_NO_TILE_SPARSE
    info = check_problem(p, expect_sparse=1, expect_schur=1, lam_scale=1e-4, flags=NTS)
    _, st = _upload_info(p, NTS)
    assert info.solve_mode == 1 and info.nreduced_dof == 6 * gw * gh
    if 6 * gw * gh >= 1024:
        # (row-major numbering: 2 rows + 2 cameras; reverse Cuthill-McKee from a corner of a shuffled grid walks L-shaped shells: up to twice that)
        assert st["dense_window"] == 1 and 80 < info.bandwidth <= 6 * ((2 if shuffle is None else 4) * max(gw, gh) + 4), (st, info.bandwidth)
        info2 = check_problem(p, expect_sparse=1, expect_schur=1, lam_scale=1e-4, flags=_capi.FLAG_NO_BAND)           # the same system by the full dense LDL'
        assert info2.solve_mode == 1 and _upload_info(p, _capi.FLAG_NO_BAND)[1]["dense_window"] == 0
    if shuffle is not None:
        assert st["reordered"] == 1
    q_vars = p.variables.copy()
    op = oracle_problem(p); ro = op.optimize(iterator=1, maxiters=5)
    p.variables[:] = q_vars
    os.environ["NLLS_NO_TSPARSE"] = "1"
    try:
        rg = N.optimize(p, N.NLLSOptions(maxiters=5))
    finally:
        del os.environ["NLLS_NO_TSPARSE"]
    assert np.isclose(rg.bestcost, ro.bestcost, rtol=1e-7), (rg.bestcost, ro.bestcost)


@pytest.mark.parametrize("gw,gh,shuffle,kind", [(24, 24, None, "affine"), (24, 24, 5, "affine"), (31, 17, 9, "affine"), (20, 20, 3, "so3_adaptive"), (13, 40, None, "affine")])
def test_grid_camera_graph_tile_sparse_solve(gw, gh, shuffle, kind):
    """The same camera grids through the TILE-SPARSE reduced solver (solve_mode 3, nlls_tsp.hip): the reduced blocks ordered by nested dissection and packed into
    128-row tiles, the LDL' factored level by level of the tile elimination tree -- the counterpart of the reference's ldl_analyze + ldl_factorize
    (src/linearsystem.jl:52,68, src/linearsolver.jl:28-32), which take any sparsity and any numbering.  Labels shuffled or not, affine cameras or SO(3) poses
    under an adaptive kernel whose variable couples to every block (a BORDER of the reduced system: its tile is a neighbour of every tile): sweep, damped solve
    (x at rtol 1e-7), retraction and five LM iterations against the oracle; x also against the dense LDL' of the same system."""
    if kind == "affine":
        p = synthetic.create_grid_ba_problem(gw, gh, 4, seed=3, robust=N.HuberKernel(0.05), outlier_frac=0.05, outlier_sigma=0.05, noise=1e-3)
        first, cdof = 1, 6
    else:
        cam, lm, npts = synthetic.grid_visibility(gw, gh, 4)
        p = synthetic.create_so3_ba_problem(gw * gh, npts, 0.0, seed=3, adaptive=True, visibility=(cam, lm))
        first, cdof = 2, 6
    if shuffle is not None:
        p = synthetic.shuffle_camera_labels(p, gw * gh, shuffle, first=first)
    p = synthetic.perturb_ba_problem(p, 1e-3, 1e-3) if kind == "affine" else p
    info = check_problem(p, expect_sparse=1, expect_schur=1, lam_scale=1e-4)
    assert info.solve_mode == 3 and info.nreduced_dof == cdof * gw * gh + (3 if kind != "affine" else 0), (info.solve_mode, info.nreduced_dof)
    if kind != "affine":
        assert info.nborder_dof == 3
    # the same damped system by the dense LDL'
    bi = np.arange(1, p.nvariables + 1, dtype=np.uint64); xs = []
    for flags in (0, _capi.FLAG_NO_BAND):
        ctx = _capi.Context(); ctx.upload(p.var_kind, p.var_dim, bi, p.groups(), flags)
        ctx.set_variables(p.variables); ctx.sweep_gradhess(); ctx.damp(1e-4 * ctx.max_abs_diag()); ctx.solve(); xs.append(ctx.get_step().copy()); ctx.close()
    assert np.linalg.norm(xs[0] - xs[1]) <= 1e-10 * np.linalg.norm(xs[1])
    q_vars = p.variables.copy()
    op = oracle_problem(p); ro = op.optimize(iterator=1, maxiters=5)
    p.variables[:] = q_vars
    rg = N.optimize(p, N.NLLSOptions(maxiters=5))
    assert np.isclose(rg.bestcost, ro.bestcost, rtol=1e-7), (rg.bestcost, ro.bestcost)


@pytest.mark.parametrize("ncam,npts,nviews,loop,overview", [(700, 5000, 6, False, 0), (900, 5000, 5, True, 0), (800, 5000, 6, False, 5)])
def test_scattered_camera_graph_tile_sparse_solve(ncam, npts, nviews, loop, overview):
    """Cameras scattered over a square -- or along a closed ring: a loop closure, whose reduced system no ordering turns into a narrow band -- every landmark seen by
    the cameras nearest to it: a camera graph with no grid and no numbering to exploit.  The upload's nested dissection (breadth-first level structures from
    pseudo-peripheral nodes) must still find a shallow elimination tree; sweep, damped solve, retraction and five LM iterations against the oracle, whose LDL' orders
    itself (as the reference's: src/linearsystem.jl:52,68).  overview: that many cameras also see 30 % of ALL landmarks each -- hubs coupled to most other cameras, within
    two steps of which everything lies: two of them fit the border of the reduced system (15 dof), the others are ordered last by the symbolic phase (the root front)."""
    mk = lambda: synthetic.perturb_ba_problem(synthetic.create_scattered_ba_problem(ncam, npts, nviews, seed=5, robust=N.HuberKernel(0.05), outlier_frac=0.05, outlier_sigma=0.05,
                                                                                     noise=1e-3, loop=loop, overview=overview), 1e-3, 1e-3)
    p = mk()
    info = check_problem(p, expect_sparse=1, expect_schur=1, lam_scale=1e-4)
    _, st = _upload_info(p)
    assert info.solve_mode == 3 and info.nreduced_dof == 6 * ncam, (info.solve_mode, info.bandwidth)
    assert st["tsp_levels"] <= 0.5 * ((6 * ncam + 127) // 128), st                     # the dependent chain: levels of the tree, not tile columns
    op = oracle_problem(mk()); ro = op.optimize(iterator=1, maxiters=5)
    rg = N.optimize(p, N.NLLSOptions(maxiters=5))
    assert np.isclose(rg.bestcost, ro.bestcost, rtol=1e-7), (rg.bestcost, ro.bestcost)


def test_ba_sparse_small():          # test/optimizeba.jl:71 shape (10 x 50 @ 0.3 -> sparse path)
    p = synthetic.perturb_ba_problem(synthetic.create_ba_problem(10, 50, 0.3, seed=1), 1e-3, 1e-3)
    info = check_problem(p, expect_sparse=1, expect_schur=1)
    assert info.ndof == 210 and info.owner_path == 1


def test_ba_dense_small():           # test/optimizeba.jl:51 shape (3 x 5 -> 33 dof -> dense path)
    p = synthetic.perturb_ba_problem(synthetic.create_ba_problem(3, 5, 1.0, seed=1), 1e-3, 1e-3)
    check_problem(p, expect_sparse=0, expect_schur=0)


def test_ba_heavy_camera_rows():     # cameras with > 128 observations take the register-accumulate path
    p = synthetic.perturb_ba_problem(synthetic.create_ba_problem(40, 3000, 0.12, seed=3), 1e-3, 1e-3)
    info = check_problem(p, expect_sparse=1, expect_schur=1)
    assert info.nreduced_dof == 240


def test_ba_blocked_cholesky():      # reduced system larger than one 64-wide panel -> MFMA trailing updates
    p = synthetic.perturb_ba_problem(synthetic.create_ba_problem(60, 1500, 0.1, seed=4), 1e-3, 1e-3)
    info = check_problem(p, expect_sparse=1, expect_schur=1)
    assert info.nreduced_dof == 360


def test_ba_force_atomic_and_no_schur():
    p = synthetic.perturb_ba_problem(synthetic.create_ba_problem(12, 80, 0.3, seed=5), 1e-3, 1e-3)
    info = check_problem(p, flags=_capi.FLAG_FORCE_ATOMIC)
    assert info.owner_path == 0
    info = check_problem(p, flags=_capi.FLAG_NO_SCHUR, expect_schur=0)
    assert info.nreduced_dof == info.ndof


def test_ba_fixed_variables():       # varflags path: some cameras and points fixed (src/cost.jl:27-52)
    p = synthetic.perturb_ba_problem(synthetic.create_ba_problem(10, 60, 0.4, seed=6), 1e-3, 1e-3)
    unfixed = np.ones(p.nvariables, bool); unfixed[[0, 3, 15, 16, 40]] = False
    check_problem(p, unfixed=unfixed, expect_sparse=1)


def test_ba_huber():                 # BASELINE config 4 robustifier
    p = synthetic.create_ba_problem(10, 60, 0.4, seed=7, robust=N.HuberKernel(0.01), outlier_frac=0.2, outlier_sigma=0.1)
    p = synthetic.perturb_ba_problem(p, 1e-3, 1e-3)
    check_problem(p, expect_sparse=1)
    p2 = synthetic.create_ba_problem(10, 60, 0.4, seed=7, robust=N.Scaled(N.Huber2oKernel(0.01), 2.0), outlier_frac=0.2, outlier_sigma=0.1)
    check_problem(synthetic.perturb_ba_problem(p2, 1e-3, 1e-3), lam_scale=1e-2)
    p3 = synthetic.create_ba_problem(10, 60, 0.4, seed=7, robust=N.GemanMcclureKernel(0.05), outlier_frac=0.2, outlier_sigma=0.1)
    check_problem(synthetic.perturb_ba_problem(p3, 1e-3, 1e-3), lam_scale=1e-1)


@pytest.mark.parametrize("ncam,npts", [(3, 5), (8, 20), (12, 10)])
def test_indefinite_dense_system(ncam, npts):
    """test/linearsolve.jl:29-44 on the device: a symmetric system that is NOT positive definite must still be solved exactly (the
    reference falls from cholesky to qr, src/linearsolver.jl:20-26).  Such systems reach the dense path for real: the second-order
    term of a redescending kernel (GemanMcclure, src/robust.jl) makes H indefinite far from the optimum.  33 dof: the one-wave
    Cholesky -> pivoted LU; 108 / 102 dof: the blocked LDL' (no pivot sign required).  (A NON-symmetric system, test/linearsolve.jl:18-27,
    cannot cross this boundary: the sweeps only ever write the lower triangle of H -- the oracle pins that case on the CPU.)"""
    p = synthetic.perturb_ba_problem(synthetic.create_ba_problem(ncam, npts, 1.0, seed=2, robust=N.GemanMcclureKernel(0.01),
                                                                 outlier_frac=0.3, outlier_sigma=0.2), 1e-2, 1e-2)
    op = oracle_problem(p); ols = op.linear_system(blockindices(p)); ols.costgradhess()
    n = ols.info.ndof; M = ols.data.reshape(n, n).T; H = np.tril(M) + np.tril(M, -1).T
    ev = np.linalg.eigvalsh(H)
    assert ev[0] < -1e-3 * ev[-1] and ev[-1] > 0                       # genuinely indefinite
    info = check_problem(p, expect_sparse=0)
    assert info.ndof == 6 * ncam + 3 * npts


def test_rosenbrock_and_curvefit():  # dense 2-dof and 4-dof systems (BASELINE configs 1-2)
    p = N.NLLSProblem(); p.addvariable(-0.5); p.addvariable(2.5)
    p.addcosts(K.RES_ROSENBROCK_A, [[1]], [[1.0]], N.Scaled(N.Huber2oKernel(1.6), 1.0))
    p.addcosts(K.RES_ROSENBROCK_B, [[1, 2]], [[10.0]])
    check_problem(p, expect_sparse=0, lam_scale=1e-3)
    q = N.NLLSProblem(); q.addvariable([-0.5, 2.5]); q.addcosts(K.RES_ROSENBROCK_2D, [[1]], [[1.0, 10.0]])
    check_problem(q, expect_sparse=0, lam_scale=1e-3)
    c, _ = synthetic.create_curvefit_problem(10_000, seed=1)
    check_problem(c, expect_sparse=0)


def test_adaptive_mean():            # test/adaptivecost.jl shape: kernel variable + two means
    rng = np.random.default_rng(1)
    pts = np.concatenate([rng.standard_normal(800), rng.standard_normal(200) * 10.0])
    p = N.NLLSProblem()
    p.addvariable(contaminated_gaussian(0.5, 5.0, 0.6), K.VAR_CONTAMINATED_GAUSSIAN); p.addvariable(0.0); p.addvariable(0.0)
    vi = np.empty((2000, 2), np.int64); da = np.empty((2000, 1))
    vi[:, 0] = 1; vi[0::2, 1] = 2; vi[1::2, 1] = 3; da[0::2, 0] = pts - 1; da[1::2, 0] = pts + 1
    p.addcosts(K.RES_ADAPTIVE_MEAN, vi, da)
    check_problem(p, expect_sparse=0, lam_scale=1e-3)
    check_problem(p, unfixed=[False, True, True], lam_scale=1e-3)     # kernel fixed: robustifydcost path
    check_problem(p, unfixed=[True, False, True], lam_scale=1e-3)


def test_so3_ba():                   # BASELINE config 5 shapes (new kinds)
    p = synthetic.create_so3_ba_problem(8, 60, 0.5, seed=2, adaptive=False, robust=N.HuberKernel(0.05))
    check_problem(synthetic.perturb_ba_problem(p, 1e-3, 1e-3), expect_sparse=1, lam_scale=1e-4)
    q = synthetic.create_so3_ba_problem(8, 60, 0.5, seed=2, adaptive=True)
    check_problem(synthetic.perturb_ba_problem(q, 1e-3, 1e-3), expect_sparse=1, lam_scale=1e-4)
    unfixed = np.ones(q.nvariables, bool); unfixed[0] = False        # adaptive kernel held fixed
    check_problem(q, unfixed=unfixed, expect_sparse=1, lam_scale=1e-4)


def test_closed_forms_against_dual_numbers():
    """Round 5: the pinhole kinds bring their Jacobian in closed form (Res<KIND>::jac) and the adaptive kernel's second derivatives are closed forms
    (cg_robustifydkernel_closed); the statement through dual numbers (src/autodiff.jl:81-93,164-165: Dual<N> through update(), Dual2 over (kernel, cost)) stays
    in the library as the check.  nlls_check_analytic evaluates EVERY block of the problem both ways on the device: J to 1e-14, everything else to 1e-12 of the
    largest magnitude of the quantity in its block (the tolerance is stated here: fp64, two different orders of the same arithmetic)."""
    for adaptive, robust in ((True, None), (False, N.HuberKernel(0.05)), (False, N.GemanMcclureKernel(0.1))):
        p = synthetic.perturb_ba_problem(synthetic.create_so3_ba_problem(40, 2000, 0.2, seed=11, adaptive=adaptive, robust=robust), 1e-2, 1e-2)
        ctx = _capi.Context(0)
        ctx.upload(p.var_kind, p.var_dim, blockindices(p), p.groups())
        ctx.set_variables(p.variables)
        d = ctx.check_analytic(); ctx.close()
        assert d["J"] < 1e-14 and d["Jtr"] < 1e-13 and d["cost"] < 1e-13 and d["drho"] < 1e-12 and d["d2rho"] < 1e-11, d
        if adaptive:
            assert 0 < d["dkernel"] < 1e-11 and 0 < d["d2kernel"] < 1e-10, d      # (0 would mean the two paths are the same code)
    # kinds without a closed form go through dual numbers on both sides: identical
    q = synthetic.perturb_ba_problem(synthetic.create_ba_problem(30, 800, 0.2, seed=3, robust=N.HuberKernel(0.01)), 1e-3, 1e-3)
    ctx = _capi.Context(0); ctx.upload(q.var_kind, q.var_dim, blockindices(q), q.groups()); ctx.set_variables(q.variables)
    d = ctx.check_analytic(); ctx.close()
    assert d["J"] == 0 and d["Jtr"] == 0 and d["cost"] == 0, d


def test_ba_band_solver():           # narrow-band reduced system -> persistent-workgroup bordered-band LDL'
    p = synthetic.perturb_ba_problem(synthetic.create_ba_problem(120, 4000, 0.06, seed=8), 1e-3, 1e-3)
    info = check_problem(p, expect_sparse=1, expect_schur=1)
    assert info.solve_mode == 2 and info.nreduced_dof == 720 and 0 < info.bandwidth < 128
    info = check_problem(p, flags=_capi.FLAG_NO_BCR, expect_schur=1)       # round-1 chain kernels: twisted (two workgroups meet in the middle)
    assert info.solve_mode == 2
    info = check_problem(p, flags=_capi.FLAG_NO_BCR | _capi.FLAG_NO_TWIST, expect_schur=1)     # ... and one-sided
    assert info.solve_mode == 2
    info = check_problem(p, flags=_capi.FLAG_NO_BAND, expect_schur=1)      # same system through the dense MFMA path
    assert info.solve_mode == 1


def test_so3_adaptive_band_with_border():   # the kernel variable couples to every camera: ordered last as a border
    q = synthetic.create_so3_ba_problem(100, 3000, 0.08, seed=4, adaptive=True)
    info = check_problem(synthetic.perturb_ba_problem(q, 1e-3, 1e-3), expect_sparse=1, lam_scale=1e-4)
    assert info.solve_mode == 2 and info.nborder_dof == 3


def test_ba_wide_points_fast_elimination():   # points seen by 11-12 cameras: more E columns than lanes (two per lane)
    p = synthetic.perturb_ba_problem(synthetic.create_ba_problem(150, 3000, 0.075, seed=9), 1e-3, 1e-3)
    info = check_problem(p, expect_sparse=1, expect_schur=1)
    assert info.solve_mode == 2 and info.bandwidth >= 64


@pytest.mark.parametrize("ncam,npts,prop,seed", [
    (70, 1500, 0.05, 11),      # bandwidth ~ 3 cameras: NBW = 2, short band
    (97, 2500, 0.04, 12),      # reduced size not a multiple of 16
    (160, 4000, 0.03, 13),     # NBW = 2..3, both sides of the twisted factorisation several blocks long
    (240, 5000, 0.045, 14),    # ~11 cameras per point: NBW = 5, the widest band the blocked path takes
    (300, 6000, 0.012, 15),    # ~4 cameras per point, long band
])
def test_band_solver_shapes(ncam, npts, prop, seed):
    """The band solvers over bandwidths, lengths and remainders (block cyclic reduction, twisted and one-sided chain); x against the oracle."""
    p = synthetic.perturb_ba_problem(synthetic.create_ba_problem(ncam, npts, prop, seed=seed, robust=N.HuberKernel(0.01),
                                                                 outlier_frac=0.05, outlier_sigma=0.05), 1e-3, 1e-3)
    info = check_problem(p, expect_sparse=1, expect_schur=1)
    assert info.nreduced_dof == 6 * ncam
    if info.solve_mode == 2:
        check_problem(p, flags=_capi.FLAG_NO_BCR, expect_schur=1)
        check_problem(p, flags=_capi.FLAG_NO_BCR | _capi.FLAG_NO_TWIST, expect_schur=1)


@pytest.mark.parametrize("ncam", [11, 21, 22, 31, 32, 33, 43, 53, 64, 75])
def test_dense_backward_block_edges(ncam):
    """The dense reduced solve's one-launch backward substitution works in 128-column blocks over a system padded to a multiple of 64 (+ the
    right-hand side's row): reduced sizes just below / at / above a block edge (6 ncam = 66 .. 450: 126 | 132, 186 | 192 | 198, 258, 318, 384, 450),
    with an even and an odd number of 64-blocks; x against the oracle's sparse LDL', and against the one-launch-per-block substitution it replaced."""
    p = synthetic.perturb_ba_problem(synthetic.create_ba_problem(ncam, 40 * ncam, min(1.0, 8.0 / ncam), seed=900 + ncam, robust=N.HuberKernel(0.02),
                                                                 outlier_frac=0.05, outlier_sigma=0.05), 1e-3, 1e-3)
    info = check_problem(p, flags=_capi.FLAG_NO_BAND, expect_schur=1)
    assert info.nreduced_dof == 6 * ncam and info.solve_mode == 1


@pytest.mark.parametrize("seed", list(range(100, 140)))
def test_randomized_ba_against_oracle(seed):
    """Seeded random shapes: cameras, points, visibility, robustifier, outliers and fixed variables drawn per case --
    sweeps (exact structure, cost, A.data, b), damped solve, quadratic form and retraction against the oracle."""
    rng = np.random.default_rng(seed)
    ncam = int(rng.integers(4, 60)); npts = int(rng.integers(20, 1500)); prop = float(rng.uniform(0.05, 0.6))
    prop = max(prop, 3.5 / ncam)                                        # well posed: every point is seen by at least three cameras
    kind = int(rng.integers(0, 4))
    robust = [None, N.HuberKernel(float(rng.uniform(0.005, 0.1))), N.GemanMcclureKernel(float(rng.uniform(0.02, 0.2))),
              N.Scaled(N.Huber2oKernel(float(rng.uniform(0.005, 0.1))), float(rng.uniform(0.5, 3.0)))][kind]
    kw = dict(robust=robust, outlier_frac=float(rng.uniform(0.0, 0.3)), outlier_sigma=0.1) if robust is not None else {}
    p = synthetic.perturb_ba_problem(synthetic.create_ba_problem(ncam, npts, prop, seed=seed, **kw), 1e-3, 1e-3)
    unfixed = None
    if rng.random() < 0.5:                                              # fix a few cameras and points (varflags path)
        unfixed = np.ones(p.nvariables, bool)
        unfixed[rng.choice(p.nvariables, size=max(1, p.nvariables // 20), replace=False)] = False
    flags = [0, _capi.FLAG_NO_BCR, _capi.FLAG_FORCE_ATOMIC, _capi.FLAG_NO_BAND, _capi.FLAG_NO_BCR | _capi.FLAG_NO_TWIST, _capi.FLAG_FORCE_SPARSE][int(rng.integers(0, 6))]
    if seed % 3 == 0: flags |= _capi.FLAG_DETERMINISTIC
    check_problem(p, unfixed=unfixed, flags=flags, lam_scale=[1e-6, 1e-4, 1e-1, 1e-2][kind])


def test_reupload_on_one_context():
    """A context is re-used for different structures (Schur, full system, Schur again): nothing of an earlier upload may leak
    into the solve of a later one."""
    p1 = synthetic.perturb_ba_problem(synthetic.create_ba_problem(40, 900, 0.1, seed=31), 1e-3, 1e-3)
    p2 = synthetic.perturb_ba_problem(synthetic.create_ba_problem(12, 80, 0.3, seed=32), 1e-3, 1e-3)
    ctx = _capi.Context()
    for p, flags in ((p1, 0), (p2, _capi.FLAG_NO_SCHUR), (p1, 0), (p2, 0), (p1, _capi.FLAG_NO_BAND)):
        bi = np.arange(1, p.nvariables + 1, dtype=np.uint64)
        info = ctx.upload(p.var_kind, p.var_dim, bi, p.groups(), flags)
        ctx.set_variables(p.variables)
        ctx.sweep_gradhess(); lam = ctx.max_abs_diag() * 1e-4; ctx.damp(lam)
        x = ctx.solve(want_x=True)
        ols = oracle_problem(p).linear_system(bi, 0); ols.costgradhess(); assert ols.solve(lam) == 0
        assert rel(x, ols.x) < RTOL_X, (flags, rel(x, ols.x))
    ctx.close()


@pytest.mark.parametrize("seed", list(range(300, 316)))
def test_randomized_so3_against_oracle(seed):
    """BASELINE config 5 kinds (SO(3) poses, pinhole projection, optional ContaminatedGaussian kernel variable) over
    seeded random shapes -- border ordering, band / dense reduced systems, 6-dof camera blocks against 3-dof points."""
    rng = np.random.default_rng(seed)
    ncam = int(rng.integers(6, 120)); npts = int(rng.integers(60, 2500)); prop = max(float(rng.uniform(0.04, 0.5)), 4.0 / ncam)
    adaptive = bool(rng.integers(0, 2))
    robust = None if adaptive else [None, N.HuberKernel(0.05), N.GemanMcclureKernel(0.1)][int(rng.integers(0, 3))]
    p = synthetic.create_so3_ba_problem(ncam, npts, prop, seed=seed, adaptive=adaptive, robust=robust)
    p = synthetic.perturb_ba_problem(p, 1e-3, 1e-3)
    unfixed = None
    if rng.random() < 0.3:
        unfixed = np.ones(p.nvariables, bool); unfixed[rng.choice(p.nvariables, size=max(1, p.nvariables // 25), replace=False)] = False
    check_problem(p, unfixed=unfixed, expect_sparse=1, lam_scale=1e-4 if robust is None or adaptive else 1e-1)


@pytest.mark.parametrize("seed", list(range(600, 616)))
def test_randomized_band_shapes(seed):
    """Long camera chains with 3-11 cameras per point: the band solvers over seeded random lengths, bandwidths (1..5 tiles),
    block counts and remainders -- block cyclic reduction (default), twisted and one-sided chain -- against the oracle's sparse LDL'."""
    rng = np.random.default_rng(seed)
    ncam = int(rng.integers(50, 420)); cpp = float(rng.uniform(3.0, 11.0)); npts = int(rng.integers(10 * ncam, 30 * ncam))
    kw = dict(robust=N.HuberKernel(0.01), outlier_frac=0.05, outlier_sigma=0.05) if rng.random() < 0.5 else {}
    p = synthetic.perturb_ba_problem(synthetic.create_ba_problem(ncam, npts, cpp / ncam, seed=seed, **kw), 1e-3, 1e-3)
    info = check_problem(p, expect_sparse=1, expect_schur=1)
    assert info.nreduced_dof == 6 * ncam
    if info.solve_mode == 2 and rng.random() < 0.5:
        check_problem(p, flags=[_capi.FLAG_NO_BCR, _capi.FLAG_NO_BCR | _capi.FLAG_NO_TWIST][int(rng.integers(0, 2))], expect_schur=1)


@pytest.mark.parametrize("ncam,npts,cpp,adaptive,seed", [
    (48, 1200, 10.5, False, 901),    # 288 dof, 80-column blocks: four blocks (fewer cameras would all be ordered as border)
    (54, 1500, 10.5, False, 903),    # five blocks: both ends go at once
    (130, 3000, 10.5, False, 904),   # ten blocks, last one partly padding
    (333, 7000, 10.5, False, 905),   # 25 blocks
    (64, 1200, 2.6, False, 906),     # one tile per block (bandwidth < 16): 24 blocks
    (96, 2000, 5.2, False, 907),     # two tiles per block
    (150, 3000, 7.9, False, 908),    # three..four tiles per block
    (60, 1500, 10.5, True, 909),     # SO(3) cameras + kernel variable: border rows ride through every level
    (200, 5000, 8.0, True, 910),
])
def test_block_cyclic_reduction_shapes(ncam, npts, cpp, adaptive, seed):
    """The block cyclic reduction over block counts (even, odd, partly padded last block), tiles per block and border rows;
    x against the oracle's sparse LDL'."""
    if adaptive:
        p = synthetic.perturb_ba_problem(synthetic.create_so3_ba_problem(ncam, npts, cpp / ncam, seed=seed, adaptive=True), 1e-3, 1e-3)
        info = check_problem(p, expect_sparse=1, lam_scale=1e-4)
        assert info.nborder_dof > 0
    else:
        p = synthetic.perturb_ba_problem(synthetic.create_ba_problem(ncam, npts, cpp / ncam, seed=seed), 1e-3, 1e-3)
        info = check_problem(p, expect_sparse=1, expect_schur=1)
    assert info.solve_mode == 2


@pytest.mark.parametrize("ncam,npts,cpp,seed,block", [(130, 3000, 10.5, 921, 64), (333, 7000, 10.5, 922, 64), (130, 3000, 11.5, 923, 80), (150, 3000, 7.9, 924, None)])
def test_block_size_of_the_cyclic_reduction_comes_from_the_structure(ncam, npts, cpp, seed, block, monkeypatch):
    """Round 6: the blocks of the block cyclic reduction are the smallest multiple of 16 unknowns that keeps the band part of S block TRIDIAGONAL -- found from the coupled
    pairs themselves, not from the bandwidth (cameras of 6 unknowns that share points with ten neighbours: bandwidth 65, and no coupling crosses two blocks of 64; with eleven
    neighbours some do, and the blocks stay at 80).  x of both block sizes against the oracle (check_problem) and against each other."""
    p = synthetic.perturb_ba_problem(synthetic.create_ba_problem(ncam, npts, cpp / ncam, seed=seed), 1e-3, 1e-3)
    info = check_problem(p, expect_sparse=1, expect_schur=1)
    assert info.solve_mode == 2
    bi = np.arange(1, p.nvariables + 1, dtype=np.uint64); xs = {}
    for full in (0, 1):
        if full: monkeypatch.setenv("NLLS_BCR_NT_FULL", "1")
        ctx = _capi.Context(); ctx.upload(p.var_kind, p.var_dim, bi, p.groups(), 0)
        st = ctx.solve_stats(); bw = st["bandwidth"]
        if full: assert st["bcr_block"] == 16 * ((bw + 15) // 16)
        elif block is not None: assert st["bcr_block"] == block, (st["bcr_block"], bw)
        assert st["bcr_block"] <= 16 * ((bw + 15) // 16)
        ctx.set_variables(p.variables); ctx.sweep_gradhess(); ctx.damp(1e-4 * ctx.max_abs_diag())
        xs[full] = ctx.solve(want_x=True).copy(); ctx.close()
    monkeypatch.delenv("NLLS_BCR_NT_FULL")
    assert rel(xs[0], xs[1]) < 1e-10


@pytest.mark.parametrize("ncam,npts,cpp,seed", [(130, 3000, 10.5, 951), (96, 2000, 5.2, 952), (300, 9000, 10.5, 953)])
def test_deterministic_flag_is_bit_reproducible(ncam, npts, cpp, seed):
    """NLLS_FLAG_DETERMINISTIC: the reduced system is assembled from per-supernode slabs by an ordered gather (no atomics) and
    solved by block cyclic reduction (no atomics either): x against the oracle, and bit-identical from one solve to the next."""
    p = synthetic.perturb_ba_problem(synthetic.create_ba_problem(ncam, npts, cpp / ncam, seed=seed, robust=N.HuberKernel(0.01), outlier_frac=0.05, outlier_sigma=0.05), 1e-3, 1e-3)
    info = check_problem(p, flags=_capi.FLAG_DETERMINISTIC, expect_sparse=1, expect_schur=1)
    assert info.solve_mode == 2
    ctx = _capi.Context(); bi = np.arange(1, p.nvariables + 1, dtype=np.uint64)
    ctx.upload(p.var_kind, p.var_dim, bi, p.groups(), _capi.FLAG_DETERMINISTIC)
    ctx.set_variables(p.variables); ctx.sweep_gradhess(); ctx.damp(1e-4 * ctx.max_abs_diag())
    x0 = ctx.solve(want_x=True).copy()
    for _ in range(5):
        assert np.array_equal(ctx.solve(want_x=True), x0)
    ctx.close()


def test_deterministic_flag_with_reordered_cameras():
    """NLLS_FLAG_DETERMINISTIC on a problem whose reduced system is re-ordered at upload: a supernode's columns are in MEMORY order, the reduced
    order is a permutation of it -- shares above the diagonal of S are gathered transposed.  x against the oracle, bit-identical run to run."""
    ncam, npts = 120, 4000
    p = synthetic.create_ba_problem(ncam, npts, 8 / ncam, seed=77, robust=N.HuberKernel(0.01), outlier_frac=0.05, outlier_sigma=0.05)
    p = synthetic.perturb_ba_problem(synthetic.shuffle_camera_labels(p, ncam, 77), 1e-3, 1e-3)
    info = check_problem(p, flags=_capi.FLAG_DETERMINISTIC, expect_sparse=1, expect_schur=1)
    assert info.solve_mode == 2
    ctx = _capi.Context(); bi = np.arange(1, p.nvariables + 1, dtype=np.uint64)
    ctx.upload(p.var_kind, p.var_dim, bi, p.groups(), _capi.FLAG_DETERMINISTIC)
    assert ctx.solve_stats()["reordered"] == 1
    ctx.set_variables(p.variables); ctx.sweep_gradhess(); ctx.damp(1e-4 * ctx.max_abs_diag())
    x0 = ctx.solve(want_x=True).copy()
    for _ in range(5):
        assert np.array_equal(ctx.solve(want_x=True), x0)
    ctx.close()


@pytest.mark.parametrize("seed", list(range(700, 706)))
def test_randomized_dense_curvefit(seed):
    """BASELINE config 2 shape (scalar residuals over four scalar variables -> BlockDenseMatrix path) at seeded random sizes."""
    rng = np.random.default_rng(seed)
    c, _ = synthetic.create_curvefit_problem(int(rng.integers(50, 30_000)), seed=seed)
    check_problem(c, expect_sparse=0)


@pytest.mark.parametrize("seed", list(range(800, 810)))
def test_randomized_tiny_dense_ba(seed):
    """test/optimizeba.jl:51 shape (a handful of cameras and points, everything visible): the BlockDenseMatrix path and
    the one-wave / small dense solvers over seeded random tiny sizes, some variables fixed."""
    rng = np.random.default_rng(seed)
    ncam = int(rng.integers(2, 7)); npts = int(rng.integers(4, 30))
    p = synthetic.perturb_ba_problem(synthetic.create_ba_problem(ncam, npts, 1.0, seed=seed), 1e-3, 1e-3)
    unfixed = None
    if rng.random() < 0.5:
        unfixed = np.ones(p.nvariables, bool); unfixed[rng.choice(p.nvariables, size=int(rng.integers(1, 3)), replace=False)] = False
    check_problem(p, unfixed=unfixed, lam_scale=1e-4)


@pytest.mark.gpu
@pytest.mark.parametrize("env", [{"NLLS_TSP_NO_MASKS": "1"}, {"NLLS_TSP_CARRY": "0"}, {"NLLS_TSP_SCHEME": "1"}, {"NLLS_TSP_SCHEME": "2"}, {"NLLS_TSP_SCHEME": "3"}, {"NLLS_TSP_CAP": "1"},
                                 {"NLLS_TSP_CAP": "64"}, {"NLLS_TSP_QUAD_MAX": "0"}, {"NLLS_TSP_QUAD_MAX": "100000"}, {"NLLS_TSP_SLOTS": "64"}])
def test_tile_sparse_ab_switches_still_match_the_oracle(env, monkeypatch):
    """The tile-sparse solver's A/B switches (every tile product in full, no tails carried up, each of the three panel schemes at every level,
    update jobs never / always cut into atomic pieces, whole-tile / quarter-tile update jobs everywhere, a smaller chip) select launch shapes the default run of
    one problem does not reach: the same parity -- a shuffled 26 x 26 camera grid, sweep, damped solve, retraction against the oracle."""
    for k, v in env.items(): monkeypatch.setenv(k, v)
    p = synthetic.perturb_ba_problem(synthetic.shuffle_camera_labels(synthetic.create_grid_ba_problem(26, 26, 4, seed=8, robust=N.HuberKernel(0.05), outlier_frac=0.05, outlier_sigma=0.05,
                                                                                                       noise=1e-3), 676, 4), 1e-3, 1e-3)
    info = check_problem(p, expect_sparse=1, expect_schur=1, lam_scale=1e-4)
    assert info.solve_mode == 3
    _, st = _upload_info(p)                                   # (the switches are read at every upload: what is visible from outside must have moved)
    for k in env: monkeypatch.delenv(k)
    _, st0 = _upload_info(p)
    if "NLLS_TSP_CARRY" in env: assert st["tsp_tiles"] > st0["tsp_tiles"], (st, st0)
    if env.get("NLLS_TSP_SCHEME") == "3": assert st["tsp_launches"] > st0["tsp_launches"], (st, st0)
    if env.get("NLLS_TSP_SCHEME") == "1": assert st["tsp_launches"] < st0["tsp_launches"], (st, st0)


@pytest.mark.gpu
@pytest.mark.parametrize("tiny", ["0", "1", "nofinrole"])
def test_small_dense_system_switch(tiny, monkeypatch):
    """NLLS_TINY_DENSE (read by nlls_create): 1 (default) -- a dense system of fewer than 64 unknowns takes its own route (one image of [A | b] per sweep workgroup summed by
    one gathering launch, no atomics on HBM; the LM trial's damping, factorisation, step statistics and retraction in ONE single-wavefront launch); 0 -- the general dense
    kernels of rounds 1-4.  The same parity either way: a curve fit, a robustified one, and six free cameras over 5000 FIXED points (more variables than the trial launch
    retracts itself: the retraction in a launch of its own)."""
    if tiny == "nofinrole": monkeypatch.setenv("NLLS_TINY_FIN_ROLE", "0"); tiny = "1"    # (the trial's finishing reduction in a launch of its own instead of workgroup 0 of the look-ahead sweep)
    else: monkeypatch.setenv("NLLS_TINY_DENSE", tiny)
    c, _ = synthetic.create_curvefit_problem(3000, seed=5)
    check_problem(c, expect_sparse=0)
    p = synthetic.perturb_ba_problem(synthetic.create_ba_problem(6, 5000, 0.5, seed=9, robust=N.HuberKernel(0.05)), 1e-3, 1e-3)
    unfixed = np.zeros(p.nvariables, bool); unfixed[:6] = True
    info = check_problem(p, unfixed=unfixed, lam_scale=1e-4, expect_sparse=0)
    assert info.ndof == 36 and p.nvariables > 4096
    if tiny == "1":      # the cost is a fixed-order sum; a workgroup's image is built with LDS atomics (rounding-level run-to-run differences), its mirror is exact
        ctx = _capi.Context(); ctx.upload(p.var_kind, p.var_dim, blockindices(p, unfixed), p.groups(), 0); ctx.set_variables(p.variables)
        c0 = ctx.sweep_gradhess(); A0, b0 = ctx.get_bsm_data().copy(), ctx.get_grad().copy()
        for _ in range(3):
            assert ctx.sweep_gradhess() == c0 and rel(ctx.get_bsm_data(), A0) < 1e-14 and rel(ctx.get_grad(), b0) < 1e-13
        assert np.array_equal(A0.reshape(36, 36), A0.reshape(36, 36).T)
        ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("fold", ["0", "1"])
def test_folded_sweep_switch(fold, monkeypatch):
    """NLLS_SWEEP_FOLD (read at upload): 0 -- no group takes the folded accumulate sweep (a three-slot group then takes one launch per role, every block evaluated once per
    role: rounds 1-4); 1 -- every group that qualifies takes it, two-slot bundle adjustment included (default: three-slot groups only).  The same sums either way."""
    monkeypatch.setenv("NLLS_SWEEP_FOLD", fold)
    q = synthetic.perturb_ba_problem(synthetic.create_so3_ba_problem(60, 1500, 0.15, seed=41, adaptive=True), 1e-3, 1e-3)
    check_problem(q, lam_scale=1e-4)
    unfixed = np.ones(q.nvariables, bool); unfixed[0] = False        # the adaptive kernel's variable fixed: its heavy slot has no row at all
    check_problem(q, unfixed=unfixed, lam_scale=1e-4)
    p = synthetic.perturb_ba_problem(synthetic.create_ba_problem(60, 1500, 0.15, seed=42, robust=N.HuberKernel(0.02)), 1e-3, 1e-3)
    check_problem(p, lam_scale=1e-4)
    unfixed = np.ones(p.nvariables, bool); unfixed[3:40:5] = False; unfixed[100:900:7] = False     # fixed cameras and points: entries without a heavy row, rows without entries
    check_problem(p, unfixed=unfixed, lam_scale=1e-4)


@pytest.mark.gpu
@pytest.mark.parametrize("env", [{"NLLS_DENSE_T64": "1"}, {"NLLS_DENSE_T128_MIN": "1"}, {"NLLS_ELIM_TILED": "1"}, {"NLLS_BCR_CHROWS_SLOTS": "0"},
                                 {"NLLS_DENSE_STEP_BACKWARD": "1"}, {"NLLS_BCR_LEVEL_BACKWARD": "1"}, {"NLLS_ELIM_SPLIT": "1"}, {"NLLS_POST_SPLIT": "1"},
                                 {"NLLS_HEAVY_MAX_ENTRIES": "256"}, {"NLLS_SUPERNODE_PIECE": "128"}, {"NLLS_SUPERNODE_PIECE": "5"}])
def test_ab_switches_select_paths_that_still_match_the_oracle(env, monkeypatch):
    """The environment switches read by nlls_create (the register-tiled elimination instead of the matrix-core one, the
    64 x 64-tile dense update instead of the 128 x 128 one, the 128 x 128 one from the first pass on, three X rows per panel workgroup at every
    level of the block cyclic reduction, one backward launch per block / per level instead of the one-launch substitutions, the assembly in three
    launches, the trial's retraction and step statistics in a launch of their own, the runs of eliminated blocks never cut / cut into pieces of five members) select kernels or launch shapes no default run of this size reaches:
    the same parity as every other path -- band mode and the dense reduced solve (NLLS_FLAG_NO_BAND) of a camera chain, 2100 reduced dof."""
    for k, v in env.items(): monkeypatch.setenv(k, v)
    p = synthetic.perturb_ba_problem(synthetic.create_ba_problem(350, 7000, 10.0 / 350, seed=77, robust=N.HuberKernel(0.02)), 1e-3, 1e-3)
    for flags in (0, _capi.FLAG_NO_BAND):
        info = check_problem(p, flags=flags, lam_scale=1e-4)
        assert info.has_schur and info.nreduced_dof == 2100 and info.solve_mode == (1 if flags else 2)


def _ba(ncam, npts, prop, seed, robust=None):
    return synthetic.perturb_ba_problem(synthetic.create_ba_problem(ncam, npts, prop, seed=seed, robust=robust, outlier_frac=0.05 if robust else 0.0, outlier_sigma=0.05), 1e-3, 1e-3)


@pytest.mark.parametrize("ncam,npts,prop,robust", [(120, 3000, 0.06, None), (100, 10000, 0.1, "huber"), (200, 8000, 0.04, "huber2o"), (64, 2000, 0.1, "gm")])
def test_matrix_free_trial_against_the_materialised_one_and_the_oracle(ncam, npts, prop, robust):
    """nlls_mf.hip / nlls_mfb.hip (round 6): an LM trial that evaluates the cost blocks inside the Schur elimination and the back-substitution instead of reading the eliminated rows of
    A.data -- against the SAME trial through the materialised kernels (NLLS_OPT_MATERIALIZE on one upload) and against the oracle's full sparse LDL' (src/linearsolver.jl:28-32): step
    1e-7 (oracle) / 1e-9 (materialised), trial point, trial cost, x'Hx, g'x, max |x|.  The matrix-free assembly uses no atomics: two calls give the SAME BITS.  A rejected trial (more
    damping from the same point) and an accepted one (swap, next trial) go through the look-ahead of the reduced rows."""
    rk = {None: None, "huber": N.HuberKernel(0.01), "huber2o": N.Huber2oKernel(0.02), "gm": N.GemanMcclureKernel(0.05)}[robust]
    p = _ba(ncam, npts, prop, 11, rk)
    bi = np.arange(1, p.nvariables + 1, dtype=np.uint64)
    op = oracle_problem(p); ols = op.linear_system(bi); ols.costgradhess()
    ctx = _capi.Context(); info = ctx.upload(p.var_kind, p.var_dim, bi, p.groups())
    assert info.has_schur and info.solve_mode in (1, 2)
    ctx.set_variables(p.variables); c0 = ctx.sweep_gradhess()
    assert ctx.sweep_cost() == c0                                     # one fixed sum for every cost of the path
    lam = ols.max_abs_diag() * 1e-6
    res = {}
    for name, mat in (("mat", 1), ("mf", 0), ("mf2", 0)):
        ctx.set_option(_capi.OPT_MATERIALIZE, mat)
        if name != "mf2":                                             # (mf2: the SAME linearisation once more -- the accumulate sweep's LDS atomics are not bit-reproducible from sweep to sweep, the trial is)
            ctx.set_variables(p.variables); ctx.sweep_gradhess()
        n0 = ctx.solve_stats()["mf_trials"]
        ct = ctx.lm_trial(lam if name != "mf2" else 0.0)
        assert (ctx.solve_stats()["mf_trials"] - n0) == (0 if mat else 1), "the structure should qualify for the matrix-free trial"
        res[name] = (ct, ctx.get_step(), ctx.get_variables(_capi.VARS_NEXT), ctx.quadform(), ctx.step_maxabs(), ctx.step_norm())
    assert ols.solve(lam) == 0
    assert rel(res["mf"][1], ols.x) < RTOL_X and rel(res["mat"][1], ols.x) < RTOL_X
    assert rel(res["mf"][1], res["mat"][1]) < 1e-9 and rel(res["mf"][2], res["mat"][2]) < 1e-11
    assert np.isclose(res["mf"][0], res["mat"][0], rtol=1e-9, atol=1e-13 * abs(c0)) and np.allclose(res["mf"][3], res["mat"][3], rtol=1e-9) and np.isclose(res["mf"][4], res["mat"][4], rtol=1e-9)
    assert res["mf"][0] == res["mf2"][0] and np.array_equal(res["mf"][1], res["mf2"][1]) and np.array_equal(res["mf"][2], res["mf2"][2])      # bit-reproducible
    assert ctx.sweep_cost(_capi.VARS_NEXT) == res["mf2"][0]           # cost(trial point) == the trial's own cost, bit for bit
    # a rejected trial (ten times the damping from the same point), then the accepted path: swap, sweep(NULL), next trial -- against the same sequence materialised
    seq = {}
    for name, mat in (("mf", 0), ("mat", 1)):
        ctx.set_option(_capi.OPT_MATERIALIZE, mat)
        ctx.set_variables(p.variables); ctx.sweep_gradhess()
        a = ctx.lm_trial(lam); b = ctx.lm_trial(9 * lam)               # (damping accumulates: 10 lam)
        ctx.damp(-10 * lam); ctx.swap_variables(_capi.VARS_CURRENT, _capi.VARS_NEXT); ctx.sweep_gradhess(want_cost=False)
        c2 = ctx.lm_trial(lam)
        seq[name] = (a, b, c2, ctx.get_step())
    assert np.allclose(seq["mf"][:3], seq["mat"][:3], rtol=1e-9, atol=1e-13 * abs(c0)) and rel(seq["mf"][3], seq["mat"][3]) < 1e-8
    ctx.close()


def test_optimize_singles_invalidates_a_lookahead_sweep():
    """Advisor (round 5): nlls_optimize_singles rewrites CURRENT in place; a look-ahead sweep of that very set (enqueued behind the last trial) must not be taken for the
    linearisation at the relaxed point.  LM trial + accept + optimize_singles + sweep(NULL) + trial: the same with and without the look-ahead (NLLS_OPT_LOOKAHEAD)."""
    p = _ba(60, 1500, 0.12, 5, N.HuberKernel(0.02))
    bi = np.arange(1, p.nvariables + 1, dtype=np.uint64)
    out = {}
    for la in (1, 0):
        for mat in (0, 1):
            ctx = _capi.Context(); ctx.upload(p.var_kind, p.var_dim, bi, p.groups())
            ctx.set_option(_capi.OPT_LOOKAHEAD, la); ctx.set_option(_capi.OPT_MATERIALIZE, mat)
            ctx.set_variables(p.variables); ctx.sweep_gradhess(); lam = 1e-5 * ctx.max_abs_diag()
            ctx.copy_variables(_capi.VARS_NEXT, _capi.VARS_CURRENT)
            ctx.sweep_gradhess(want_cost=False); ctx.lm_trial(lam)            # (the second sweep of the set arms the look-ahead)
            ctx.damp(-lam); ctx.swap_variables(_capi.VARS_CURRENT, _capi.VARS_NEXT)
            pts = np.arange(61, 61 + 1500, dtype=np.int64)                       # every point on its own (src/optimize.jl:60-76)
            # the points' cost lists: block k of the one group touches point varind[k, 1]
            (g,) = p.groups(); vi = np.asarray(g["varind"]); order = np.argsort(vi[:, 1], kind="stable"); cnt = np.bincount(vi[:, 1] - 61, minlength=1500)
            cptr = np.concatenate([[0], np.cumsum(cnt)])
            ctx.optimize_singles(pts, cptr, np.zeros(len(order), np.int32), order.astype(np.int64), np.ones(len(order), np.int32), maxiters=3)
            ctx.sweep_gradhess(want_cost=False)
            c = ctx.lm_trial(lam)
            out[(la, mat)] = (c, ctx.get_step()); ctx.close()
    for mat in (0, 1):
        assert np.isclose(out[(1, mat)][0], out[(0, mat)][0], rtol=1e-10), (mat, out[(1, mat)][0], out[(0, mat)][0])
        assert rel(out[(1, mat)][1], out[(0, mat)][1]) < 1e-6            # (a stale linearisation shows at 1e-2; rounding through the damped gauge directions at 1e-7)
