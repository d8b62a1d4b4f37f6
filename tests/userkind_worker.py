"""Worker of tests/test_gpu_userkind.py (its own process: NLLS_AMD_LIB must be set before the library is loaded).  A library built with a USER header
(tests/user_kinds/radial_ba.hpp: `make user USER_KINDS=...`) runs two residual kinds the registry does not have -- with no oracle to hold them against (the oracle is the
reference's restatement: it has no such kinds), they are checked the way a user would check a new residual: the cost against numpy, the gradient against central differences
of the device's own cost, and a noise-free problem driven to the zero-residual optimum (the criterion of test/optimizeba.jl:62-75)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import nllssolver_jl_amd as N
from nllssolver_jl_amd import kinds as K, _capi

assert os.environ.get("NLLS_AMD_LIB"), "run through tests/test_gpu_userkind.py"
USER0, USER1 = 100, 101
K.register_user_kind(USER0, 2, 2, 2, ((K.VAR_EUCLIDEAN, 7), (K.VAR_EUCLIDEAN, 3)))
K.register_user_kind(USER1, 3, 2, 2, ((K.VAR_EUCLIDEAN, 1), (K.VAR_EUCLIDEAN, 6), (K.VAR_EUCLIDEAN, 3)))


def model(kind, vars_of, vi):
    if kind == USER0:
        c, X = vars_of(vi[:, 0], 7), vars_of(vi[:, 1], 3)
        u, v = (c[:, 0:3] * X).sum(1), (c[:, 3:6] * X).sum(1); s = 1.0 + c[:, 6] * (u * u + v * v)
        return np.stack([s * u, s * v], 1)
    f, c, X = vars_of(vi[:, 0], 1)[:, 0], vars_of(vi[:, 1], 6), vars_of(vi[:, 2], 3)
    return np.stack([f * (c[:, 0:3] * X).sum(1), f * (c[:, 3:6] * X).sum(1)], 1)


def make(kind, ncam, npts, percam, seed, robust=None):
    rng = np.random.default_rng(seed)
    p = N.NLLSProblem(); nshared = 0
    if kind == USER1:
        p.addvariable([1.3]); nshared = 1
    cams = []
    for i in range(ncam):
        c = rng.standard_normal(6) * 0.3 + np.array([1, 0, 0, 0, 1, 0.0])
        cams.append(p.addvariable(np.concatenate([c, [0.02 * rng.standard_normal()]]) if kind == USER0 else c))
    pts = [p.addvariable(rng.uniform(-0.5, 0.5, 3) + np.array([0, 0, 2.0])) for _ in range(npts)]
    vi = []
    for j in range(npts):                       # a band: every point seen by `percam` neighbouring cameras
        c0 = int(round(j * (ncam - percam) / max(npts - 1, 1)))
        for k in range(percam):
            vi.append(([1] if kind == USER1 else []) + [cams[c0 + k], pts[j]])
    vi = np.array(vi, np.int64)
    off = p.var_offsets; truth = p.variables.copy()
    vars_of = lambda idx, d, v=truth: v[off[idx - 1][:, None] + np.arange(d)]
    meas = model(kind, vars_of, vi)
    p.addcosts(kind, vi, meas, robust)
    return p, vi, meas, truth, nshared


def numpy_cost(kind, p, vi, meas, v):
    off = p.var_offsets
    r = model(kind, lambda idx, d: v[off[idx - 1][:, None] + np.arange(d)], vi) - meas
    return 0.5 * float((r * r).sum())


def check(kind):
    p, vi, meas, truth, nshared = make(kind, 40, 1500, 8, seed=kind)
    rng = np.random.default_rng(5)
    start = truth + 1e-3 * rng.standard_normal(truth.size)
    p.variables[:] = start
    unfixed = np.ones(p.nvariables, bool)
    ctx = _capi.Context(0)
    bi = np.zeros(p.nvariables, np.uint64); bi[unfixed] = np.arange(1, p.nvariables + 1, dtype=np.uint64)
    info = ctx.upload(p.var_kind, p.var_dim, bi, p.groups())
    assert info.is_sparse and info.has_schur, (info.is_sparse, info.has_schur)
    ctx.set_variables(start)
    c_dev = ctx.sweep_gradhess(); c_np = numpy_cost(kind, p, vi, meas, start)
    assert np.isclose(c_dev, c_np, rtol=1e-11), (c_dev, c_np)
    assert np.isclose(ctx.sweep_cost(), c_np, rtol=1e-11)
    b = ctx.get_grad()
    # gradient against central differences of the cost (numpy model: the device's cost equals it to 1e-11), 24 random dof incl. the shared focal length
    dofs = rng.choice(b.size, 24, replace=False); dofs[0] = 0
    for k in dofs:
        h = 1e-6; e = np.zeros_like(start); e[k] = h           # (all variables Euclidean: storage offset = dof offset)
        fd = (numpy_cost(kind, p, vi, meas, start + e) - numpy_cost(kind, p, vi, meas, start - e)) / (2 * h)
        assert abs(fd - b[k]) <= 1e-6 * max(1.0, np.max(np.abs(b))), (k, fd, b[k])
    # the damped step solves the device's own system: x'(H + lambda I)x = -g'x
    lam = 1e-6 * ctx.max_abs_diag(); ctx.damp(lam); x = ctx.solve(want_x=True); xHx, gx = ctx.quadform()
    assert abs(xHx + gx) <= 1e-8 * abs(gx), (xHx, gx)
    ctx.close()
    res = N.optimize(p, N.NLLSOptions(maxiters=60))
    assert res.bestcost < 1e-15 * len(vi), (kind, res.bestcost, res.niterations)
    # optimizesingles on the points (every other variable fixed at the truth): back to the zero-residual optimum
    q, vi2, meas2, truth2, _ = make(kind, 40, 1500, 8, seed=kind)
    v0 = truth2.copy(); ptsel = np.nonzero(q.var_dim == 3)[0] + 1
    off = q.var_offsets
    for j in ptsel: v0[off[j - 1]:off[j - 1] + 3] += 1e-2 * rng.standard_normal(3)
    q.variables[:] = v0
    it = N.optimizesingles(q, N.NLLSOptions(), ptsel)
    assert N.cost(q) < 1e-15 * len(vi2) and it.min() >= 1, (N.cost(q), it[:5])
    print(f"user kind {kind}: cost {c_dev:.6e} = numpy, gradient = central differences, optimize -> {res.bestcost:.2e} in {res.niterations} iterations, optimizesingles ok")


def check_five_slots():
    """USER2: five variables per cost block (more than any built-in kind): a quartic fitted through 4000 noisy samples -- the optimum is the linear least-squares solution"""
    USER2 = 102
    K.register_user_kind(USER2, 5, 1, 2, ((K.VAR_EUCLIDEAN, 1),) * 5)
    rng = np.random.default_rng(9); t = rng.uniform(-1, 1, 4000); coef = np.array([0.3, -1.2, 0.7, 2.0, -0.5])
    y = np.vander(t, 5, increasing=True) @ coef + 0.01 * rng.standard_normal(t.size)
    p = N.NLLSProblem(); idx = [p.addvariable([0.0]) for _ in range(5)]
    p.addcosts(USER2, np.tile(np.array(idx, np.int64), (t.size, 1)), np.stack([t, y], 1))
    res = N.optimize(p, N.NLLSOptions(maxiters=30))
    ref = np.linalg.lstsq(np.vander(t, 5, increasing=True), y, rcond=None)[0]
    assert np.allclose(p.variables, ref, rtol=1e-8, atol=1e-10), (p.variables, ref)
    print(f"user kind {USER2} (five slots): optimum = numpy lstsq, cost {res.bestcost:.6e}")


if __name__ == "__main__":
    check(USER0); check(USER1); check_five_slots()
    print("user kinds ok")
