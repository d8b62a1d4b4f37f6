#!/usr/bin/env python3
"""Why do device trajectories take more rejected LM trials than the oracle's near the noise floor (profiles/r03_stress.txt, seeds 2016 / 2020)?

One Levenberg-Marquardt loop (src/iterators.jl:139-172 + src/optimize.jl:124-171), everything evaluated by the CPU oracle -- costs, gradient, Hessian,
retraction, step quality -- EXCEPT the damped step x, which comes from one of
  oracle : the oracle's sparse LDL'                                        (the reference's solver)
  device : the device's Schur elimination + block cyclic reduction         (nlls_solve on the SAME variables)
  refined: the device's step + one step of iterative refinement            (x -= M^-1 ((H + lambda I) x + g), M^-1 = the oracle's factorisation)
so that the ONLY thing that differs between the runs is the accuracy of x.  Logged per trial: ||(H + lambda I) x + g|| / ||g|| (H, g: the oracle's),
accepted / rejected; per run: trials for the same number of iterations.  Usage (GPU box): python tools/step_residual.py 2016 2020 [...seeds]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import scipy.sparse as sp
import nllssolver_jl_amd as N
from nllssolver_jl_amd import synthetic, _capi
from oracle import oracle as O
from tests.helpers import oracle_problem


def bsm_to_csr(ols):
    """the symmetric H of the oracle's linear system as scipy CSR (from the BlockSparseMatrix layout, src/BlockSparseMatrix.jl:30-47)"""
    cp, rv, nz, bo = ols.bsm_index(); data = ols.data; nb = len(cp) - 1
    bo0 = bo - 1; n = ols.info.ndof
    bs = np.diff(np.r_[bo0, n])
    rows = np.repeat(np.arange(nb), np.diff(cp)); cols = rv - 1; offs = nz - 1
    I, J, V = [], [], []
    for (br, bc) in {(int(a), int(b)) for a, b in zip(bs[rows], bs[cols])}:
        m = (bs[rows] == br) & (bs[cols] == bc)
        r0, c0, o0 = bo0[rows[m]], bo0[cols[m]], offs[m]
        ii, jj = np.meshgrid(np.arange(br), np.arange(bc), indexing="ij")          # block is column-major br x bc
        idx = o0[:, None, None] + ii[None] + br * jj[None]
        I.append((r0[:, None, None] + ii[None]).ravel()); J.append((c0[:, None, None] + jj[None]).ravel()); V.append(data[idx].ravel())
    I, J, V = np.concatenate(I), np.concatenate(J), np.concatenate(V)
    L = sp.coo_matrix((V, (I, J)), shape=(n, n)).tocsr()
    D = sp.diags(L.diagonal())
    strict = sp.tril(L, -1)
    # diagonal blocks are stored full: keep their lower triangle once
    return (strict + strict.T + D).tocsr()


def run(mk, step_kind, iters):
    p = mk(); op = oracle_problem(p); bi = np.arange(1, p.nvariables + 1, dtype=np.uint64)
    ols = op.linear_system(bi)
    ctx = None
    if step_kind != "oracle":
        ctx = _capi.Context(); ctx.upload(p.var_kind, p.var_dim, bi, p.groups(), 0)
    bestcost = ols.costgradhess(); lam = 0.0; trials = 0; log = []
    for it in range(iters):
        if lam == 0.0:
            lam = ols.max_abs_diag() * 1e-6
        H = bsm_to_csr(ols); g = ols.b.copy(); gn = np.linalg.norm(g)
        if ctx is not None:
            ctx.set_variables(op.get_variables(O.VARS_CURRENT)); ctx.sweep_gradhess()
        mu = 2.0; cur = 0.0                 # (nlls_sweep_gradhess resets the device's damping)
        while True:
            trials += 1
            if step_kind == "oracle":
                assert ols.solve(lam) == 0; x = ols.x.copy()
            else:
                ctx.damp(lam - cur); cur = lam; x = ctx.solve(want_x=True).copy()
                if step_kind == "refined":
                    r = H @ x + lam * x + g
                    ols.b[:] = r; assert ols.solve(lam) == 0; x = x + ols.x; ols.b[:] = g      # ols.x = -(H + lam I)^-1 r
            res = np.linalg.norm(H @ x + lam * x + g) / gn
            op.update(ols, O.VARS_NEXT, O.VARS_CURRENT, step=x)
            c = op.cost(O.VARS_NEXT)
            ok = not (c > bestcost)
            log.append((it, lam, res, ok))
            if ok:
                xHx = float(x @ (H @ x)); gx = float(g @ x)
                q = (c - bestcost) / (0.5 * xHx + gx)
                lam *= (1.0 - (2.0 * q - 1.0) ** 3) if q < 0.983 else 0.1
                break
            lam *= mu; mu *= 2.0
        if c <= bestcost:
            bestcost = c
        op.set_variables(op.get_variables(O.VARS_NEXT), O.VARS_CURRENT)
        ols.costgradhess()
    if ctx is not None:
        ctx.close()
    return bestcost, trials, log


def main():
    seeds = [a if a == "c4" else int(a) for a in sys.argv[1:]] or [2016, 2020]
    iters = int(os.environ.get("ITERS", "40"))
    for seed in seeds:
        if seed == "c4":            # BASELINE config 4 (bench.py's workload): 20 iterations
            mk = lambda: synthetic.perturb_ba_problem(synthetic.create_ba_problem(1000, 100000, 0.01, seed=1, robust=N.HuberKernel(0.01), outlier_frac=0.05, outlier_sigma=0.05), 1e-3, 1e-3)
            print("config 4: 1000 cameras x 100000 points, Huber(0.01)", flush=True)
            for kind in ("oracle", "device", "refined"):
                best, trials, log = run(mk, kind, 20)
                res = np.array([l[2] for l in log]); rej = sum(1 for l in log if not l[3])
                print(f"  step from {kind:8s}: best cost {best:.12e}, {trials} trials for 20 iterations ({rej} rejected); relative residual of the step: median {np.median(res):.2e}, max {res.max():.2e}", flush=True)
                print("     per trial (iteration, lambda, residual, accepted): " + " ".join(f"({l[0]},{l[1]:.1e},{l[2]:.1e},{int(l[3])})" for l in log), flush=True)
            continue
        rng = np.random.default_rng(seed)
        ncam = int(rng.integers(5, 120)); npts = int(rng.integers(50, 3000)); prop = max(float(rng.uniform(0.04, 0.5)), 4.0 / ncam)
        robust = bool(rng.integers(0, 2))
        kw = dict(robust=N.HuberKernel(float(rng.uniform(0.01, 0.05))), outlier_frac=0.1, outlier_sigma=0.2) if robust else {}
        mk = lambda: synthetic.perturb_ba_problem(synthetic.create_ba_problem(ncam, npts, prop, seed=seed, **kw), 1e-3, 1e-3)
        print(f"seed {seed}: {ncam} cameras x {npts} points, prop {prop:.3f}, robust {robust}", flush=True)
        for kind in ("oracle", "device", "refined"):
            best, trials, log = run(mk, kind, iters)
            res = np.array([l[2] for l in log]); rej = sum(1 for l in log if not l[3])
            late = np.array([l[2] for l in log if l[0] >= iters // 2])
            print(f"  step from {kind:8s}: best cost {best:.12e}, {trials} trials for {iters} iterations ({rej} rejected); relative residual of the step: "
                  f"median {np.median(res):.2e}, max {res.max():.2e}, second half median {np.median(late):.2e}", flush=True)


if __name__ == "__main__":
    main()
