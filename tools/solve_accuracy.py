#!/usr/bin/env python3
"""Backward error of the device's damped solve on ITS OWN system (A.data, b fetched from the device): || (H + lambda I) x + g || / || g || for a ladder of
dampings, next to (b) the same Schur elimination done in numpy (block-wise inverses of the point blocks, LAPACK on the dense reduced system) and (c) a sparse
LU of the whole system (scipy splu) -- is the device's error the price of forming a Schur complement at all, or of HOW it is solved?
Usage (GPU box): python tools/solve_accuracy.py [ncam npts prop]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla
import nllssolver_jl_amd as N
from nllssolver_jl_amd import synthetic, _capi
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from step_residual import bsm_to_csr


class _View:        # what bsm_to_csr reads, filled from the device
    def __init__(self, ctx):
        self._idx = ctx.bsm_index(); self.data = ctx.get_bsm_data(); self.info = ctx.info
    def bsm_index(self):
        return self._idx


def main():
    ncam, npts, prop = (int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3])) if len(sys.argv) > 3 else (100, 10000, 0.1)
    flags = int(os.environ.get("FLAGS", "0"))
    p = synthetic.perturb_ba_problem(synthetic.create_ba_problem(ncam, npts, prop, seed=1, robust=N.HuberKernel(0.01), outlier_frac=0.05, outlier_sigma=0.05), 1e-3, 1e-3)
    # a few LM iterations first: the interesting regime is near the optimum, where the dampings are small
    N.optimize(p, N.NLLSOptions(maxiters=8))
    ctx = _capi.Context(); ctx.upload(p.var_kind, p.var_dim, np.arange(1, p.nvariables + 1, dtype=np.uint64), p.groups(), flags)
    ctx.set_variables(p.variables); ctx.sweep_gradhess()
    H = bsm_to_csr(_View(ctx)); g = ctx.get_grad(); gn = np.linalg.norm(g); md = ctx.max_abs_diag()
    nc = 6 * ncam; n = H.shape[0]
    B = H[:nc, :nc].toarray(); E = H[nc:, :nc].tocsr(); C = H[nc:, nc:].tocsr()
    Cb = np.stack([C[3 * i:3 * i + 3, 3 * i:3 * i + 3].toarray() for i in range(min(npts, 20000))]) if npts <= 20000 else None
    print(f"{ncam} cameras x {npts} points: ndof {n}, max |H_ii| {md:.3e}, solve_mode {ctx.info.solve_mode}, bandwidth {ctx.info.bandwidth}, ||g|| {gn:.3e}", flush=True)
    cur = 0.0
    for rel in (1e-6, 1e-8, 1e-10, 1e-12, 1e-14, 1e-16, 1e-18):
        lam = md * rel
        ctx.damp(lam - cur); cur = lam
        try:
            x = ctx.solve(want_x=True).copy()
        except _capi.NllsError as e:
            print(f"  lambda/max|H_ii| {rel:.0e}: device solve failed ({e})"); continue
        r_dev = np.linalg.norm(H @ x + lam * x + g) / gn
        line = f"  lambda/max|H_ii| {rel:.0e}: device {r_dev:.2e} (||x|| {np.linalg.norm(x):.2e})"
        if Cb is not None:
            Ci = np.linalg.inv(Cb + lam * np.eye(3)[None])
            Cinv = sp.block_diag(list(Ci), format="csr") if npts <= 2000 else sp.bsr_matrix((Ci, np.arange(npts), np.arange(npts + 1)), shape=(3 * npts, 3 * npts)).tocsr()
            EtCi = (E.T @ Cinv).tocsr()
            S = B + lam * np.eye(nc) - (EtCi @ E).toarray(); s = g[:nc] - EtCi @ g[nc:]
            y = -np.linalg.solve(S, s); xp = -(Cinv @ (g[nc:] + E @ y)); xs = np.r_[y, xp]
            line += f", numpy Schur + LAPACK {np.linalg.norm(H @ xs + lam * xs + g) / gn:.2e}"
            try:
                cS = np.linalg.cholesky(S); y2 = -np.linalg.solve(cS.T, np.linalg.solve(cS, s)); xp2 = -(Cinv @ (g[nc:] + E @ y2)); xs2 = np.r_[y2, xp2]
                line += f" (Cholesky of S: {np.linalg.norm(H @ xs2 + lam * xs2 + g) / gn:.2e}; cond(S) {np.linalg.cond(S):.1e})"
            except np.linalg.LinAlgError:
                line += " (S is not numerically positive definite)"
        if n <= 400000:
            lu = spla.splu((H + lam * sp.identity(n)).tocsc()); xl = -lu.solve(g)
            line += f", sparse LU {np.linalg.norm(H @ xl + lam * xl + g) / gn:.2e}"
        print(line, flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
