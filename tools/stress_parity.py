#!/usr/bin/env python3
"""Stress run (not part of the test suite): the randomized parity checks of tests/test_gpu_parity.py over many more seeds, including larger
band shapes (more cameras, narrow visibility: the block-cyclic-reduction and matrix-core elimination paths).  Prints failures, exits 1 if any."""
import os, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import nllssolver_jl_amd as N
from nllssolver_jl_amd import synthetic, _capi
from tests.test_gpu_parity import check_problem

lo, hi = int(sys.argv[1]), int(sys.argv[2])
fails = 0
for seed in range(lo, hi):
    rng = np.random.default_rng(seed)
    try:
        mode = seed % 8           # (round 4: 5 = camera chains with SHUFFLED labels -- the reduced system is re-ordered at upload; 6 = 2-D camera grids, some shuffled: windowed dense /
                                  #  tile-sparse solve; 7 = scattered cameras, loop closures, overview cameras, larger grids: the tile-sparse solver (nested dissection at upload))
        if mode == 7:
            kind = int(rng.integers(0, 3)); robust = [None, N.HuberKernel(float(rng.uniform(0.01, 0.1))), N.GemanMcclureKernel(float(rng.uniform(0.05, 0.2)))][kind]
            shape = int(rng.integers(0, 4))
            if shape == 3:
                gw = int(rng.integers(20, 46)); gh = int(rng.integers(20, 46))
                p = synthetic.create_grid_ba_problem(gw, gh, int(rng.integers(2, 5)), seed=seed, robust=robust, outlier_frac=0.05 if robust else 0.0, outlier_sigma=0.05, noise=1e-3); ncam = gw * gh
            else:
                ncam = int(rng.integers(300, 1500)); npts = int(rng.integers(5, 9)) * ncam
                p = synthetic.create_scattered_ba_problem(ncam, npts, int(rng.integers(4, 8)), seed=seed, robust=robust, outlier_frac=0.05 if robust else 0.0, outlier_sigma=0.05, noise=1e-3,
                                                          loop=shape == 1, overview=int(rng.integers(1, 7)) if shape == 2 else 0)
            if rng.random() < 0.5: p = synthetic.shuffle_camera_labels(p, ncam, seed)
            p = synthetic.perturb_ba_problem(p, 1e-3, 1e-3)
            flags = [0, 0, 0, _capi.FLAG_NO_TILE_SPARSE, _capi.FLAG_FORCE_ATOMIC][int(rng.integers(0, 5))]
            info = check_problem(p, flags=flags, lam_scale=[1e-4, 1e-3, 1e-1][kind])
            modes_seen = globals().setdefault("modes_seen", {}); modes_seen[int(info.solve_mode)] = modes_seen.get(int(info.solve_mode), 0) + 1
            continue
        if mode == 3:            # SO(3) cameras, pinhole, optionally the adaptive kernel as a border variable (BASELINE config 5 kinds)
            ncam = int(rng.integers(6, 300)); npts = int(rng.integers(60, 6000)); prop = max(float(rng.uniform(0.02, 0.5)), 4.0 / ncam)
            adaptive = bool(rng.integers(0, 2))
            robust = None if adaptive else [None, N.HuberKernel(0.05), N.GemanMcclureKernel(0.1)][int(rng.integers(0, 3))]
            p = synthetic.perturb_ba_problem(synthetic.create_so3_ba_problem(ncam, npts, prop, seed=seed, adaptive=adaptive, robust=robust), 1e-3, 1e-3)
            unfixed = None
            if rng.random() < 0.3:
                unfixed = np.ones(p.nvariables, bool); unfixed[rng.choice(p.nvariables, size=max(1, p.nvariables // 25), replace=False)] = False
            check_problem(p, unfixed=unfixed, lam_scale=1e-4 if robust is None or adaptive else 1e-1)
            continue
        if mode == 6:
            gw = int(rng.integers(5, 34)); gh = int(rng.integers(5, 34)); ppc = int(rng.integers(2, 6))
            kind = int(rng.integers(0, 3)); robust = [None, N.HuberKernel(float(rng.uniform(0.01, 0.1))), N.GemanMcclureKernel(float(rng.uniform(0.05, 0.2)))][kind]
            p = synthetic.create_grid_ba_problem(gw, gh, ppc, seed=seed, robust=robust, outlier_frac=0.05 if robust else 0.0, outlier_sigma=0.05, noise=1e-3)
            if rng.random() < 0.5: p = synthetic.shuffle_camera_labels(p, gw * gh, seed)
            p = synthetic.perturb_ba_problem(p, 1e-3, 1e-3)
            flags = [0, 0, _capi.FLAG_NO_REORDER, _capi.FLAG_FORCE_ATOMIC, _capi.FLAG_NO_TILE_SPARSE][int(rng.integers(0, 5))]
            check_problem(p, flags=flags, lam_scale=[1e-4, 1e-3, 1e-1][kind])
            continue
        if mode == 0:            # generic small shapes, all flags
            ncam = int(rng.integers(4, 60)); npts = int(rng.integers(20, 1500)); prop = max(float(rng.uniform(0.05, 0.6)), 3.5 / ncam)
        elif mode == 1 or mode == 5:          # camera chains: band mode with several BCR levels (5: labels shuffled below)
            ncam = int(rng.integers(48, 400)); per = int(rng.integers(3, 12)); npts = int(rng.integers(10, 40)) * ncam; prop = per / ncam
        elif mode == 4:          # camera chains with a tail of widely seen landmarks: wide supernodes (generic LDS-budgeted elimination, both classes)
            ncam = int(rng.integers(48, 300)); per = int(rng.integers(3, 10)); npts = int(rng.integers(10, 30)) * ncam; prop = per / ncam
        else:                    # dense reduced systems of a few hundred dof
            ncam = int(rng.integers(12, 90)); npts = int(rng.integers(200, 3000)); prop = float(rng.uniform(0.3, 0.9))
        kind = int(rng.integers(0, 4))
        robust = [None, N.HuberKernel(float(rng.uniform(0.005, 0.1))), N.GemanMcclureKernel(float(rng.uniform(0.02, 0.2))),
                  N.Scaled(N.Huber2oKernel(float(rng.uniform(0.005, 0.1))), float(rng.uniform(0.5, 3.0)))][kind]
        kw = dict(robust=robust, outlier_frac=float(rng.uniform(0.0, 0.3)), outlier_sigma=0.1) if robust is not None else {}
        p = synthetic.create_ba_problem(ncam, npts, prop, seed=seed, **kw)
        if mode == 4:
            nw = int(rng.integers(1, 12))
            wide = {int(l): int(rng.integers(per + 2, ncam + 1)) if rng.random() < 0.8 else ncam for l in rng.choice(npts, size=nw, replace=False) + 1}
            p = synthetic.widen_visibility(p, ncam, wide)
        if mode == 5: p = synthetic.shuffle_camera_labels(p, ncam, seed)
        p = synthetic.perturb_ba_problem(p, 1e-3, 1e-3)
        unfixed = None
        if rng.random() < 0.4:
            unfixed = np.ones(p.nvariables, bool); unfixed[rng.choice(p.nvariables, size=max(1, p.nvariables // 20), replace=False)] = False
        flags = [0, 0, _capi.FLAG_NO_BCR, _capi.FLAG_FORCE_ATOMIC, _capi.FLAG_NO_BAND, _capi.FLAG_DETERMINISTIC][int(rng.integers(0, 6))]
        check_problem(p, unfixed=unfixed, flags=flags, lam_scale=[1e-6, 1e-4, 1e-1, 1e-2][kind])
    except Exception as e:
        fails += 1
        print(f"seed {seed} FAILED: {type(e).__name__}: {str(e)[:200]}", flush=True)
    finally:
        pass
    if (seed - lo) % 20 == 19:
        print(f"... {seed - lo + 1} cases, {fails} failures", flush=True)
print(f"{hi - lo} cases, {fails} failures" + (f"; solve modes of the mode-7 cases: {dict(sorted(globals().get('modes_seen', {}).items()))}" if globals().get("modes_seen") else ""))
sys.exit(1 if fails else 0)
