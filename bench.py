#!/usr/bin/env python3
"""bench.py -- LM iterations/s (+ residual-blocks/s) on the synthetic bundle adjustment of BASELINE.json
(configs[3]: 1k cameras x 100k points x ~1M 2-dim reprojection residuals, Huber-robustified).

A "step" is one Levenberg-Marquardt outer iteration of the hot path: damped solve(s) + retraction + cost
sweep(s) (src/iterators.jl:139-172) followed by the gradient sweep that builds the next linear system
(src/optimize.jl:167-170).  Inputs are resident in HBM before the timed region starts.

  python bench.py --gpus N --steps K --warmup W
For N > 1 the driver launches this file under torch.distributed.run (one rank per GPU, RCCL).

Prints ONE JSON line on rank 0 with the contract fields plus "roofline" (accumulate sweep vs the HBM roof,
timed live with HIP events on the library's stream) and "cpu_baseline" (the CPU oracle -- a port, not the
Julia reference, which cannot run here -- timed on the host cores on a bounded sample).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_ACHIEVABLE_GBS = 6290.0    # MI355X_MICROARCH.md: the copy rate measured on the chip
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s spec, 6.29 TB/s measured copy)
FP64_MFMA_PEAK_TFLOPS = 78.6   # MI355X fp64 matrix = fp64 vector peak: one v_mfma_f64_16x16x4_f64 (2048 flop) per 64 cycles per SIMD
                               # (tools/microbench/mfma_rate.hip), 1024 SIMDs, 2.4 GHz


def sweep_code_hash():
    """Hash of the sources the accumulate kernel is built from: profiles/pmc_traffic.json carries the hash of the build whose
    counters it holds, and a stale file is refused (roofline.traffic = null) instead of being quoted."""
    import hashlib
    h = hashlib.sha256()
    for f in ("nlls_sweep.hip", "nlls_wave.hpp", "nlls_kinds.hpp", "nlls_structure.cpp"):      # (the MATERIALISING accumulate sweep: the kernel `roofline` describes)
        h.update(open(os.path.join(ROOT, "nllssolver.jl_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]

CONFIGS = {
    # name: (ncameras, npoints, propvisible)            BASELINE.json configs[2], configs[3]
    "ba_100x10k": (100, 10_000, 0.1),
    "ba_1kx100k": (1000, 100_000, 0.01),
    "ba_10kx1M": (10_000, 1_000_000, 0.001),            # 10x config 4 (not in BASELINE.json): 10M residual blocks, A.data 1.5 GB
    "curvefit_10k": None,                               # BASELINE.json configs[1]: 10k scalar residuals over 4 scalar variables (dense path)
    "ba_so3_500x50k": (500, 50_000, 0.02),              # BASELINE.json configs[4]: SO(3) cameras + adaptive kernel variable
    # camera GRIDS (not in BASELINE.json; test/optimizeba.jl:22-23 leaves visibility a free parameter): (grid width, grid height, landmarks per cell), every landmark
    # seen by the 3 x 3 block of cameras around its cell -- a reduced camera system that is neither a narrow band nor small: the tile-sparse solver's workload
    "ba_grid_40x40": (40, 40, 6),
    "ba_grid_100x100": (100, 100, 3),
}


def loop_code_hash():
    """Hash of everything an LM iteration's kernels are built from: profiles/pmc_iter.json (HBM bytes per LM iteration, both paths) is quoted only for the build it was collected on."""
    import hashlib
    h = hashlib.sha256()
    for f in sorted(os.listdir(os.path.join(ROOT, "nllssolver.jl_amd", "csrc"))):
        if f.endswith((".hip", ".hpp", ".cpp")):
            h.update(open(os.path.join(ROOT, "nllssolver.jl_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def algorithmic_bytes_per_sweep(nobs, var_storage, nnz_data, ndof, M=2, ndeps=2):
    """unique reads + unique writes of one gradient sweep."""
    return nobs * (8 * M + 8 * ndeps) + 8 * var_storage + 8 * (nnz_data + ndof)


def self_launch(n, deadline_s=None):
    """`python bench.py --gpus N` without a launcher: N fresh child processes of this file, one rank each (RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_* in their environment, exactly what torch.distributed.run would set), started BEFORE this process has touched the GPU or
    imported torch -- a process that has initialised the GPU must never exec or be replaced.  Rank 0's JSON line is relayed.  ALL children are
    watched: as soon as one exits non-zero (or the overall deadline passes) the others -- which would otherwise sit in a collective until the
    driver's limit -- are terminated, and this process exits non-zero."""
    import socket
    import subprocess
    import tempfile
    deadline_s = float(os.environ.get("NLLS_BENCH_DEADLINE_S", "540")) if deadline_s is None else deadline_s
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0)); port = so.getsockname()[1]
    out0 = tempfile.TemporaryFile()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, stdout=out0 if r == 0 else subprocess.DEVNULL))
    t_end = time.monotonic() + deadline_s
    why = None
    while True:
        rcs = [p.poll() for p in procs]
        bad = [(r, rc) for r, rc in enumerate(rcs) if rc not in (None, 0)]
        if bad:
            why = f"ranks failed (rank, exit code): {bad}"; break
        if all(rc == 0 for rc in rcs):
            break
        if time.monotonic() > t_end:
            why = f"no result within {deadline_s:.0f} s"; break
        time.sleep(0.05)
    if why:
        for p in procs:                       # (children only -- fresh processes this one started; never a signal by pattern)
            if p.poll() is None:
                p.terminate()
        t_kill = time.monotonic() + 5.0
        for p in procs:
            try:
                p.wait(timeout=max(0.1, t_kill - time.monotonic()))
            except subprocess.TimeoutExpired:
                p.kill(); p.wait()
        print(f"bench.py: {why}; the remaining ranks were terminated", file=sys.stderr)
        return 1
    out0.seek(0); sys.stdout.write(out0.read().decode()); sys.stdout.flush()
    return 0


def predicted_scaling(world):
    """The scaling model of DESIGN.md 5 for `bench.py --gpus N` on BASELINE config 4, printed WITH the measured line so that the first multi-GPU run falsifies it on the spot
    (no multi-GPU node has been available to this build in six rounds: every number here is a prediction from one-GPU measurements, not a measurement)."""
    # one LM iteration of the MATERIALISED trial -- the kernels of the collective route -- on one MI355X (round 6: one damped solve per iteration, 324 us), in us:
    # profiles/r06*_kernel_stats_materialised.csv
    divides = {"accumulate sweep": 44.0, "Schur elimination": 83.0, "back-substitution + retraction": 28.0, "cost sweep + step statistics": 14.0}
    replicated = {"block cyclic reduction of the reduced system (every rank)": 150.0, "convert + finish launches": 10.0, "host turn-around + launch gaps": 10.0}
    allreduce = {2: 40.0, 4: 50.0, 8: 60.0}.get(world, 40.0 + 10.0 * max(0, world.bit_length() - 2))      # [S | s], 3.4 MB, ring over point-to-point xGMI links: latency-dominated
    route = 9.0                                                                                            # pack / combine launches + the trial's 128-byte gather (measured with one rank: DESIGN.md 5)
    d, r = sum(divides.values()), sum(replicated.values())
    strong_us = r + d / world + allreduce + route
    weak_us = r + d + allreduce + route
    one = 1e6 / (r + d)
    return {"model": "t(N) = replicated + divides / N + all-reduce(N) + route  [us per LM iteration, one damped solve each]",
            "inputs_us": {"divides_by_N": divides, "replicated": replicated, "allreduce_S": allreduce, "collective_route": route},
            "strong": {"lm_iters_per_s": round(1e6 / strong_us, 1), "speedup_over_1_gpu": round((r + d) / strong_us, 3)},
            "weak": {"lm_iters_per_s": round(1e6 / weak_us, 1), "residual_blocks_per_s": round(world * 1e6 * 1e6 / weak_us, 1), "throughput_over_1_gpu": round(world * (r + d) / weak_us, 3)},
            "one_gpu_lm_iters_per_s": round(one, 1),
            "status": "PREDICTION from one-GPU measurements; strong scaling is bounded by the replicated reduced solve (Amdahl: at most %.2fx)" % ((r + d) / r)}


def oracle_loop_fixture(workload, steps):
    try:
        rec = json.load(open(os.path.join(ROOT, "tests", "golden", "oracle_trials.json")))[workload][f"iterations_{steps}"]
        return {"oracle_trials": rec["trials"], "oracle_final_cost": rec["final_cost"]}
    except Exception:
        return {"oracle_trials": None, "oracle_final_cost": None}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="ba_1kx100k", choices=sorted(CONFIGS))
    ap.add_argument("--path", default="default", choices=["default", "materialise"],
                    help="default: the matrix-free LM trial where the structure qualifies (two-slot Schur problems: the eliminated rows of A.data are never formed); "
                         "materialise: every trial eliminates from the materialised A.data (NLLS_OPT_MATERIALIZE), the path of rounds 1-5")
    ap.add_argument("--solver", default="default", choices=["default", "dense", "chain", "deterministic", "windowed", "nofloor"],
                    help="reduced-system solver: block cyclic reduction of the band (default), the dense MFMA LDL' (NLLS_FLAG_NO_BAND), "
                         "the round-1 twisted chain kernels (NLLS_FLAG_NO_BCR), or the atomics-free assembly (NLLS_FLAG_DETERMINISTIC)")
    ap.add_argument("--shuffle-cameras", type=int, default=None, metavar="SEED",
                    help="permute the cameras' labels (seeded) before the upload: the reference generator numbers neighbouring cameras consecutively, "
                         "real image collections do not -- the reduced camera system is then re-ordered at upload (reverse Cuthill-McKee)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-iters", type=int, default=10, help="LM iterations of the CPU oracle behind cpu_baseline (about 12 s of one core at BASELINE config 4)")
    ap.add_argument("--repeats", type=int, default=7, help="the (W warm-up + K timed) loop is run this many times from the same start; value = the MEDIAN run")
    ap.add_argument("--die-rank", type=int, default=-1, help=argparse.SUPPRESS)     # test hook: this rank exits non-zero in the middle of the timed loop
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus))
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    # stdout carries exactly ONE line, the JSON result: whatever libraries print there meanwhile (RCCL writes its version banner to stdout
    # at the first collective) is sent to stderr, and stdout is restored for the result
    sys.stdout.flush(); saved_stdout = os.dup(1); os.dup2(2, 1)
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import torch
    dist = None
    # NLLS_BENCH_BACKEND=gloo: rehearsal of the multi-rank path on ONE GPU (all ranks on device 0, reductions staged through
    # the host); the real run is one rank per GPU over RCCL
    backend = os.environ.get("NLLS_BENCH_BACKEND", "nccl")
    host_staged = backend != "nccl"
    if host_staged:
        local_rank = 0
    # NLLS_BENCH_FORCE_DIST=1: one rank, but through the sharded route (local phase / RCCL all-reduce / finish): what the
    # collectives and their synchronisations cost before any work is actually divided
    force_dist = world == 1 and os.environ.get("NLLS_BENCH_FORCE_DIST") == "1"
    if force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
    if world > 1 or force_dist:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        if host_staged:
            dist.init_process_group(backend, rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    import nllssolver_jl_amd as N
    from nllssolver_jl_amd import synthetic, iterators as It
    from nllssolver_jl_amd import optimizer as Opt
    from nllssolver_jl_amd._capi import VARS_CURRENT, VARS_NEXT
    from nllssolver_jl_amd.dist import ShardedLS

    from nllssolver_jl_amd import _capi
    if args.workload == "curvefit_10k":
        ncam, npts = 0, 0
        problem, _ = synthetic.create_curvefit_problem(10_000, seed=1)
        workload_desc = {"residuals": "a exp(b t) + c t + d - y, 10k scalar residuals over four scalar variables (BlockDenseMatrix path)"}
    elif args.workload == "ba_so3_500x50k":
        ncam, npts, prop = CONFIGS[args.workload]
        problem = synthetic.create_so3_ba_problem(ncam, npts, prop, seed=1, adaptive=True)
        if args.shuffle_cameras is not None:
            problem = synthetic.shuffle_camera_labels(problem, ncam, args.shuffle_cameras, first=2)
        problem = synthetic.perturb_ba_problem(problem, 1e-3, 1e-3)
        workload_desc = {"robust": "ContaminatedGaussian adaptive kernel (variable #1)", "outliers": "10% of measurements + N(0,0.1^2), seed 1", "cameras": "SO(3) poses, pinhole"}
    elif args.workload.startswith("ba_grid_"):
        gw, gh, ppc = CONFIGS[args.workload]
        ncam, npts = gw * gh, gw * gh * ppc
        problem = synthetic.create_grid_ba_problem(gw, gh, ppc, seed=1, robust=N.HuberKernel(0.05), outlier_frac=0.05, outlier_sigma=0.05, noise=1e-3)
        if args.shuffle_cameras is not None:
            problem = synthetic.shuffle_camera_labels(problem, ncam, args.shuffle_cameras)
        problem = synthetic.perturb_ba_problem(problem, 1e-3, 1e-3)
        workload_desc = {"robust": "Huber(0.05)", "outliers": "5% of measurements + N(0,0.05^2), seed 1", "visibility": f"{gw} x {gh} camera grid, {ppc} landmarks per cell, each seen by a 3 x 3 block of cameras"}
    else:
        ncam, npts, prop = CONFIGS[args.workload]
        problem = synthetic.create_ba_problem(ncam, npts, prop, seed=1, robust=N.HuberKernel(0.01),
                                              outlier_frac=0.05, outlier_sigma=0.05)
        if args.shuffle_cameras is not None:
            problem = synthetic.shuffle_camera_labels(problem, ncam, args.shuffle_cameras)
        problem = synthetic.perturb_ba_problem(problem, 1e-3, 1e-3)
        workload_desc = {"robust": "Huber(0.01)", "outliers": "5% of measurements + N(0,0.05^2), seed 1"}
    if args.shuffle_cameras is not None and ncam:
        workload_desc["camera_labels"] = f"permuted (seed {args.shuffle_cameras})"
    nobs = problem.ncosts()
    start_vars = problem.variables.copy()
    flags = {"default": 0, "dense": _capi.FLAG_NO_BAND, "chain": _capi.FLAG_NO_BCR, "deterministic": _capi.FLAG_DETERMINISTIC, "windowed": _capi.FLAG_NO_TILE_SPARSE, "nofloor": _capi.FLAG_NO_PIVOT_FLOOR}[args.solver]

    ls = ShardedLS(problem, np.ones(problem.nvariables, bool), flags=flags, device=local_rank, rank=rank, world=world, dist=dist, host_staged=host_staged, force_collectives=force_dist)
    info = ls.info
    if args.path == "materialise":
        ls.ctx.set_option(_capi.OPT_MATERIALIZE, 1)
    # never terminate early inside the timed region: exactly K outer iterations
    # (the adaptive-kernel costs are negative log-likelihoods: 'dcost < bestcost * reldcost', src/optimize.jl:152, then needs reldcost = +inf to stay off)
    options = N.NLLSOptions(maxiters=10 ** 9, reldcost=np.inf if args.workload == "ba_so3_500x50k" else -np.inf, absdcost=-np.inf, dstep=-1.0, maxfails=10 ** 9, maxtime=1e6)

    def fresh_loop(ls, problem, start_vars):
        ls.ctx.set_variables(start_vars, VARS_CURRENT)
        ls.ctx.copy_variables(VARS_NEXT, VARS_CURRENT)
        data = Opt.NLLSInternal(ls, time.perf_counter_ns())
        loop = Opt.OuterLoop(problem, options, data, It.LevMarData(), It.iterate_levmar, N.nullcallback)
        loop.start()
        return loop

    def sync():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def timed_loop(ls, problem, start_vars, profile=False):
        """W untimed + exactly K timed outer iterations from the same start point; barrier + device synchronisation on both sides,
        MAX over ranks"""
        loop = fresh_loop(ls, problem, start_vars)
        loop.iterations(args.warmup)
        loop = fresh_loop(ls, problem, start_vars)
        if profile:
            ls.ctx.profile_sweep(True)      # stamps around the accumulate launches of the timed loop itself (in-situ figure)
        sync()
        t0 = time.perf_counter()
        # (one GPU: the K iterations run in the library's own host loop, nlls_lm_iterations -- the statements of OuterLoop.iteration +
        # iterate_levmar in C++, no interpreter between two trials; sharded: the Python loop over the same entry points)
        loop.iterations(args.steps)
        assert loop.data.iternum == args.steps, (loop.data.iternum, args.steps)      # exactly K outer iterations
        sync()
        elapsed = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if host_staged else "cuda"); dist.all_reduce(t, op=dist.ReduceOp.MAX); elapsed = float(t.item())
        return loop, elapsed

    # R runs of (W untimed + K timed) iterations from the same start: one 10 ms loop is not a measurement -- which trials get rejected near the
    # optimiser's noise floor changes with the summation order of the atomics, and with it the number of trials in K iterations
    runs = []
    for rep in range(max(1, args.repeats)):
        if rank == args.die_rank and rep == max(1, args.repeats) // 2:
            os._exit(17)
        loop_r, elapsed_r = timed_loop(ls, problem, start_vars, profile=(rep == max(1, args.repeats) - 1))
        runs.append((elapsed_r, loop_r))
    stats_loop = ls.ctx.solve_stats()
    mf_loop = world == 1 and args.path == "default" and stats_loop.get("mf_trials", 0) > 0      # did the timed loop run matrix-free trials?
    # `roofline` describes the MATERIALISING accumulate launch (north_star names it).  Where the timed loop ran matrix-free it holds no such launch: one more loop of the same K
    # iterations with NLLS_OPT_MATERIALIZE set -- the round-5 path on the same upload -- gives the launch's in-situ duration, and the materialised rate to hold the default against
    materialised = None
    if mf_loop:
        ls.ctx.profile_sweep(False, read=True)
        ls.ctx.set_option(_capi.OPT_MATERIALIZE, 1)
        mruns = [timed_loop(ls, problem, start_vars, profile=(r == 2))[::-1] for r in range(3)]      # (elapsed, loop)
        me, ml_ = sorted(mruns, key=lambda t: t[0])[1]
        materialised = {"value": round(args.steps / me, 3), "unit": "LM iters/s", "ms_per_step": round(1e3 * me / args.steps, 4), "linear_solves": int(ml_.data.linearsolvers), "final_cost": ml_.data.bestcost,
                        "runs": [round(args.steps / e, 1) for e, _ in mruns], "what": "the same K iterations from the same start with NLLS_OPT_MATERIALIZE: accumulate sweep -> A.data, elimination and back-substitution read it (rounds 1-5)"}
    insitu = ls.ctx.profile_sweep(False, read=True)       # (avg, min, max ms, samples) of the accumulate launches inside the (last) timed loop: the kernel's own first-start .. last-end stamps
    insitu_disp = ls.ctx.profile_sweep_dispatch()         # ... the same launches by their dispatch timestamps (begin .. end as a kernel trace reports them)
    if mf_loop:
        ls.ctx.set_option(_capi.OPT_MATERIALIZE, 0)
    order = sorted(range(len(runs)), key=lambda i: runs[i][0])
    elapsed, loop = runs[order[len(order) // 2]]           # the median run is the one reported
    data = loop.data
    final_cost, start_cost = data.bestcost, data.startcost
    spread = {"runs": len(runs), "min": round(args.steps / max(e for e, _ in runs), 3), "max": round(args.steps / min(e for e, _ in runs), 3),
              "values": [round(args.steps / e, 1) for e, _ in runs], "trials_per_run": [int(l.data.linearsolvers) for _, l in runs],
              "lm_trials_per_s": [round(l.data.linearsolvers / e, 1) for e, l in runs]}
    trials_rates = sorted(l.data.linearsolvers / e for e, l in runs)
    trials_median = trials_rates[len(trials_rates) // 2]
    # where the optimiser reaches its noise floor (untimed, one iteration per call): the first iteration whose relative cost decrease is below 1e-12, and -- what
    # actually costs time -- the iterations that needed more than one trial (a rejected step, src/iterators.jl:160-171: near the floor lambda has shrunk to the
    # rounding level of the gauge-free reduced system and accept / reject is decided by the last bits of the cost; tools/step_residual.py)
    floor_at, first_reject, n_reject_iters, trials_trace = None, None, 0, []
    if world == 1:
        fl = fresh_loop(ls, problem, start_vars); prev = fl.data.bestcost; prev_solves = fl.data.linearsolvers
        for it in range(args.steps):
            fl.iterations(1); cur = fl.data.bestcost
            k = fl.data.linearsolvers - prev_solves; prev_solves = fl.data.linearsolvers; trials_trace.append(int(k))
            if k > 1:
                n_reject_iters += 1
                if first_reject is None:
                    first_reject = it + 1
            if floor_at is None and not (prev - cur > 1e-12 * abs(prev)):
                floor_at = it + 1
            prev = cur
    noise_floor_iterations = (args.steps - floor_at + 1) if floor_at else 0

    # ---- roofline of the accumulate sweep (dominant HBM-bound kernels), timed with HIP events on the library's stream
    reps = 20
    sweep_ms = ls.ctx.time_sweep_accumulate(reps)             # the accumulate launch(es) alone
    sweep_cost_ms = ls.ctx.time_sweep_gradhess(reps)          # ... plus the reduction of the cost partials
    cost_ms = ls.ctx.time_sweep_cost(reps)
    ls.ctx.damp(1e-3 * ls.ctx.max_abs_diag())
    solve_ms = ls.ctx.time_solve(5)
    reduced_ms = ls.ctx.time_reduced_solve(5) if world == 1 else 0.0
    solve_stats = ls.ctx.solve_stats()
    alg_bytes = algorithmic_bytes_per_sweep(ls.local_nobs, info.var_storage, ls.local_nnz_data, ls.local_ndof_written)
    if world == 1:      # per cost group: measurement + variable indices of every block (any residual kind)
        from nllssolver_jl_amd import kinds as K
        alg_bytes = sum(len(g) * 8 * (K.res_ndata(g.res_kind) + K.res_ndeps(g.res_kind)) for g in problem.costs.values()) \
            + 8 * info.var_storage + 8 * (ls.local_nnz_data + ls.local_ndof_written)
    achieved = alg_bytes / (sweep_ms * 1e-3) / 1e9
    traffic, traffic_note = None, None
    tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")     # filled from a separate rocprofv3 --pmc pass
    if os.path.exists(tpath) and world == 1:     # the counters were collected on the unsharded sweep
        try:
            rec = json.load(open(tpath)).get(args.workload, {})
            if rec.get("sweep_code_hash") == sweep_code_hash():
                traffic = rec.get("hbm_bytes_per_sweep")
            elif rec:
                traffic_note = "profiles/pmc_traffic.json was collected on a different build of the sweep sources: refused (re-run tools/measure_round.sh)"
        except Exception:
            traffic = None
    # the figure quoted as `frac` is the IN-SITU one BY DISPATCH TIMESTAMPS: begin .. end of the accumulate dispatch(es) as the command processor records them
    # (the launch carries its own start / stop events), averaged over the launches of the timed LM loop itself -- the duration rocprofv3 --kernel-trace
    # reports per dispatch, so the line can be checked against profiles/.  The kernel's own first-workgroup-start .. last-workgroup-end stamps (shorter by the
    # dispatch boundary) are kept beside it as `kernel_span`; `frac_cold` is the same launch with its data coming from HBM.
    span_ms = insitu[0] if insitu and insitu[3] >= 3 else None
    disp_ms = insitu_disp[0] if insitu_disp and insitu_disp[3] >= 3 and insitu_disp[0] > 0 else None
    quoted_ms = disp_ms or span_ms or sweep_ms
    achieved_q = alg_bytes / (quoted_ms * 1e-3) / 1e9
    roofline = {"bound": "hbm", "achieved": round(achieved_q, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved_q / HBM_PEAK_GBS, 4), "frac_cold": None, "traffic": traffic,
                "kernel": ("the accumulate launch(es) of one gradient sweep (two-slot bundle adjustment: gh_fused_kernel, point rows as light tiles + camera rows as heavy tiles; "
                           "three-slot kinds: gh_fold_kernel + gh_fold_gather_kernel, every block evaluated once)") if info.is_sparse else
                          "the gradient sweep of a dense system: gh_dense_kernel (one image of [A | b] per workgroup) + dense_tiny_gather_kernel, bracketed by events -- launch-bound, not HBM-bound (DESIGN.md 4.5)",
                "algorithmic_bytes_per_launch": int(alg_bytes), "ms_per_launch": round(quoted_ms, 4),
                "timing": ("in situ, dispatch timestamps: begin .. end of the accumulate dispatch(es) as the command processor records them (hipExtLaunchKernelGGL start / stop events), averaged over the launches "
                           "of the timed LM loop itself -- what rocprofv3 --kernel-trace reports per dispatch") if disp_ms else
                          ("in situ: execution span of the launch (kernel stamps)" if span_ms else "back to back (nlls_time_sweep_accumulate)"),
                "in_situ": None if not disp_ms else {"avg_ms": round(insitu_disp[0], 4), "min_ms": round(insitu_disp[1], 4), "max_ms": round(insitu_disp[2], 4), "launches": int(insitu_disp[3])},
                "kernel_span": None if not span_ms else {"avg_ms": round(insitu[0], 4), "min_ms": round(insitu[1], 4), "max_ms": round(insitu[2], 4), "launches": int(insitu[3]),
                                                         "frac": round(alg_bytes / (insitu[0] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                                         "what": "first workgroup start .. last workgroup end, stamped by the kernel itself on the 100 MHz constant clock: excludes the dispatch boundary a kernel trace (and a caller) pays"},
                "back_to_back": {"ms_per_launch": round(sweep_ms, 4), "achieved": round(achieved, 1), "frac": round(achieved / HBM_PEAK_GBS, 4)},
                "ms_with_cost_reduction": round(sweep_cost_ms, 4),
                "cost_sweep_ms": round(cost_ms, 4), "solve_ms": round(solve_ms, 4), "solve_stats": solve_stats}
    if traffic_note:
        roofline["traffic_note"] = traffic_note
    # what the accumulate figure IS: the LM loop's working set against the 256 MiB memory-side (Infinity) cache -- FETCH_SIZE / WRITE_SIZE count its
    # hits too (MI355X_MICROARCH.md) -- and the same launch COLD: 512 MiB of foreign data streamed through the memory side in front of every launch
    mem = ls.ctx.memory_info()
    roofline["working_set_bytes"] = mem["working_set_bytes"]
    roofline["infinity_cache_resident"] = bool(mem["working_set_bytes"] <= 256 * 2 ** 20)
    if world == 1 and info.is_sparse:
        flush_bytes = 512 * 2 ** 20
        ls.ctx.set_option(_capi.OPT_MATERIALIZE, 1)          # (the full accumulate launch: with the matrix-free trial enabled nlls_sweep_gradhess(ctx, NULL) enqueues nothing)
        ls.ctx.profile_sweep(True)
        for _ in range(12):
            ls.ctx.flush_cache(flush_bytes); ls.ctx.sweep_gradhess(want_cost=False)
        cold = ls.ctx.profile_sweep(False, read=True); cold_disp = ls.ctx.profile_sweep_dispatch()
        ls.ctx.set_option(_capi.OPT_MATERIALIZE, 1 if args.path == "materialise" else 0)
        cold_q = cold_disp if cold_disp and cold_disp[3] >= 3 and cold_disp[0] > 0 else cold
        if cold_q and cold_q[3] >= 3:
            ach_c = alg_bytes / (cold_q[0] * 1e-3) / 1e9
            roofline["cold"] = {"ms_per_launch": round(cold_q[0], 4), "min_ms": round(cold_q[1], 4), "max_ms": round(cold_q[2], 4), "launches": int(cold_q[3]),
                                "achieved": round(ach_c, 1), "frac": round(ach_c / HBM_PEAK_GBS, 4), "flushed_bytes_before_each_launch": flush_bytes,
                                "timing": ("dispatch timestamps" if cold_q is cold_disp else "execution span of the launch (kernel stamps)") + ", each launch behind a 512 MiB device-to-device copy of foreign data",
                                "kernel_span_ms": round(cold[0], 4) if cold and cold[3] >= 3 else None}
            roofline["frac_cold"] = roofline["cold"]["frac"]
            roofline["frac_hbm"] = roofline["cold"]["frac"]                                             # the launch with its data coming from HBM: the figure to hold against "of peak HBM"
            roofline["cold"]["frac_of_achievable"] = round(ach_c / HBM_ACHIEVABLE_GBS, 4)
    # which regime `frac` is in, in the numbers themselves: `frac` = the launch inside the LM loop (its working set may sit in the 256 MiB memory-side cache: infinity_cache_resident),
    # `frac_hbm` (= frac_cold) = the same launch behind 512 MiB of foreign traffic, and both against the copy rate the guide measured (6.29 TB/s) beside the 8 TB/s specification
    roofline["frac_of_achievable"] = round(achieved_q / HBM_ACHIEVABLE_GBS, 4)
    roofline["achievable_peak"] = HBM_ACHIEVABLE_GBS
    roofline["regime"] = ("in-loop figure: working set %.1f MiB %s the 256 MiB memory-side cache; frac_hbm is the HBM-resident figure" % (mem["working_set_bytes"] / 2 ** 20, "inside" if roofline["infinity_cache_resident"] else "beyond"))
    roofline["timed_loop_path"] = "matrix-free LM trial (no accumulate launch in the loop: `frac` is taken from a second, materialising loop of the same iterations)" if mf_loop else "materialising (the accumulate launch is the timed loop's own)"
    # HBM bytes of ONE LM iteration, both paths (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE over tools/lm_iters.py, all kernels of the loop: tools/pmc_iter.sh -> profiles/pmc_iter.json)
    hbm_iter = None
    try:
        rec = json.load(open(os.path.join(ROOT, "profiles", "pmc_iter.json"))).get(args.workload)
        if rec and rec.get("loop_code_hash") == loop_code_hash():
            hbm_iter = {k: rec[k] for k in ("matrix_free", "materialised", "note") if k in rec}
        elif rec:
            hbm_iter = {"refused": "profiles/pmc_iter.json was collected on a different build (re-run tools/pmc_iter.sh)"}
    except Exception:
        hbm_iter = None
    # ---- roofline of the reduced solve (north_star: MFMA utilisation on the reduced solve against chip peak)
    roofline_solve = None
    if world == 1 and info.nreduced_dof > 0 and reduced_ms > 0:
        n, bw, mode = int(info.nreduced_dof), int(info.bandwidth), int(info.solve_mode)
        useful = float(n) * bw * bw if mode == 2 else (2.0 * 128 ** 3 * solve_stats.get("tsp_products", 0) if mode == 3 else float(n) ** 3 / 3.0)
        issued = 2048.0 * solve_stats.get("bcr_mfma_issued", 0) if mode == 2 and solve_stats.get("bcr_mfma_issued", 0) else None
        # the launcher's count of issued MFMAs against the hardware counter (profiles/pmc_mfma.json, tools/pmc_mfma.sh): quoted only when the
        # counters were collected on THIS build of the solver and agree with it
        issued_check = None
        if issued:
            try:
                sys.path.insert(0, os.path.join(ROOT, "tools")); import pmc_mfma
                rec = json.load(open(os.path.join(ROOT, "profiles", "pmc_mfma.json")))
                if args.workload != "ba_1kx100k" or args.solver != "default":
                    issued_check = "unchecked: the counters of profiles/pmc_mfma.json were collected on ba_1kx100k with the default solver"
                elif rec.get("solver_code_hash") != pmc_mfma.solver_code_hash():
                    issued_check = "unchecked: profiles/pmc_mfma.json was collected on a different build of the solver sources (re-run tools/pmc_mfma.sh)"
                else:
                    hw = 2048.0 * rec["bcr_mfma_f64_instructions_per_solve"]
                    if abs(hw - issued) <= 0.02 * issued:
                        issued_check = f"confirmed by SQ_INSTS_VALU_MFMA_F64: {rec['bcr_mfma_f64_instructions_per_solve']:.0f} instructions per solve counted by the hardware"
                    else:
                        issued_check = f"REFUSED: the launcher counts {issued / 2048:.0f} MFMAs per solve, SQ_INSTS_VALU_MFMA_F64 {rec['bcr_mfma_f64_instructions_per_solve']:.0f}"; issued = None
            except Exception as e:
                issued_check = f"unchecked: {type(e).__name__}"
        roofline_solve = {"bound": "mfma", "unit": "TFLOP/s", "peak": FP64_MFMA_PEAK_TFLOPS,
                          "kernel": {2: "block cyclic reduction of the bordered band (bcr_panel / bcr_update / bcr_backward kernels)" if solve_stats.get("bcr_levels") else "twisted blocked band LDL' (chain kernels)",
                                     1: "dense blocked LDL' (MFMA trailing update)", 0: "one-wave dense solve",
                                     3: "tile-sparse LDL' in a nested-dissection order, level by level of the tile elimination tree (dense_panel<TSP> / tsp_trsm / tsp_update / tsp_backward kernels)"}[mode],
                          "reduced_dof": n, "bandwidth": bw, "us": round(1e3 * reduced_ms, 1),
                          "useful_flops": useful, "useful_flops_formula": "n * bw^2" if mode == 2 else ("2 * 128^3 * (tile products of the updates + panel products)" if mode == 3 else "n^3 / 3"),
                          # `achieved` / `frac` count USEFUL flops only; what the matrix cores are actually issued (tile padding, the diagonal
                          # block factored redundantly by every workgroup of a panel launch, inv(L)) is kept beside them as issued_*
                          "achieved": round(useful / (reduced_ms * 1e-3) / 1e12, 4),
                          "frac": round(useful / (reduced_ms * 1e-3) / 1e12 / FP64_MFMA_PEAK_TFLOPS, 5),
                          "issued_mfma_flops": issued, "issued_mfma_check": issued_check,
                          "issued_achieved": round(issued / (reduced_ms * 1e-3) / 1e12, 4) if issued else None,
                          "issued_frac": round(issued / (reduced_ms * 1e-3) / 1e12 / FP64_MFMA_PEAK_TFLOPS, 5) if issued else None,
                          "levels": solve_stats.get("tsp_levels") if mode == 3 else solve_stats.get("bcr_levels"), "launches": solve_stats.get("tsp_launches") if mode == 3 else solve_stats.get("bcr_launches"),
                          "tiles": solve_stats.get("tsp_tiles") if mode == 3 else None, "lower_tiles_stored": solve_stats.get("tsp_lower_tiles") if mode == 3 else None,
                          "note": "a banded LDL' is a chain of dependent pivots: latency-, not MFMA-bound -- the fraction says how far, not how well tuned"
                                  if mode == 2 else ("the depth of the tile elimination tree (levels), not the flops, is the dependent chain: one 128-pivot panel per level" if mode == 3 else None)}

    # ---- the DENSE reduced solve (NLLS_FLAG_NO_BAND) of the same reduced system: the only MFMA-bound kernel of the path, and the solver of every reduced
    # system that no ordering turns into a band.  One upload of its own, outside the timed region.
    roofline_solve_dense = None
    if world == 1 and info.is_sparse and info.has_schur and 512 <= info.nreduced_dof <= 20000 and args.solver == "default" and not os.environ.get("NLLS_BENCH_NO_DENSE"):
        dls = ShardedLS(problem, np.ones(problem.nvariables, bool), flags=_capi.FLAG_NO_BAND, device=local_rank, rank=rank, world=world, dist=dist, host_staged=host_staged)
        if dls.info.solve_mode == 1:
            dls.ctx.set_variables(start_vars, VARS_CURRENT); dls.ctx.sweep_gradhess(); dls.ctx.damp(1e-3 * dls.ctx.max_abs_diag())
            dls.ctx.time_solve(1)
            dms = dls.ctx.time_reduced_solve(5); nd = int(dls.info.nreduced_dof); uf = float(nd) ** 3 / 3.0
            roofline_solve_dense = {"bound": "mfma", "unit": "TFLOP/s", "peak": FP64_MFMA_PEAK_TFLOPS, "kernel": "dense blocked LDL' of the same reduced system (NLLS_FLAG_NO_BAND): dense_panel + syrk_update128 + fused backward",
                                    "reduced_dof": nd, "us": round(1e3 * dms, 1), "useful_flops": uf, "useful_flops_formula": "n^3 / 3",
                                    "achieved": round(uf / (dms * 1e-3) / 1e12, 3), "frac": round(uf / (dms * 1e-3) / 1e12 / FP64_MFMA_PEAK_TFLOPS, 4)}
        dls.close()

    # ---- where an iteration of the SHARDED loop goes, measured (N > 1, and one rank through the collective route): stream events at the phase boundaries of the collective
    # trial (NLLS_OPT_PHASE_EVENTS), one more loop of the same K iterations, outside the timed region -- beside `predicted`, so that the first multi-GPU run diagnoses itself
    phases = None
    if (world > 1 or force_dist) and info.is_sparse:
        ls.ctx.set_option(_capi.OPT_PHASE_EVENTS, 1)
        lp, pe = timed_loop(ls, problem, start_vars)
        ph = ls.ctx.phase_times(); ls.ctx.set_option(_capi.OPT_PHASE_EVENTS, 0)
        trials_per_it = lp.data.linearsolvers / max(1, args.steps)
        phases = {"per_trial_us": {k: round(v, 1) for k, v in ph.items() if k.endswith("_us") and k != "gradient_sweep_us"}, "gradient_sweep_us": round(ph["gradient_sweep_us"], 1),
                  "trials_counted": ph["trials"], "sweeps_counted": ph["sweeps"], "trials_per_iteration": round(trials_per_it, 3),
                  "sum_per_iteration_us": round(trials_per_it * sum(v for k, v in ph.items() if k.endswith("_us") and k != "gradient_sweep_us") + ph["gradient_sweep_us"], 1),
                  "ms_per_step_of_this_loop": round(1e3 * pe / args.steps, 4),
                  "how": "hipEventRecord on the library's stream at the phase boundaries of nlls_lm_trial (collective route) and around the gradient sweep; rank 0's figures; the all-reduce phase "
                         "contains the wait for the slowest rank; the events themselves cost a marker packet each (this loop is not the timed one)"}

    # ---- strong-scaling leg (N > 1 only): the TEN-times workload (10k cameras x 1M points x 10M residual blocks) split over the ranks -- there about 70 % of an iteration divides by N
    # (sweep, elimination, back-substitution of 10M blocks against a reduced solve that grows only with the cameras); every rank generates and uploads only its share (NLLS_FLAG_PRESHARDED)
    strong = None
    if world > 1 and args.workload == "ba_1kx100k" and not os.environ.get("NLLS_BENCH_NO_STRONG"):
        sc, sp, spr = CONFIGS["ba_10kx1M"]
        sproblem = synthetic.create_ba_problem_shard(sc, sp, spr, rank, world, seed=1, robust=N.HuberKernel(0.01), outlier_frac=0.05, outlier_sigma=0.05, pointnoise=1e-3, posenoise=1e-3)
        sls = ShardedLS(sproblem, np.ones(sproblem.nvariables, bool), flags=flags, device=local_rank, rank=rank, world=world, dist=dist, host_staged=host_staged, presharded=True)
        sloop, selapsed = timed_loop(sls, sproblem, sproblem.variables.copy())
        scounts = [None] * world; dist.all_gather_object(scounts, int(sproblem.ncosts())); stotal = int(sum(scounts))
        strong = {"scaling": "strong", "workload": f"ba_10kx1M: {sc} cameras x {sp} points ({stotal} residual blocks) split by point over {world} ranks",
                  "value": round(args.steps / selapsed, 3), "unit": "LM iters/s", "ms_per_step": round(1e3 * selapsed / args.steps, 4), "residual_blocks_per_s": round(stotal * args.steps / selapsed, 1),
                  "lm_trials_per_s": round(sloop.data.linearsolvers / selapsed, 1), "local_residual_blocks": int(sls.local_nobs), "start_cost": sloop.data.startcost, "final_cost": sloop.data.bestcost,
                  "one_gpu_reference": "profiles/r06e_bench_ba_10kx1M.json (python bench.py --workload ba_10kx1M on one MI355X: 695 LM iterations/s, matrix-free)"}
        sls.close()

    # ---- weak-scaling leg (N > 1 only): N x 100k points against the SAME cameras, sharded by point -- per-rank sweeps, elimination and
    # back-substitution stay those of the one-GPU problem, the replicated reduced system keeps its size; what grows is the data volume
    weak = None
    if world > 1 and args.workload == "ba_1kx100k":
        # every rank generates and uploads ONLY its own 100k points (NLLS_FLAG_PRESHARDED): the N-times larger problem is never built anywhere
        wproblem = synthetic.create_ba_problem_shard(ncam, world * npts, prop, rank, world, seed=1, robust=N.HuberKernel(0.01), outlier_frac=0.05, outlier_sigma=0.05,
                                                     pointnoise=1e-3, posenoise=1e-3)
        wls = ShardedLS(wproblem, np.ones(wproblem.nvariables, bool), flags=flags, device=local_rank, rank=rank, world=world, dist=dist, host_staged=host_staged, presharded=True)
        wloop, welapsed = timed_loop(wls, wproblem, wproblem.variables.copy())
        wcounts = [None] * world; dist.all_gather_object(wcounts, int(wproblem.ncosts())); wtotal = int(sum(wcounts))
        weak = {"scaling": "weak", "workload": f"{ncam} cameras x {world * npts} points ({wtotal} residual blocks; {npts} points generated and uploaded per rank)",
                "value": round(args.steps / welapsed, 3), "unit": "LM iters/s", "ms_per_step": round(1e3 * welapsed / args.steps, 4),
                "residual_blocks_per_s": round(wtotal * args.steps / welapsed, 1),
                "lm_trials_per_s": round(wloop.data.linearsolvers / welapsed, 1), "local_residual_blocks": int(wls.local_nobs),
                "start_cost": wloop.data.startcost, "final_cost": wloop.data.bestcost}
        wls.close()

    # ---- CPU baseline: the oracle's own optimize! loop on a bounded sample of the same workload (rank 0, N = 1)
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import oracle as O
        problem.variables[:] = start_vars
        op = O.OracleProblem(problem.var_kind, problem.var_dim, problem.groups()); op.set_variables(problem.variables)
        r = op.optimize(maxiters=args.cpu_iters, reldcost=-1e300, absdcost=-1e300, dstep=-1.0, maxfails=10 ** 9, maxtime=1e6)
        t_iter = r.timecost + r.timegradient + r.timesolver        # excludes the one-off symbolic analysis
        cpu = {"value": round(r.niterations / t_iter, 4), "unit": "LM iters/s", "cores": 1, "kind": "port",
               "sample": f"{r.niterations} LM iterations of the same {args.workload} problem ({nobs} residual blocks); "
                         f"gradient {r.timegradient:.2f}s, cost {r.timecost:.2f}s, solver {r.timesolver:.2f}s, "
                         f"one-off setup {r.timetotal - t_iter:.2f}s excluded",
               "residual_blocks_per_s": round(nobs * r.gradientcomputations / max(r.timegradient, 1e-9), 1),
               "host_cores_available": os.cpu_count()}

    if rank == 0:
        out = {
            "metric": "LM iterations/s on synthetic BA (1k cams x 100k pts x ~1M obs)" if args.workload == "ba_1kx100k"
                      else f"LM iterations/s on {args.workload}",
            "value": round(args.steps / elapsed, 3), "unit": "LM iters/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 4),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": args.workload, "ncameras": ncam, "npoints": npts, "nobs": nobs, **workload_desc, "solver": args.solver, "ndof": int(info.ndof),
                       "reduced_dof": int(info.nreduced_dof), "sharding": ("none" if not force_dist else "none (one rank through the sharded route: NLLS_BENCH_FORCE_DIST)") if world == 1 else f"by point over {world} ranks"},
            "residual_blocks_per_s": round(nobs * args.steps / elapsed, 1),
            "sweep_residual_blocks_per_s": round(ls.local_nobs * world / (sweep_ms * 1e-3), 1),
            "lm": {"start_cost": start_cost, "final_cost": final_cost, "linear_solves": data.linearsolvers,
                   "cost_sweeps": data.costcomputations, "gradient_sweeps": data.gradientcomputations,
                   # an outer iteration that rejects a step solves again with more damping (src/iterators.jl:149-172): at the
                   # optimiser's noise floor (from about the 12th iteration of this problem) that happens often, so the rate
                   # of LM trials (damped solve + retraction + cost sweep) is the figure that does not depend on --steps
                   "lm_trials_per_s": round(data.linearsolvers / elapsed, 1), "lm_trials_per_s_median": round(trials_median, 1),
                   "noise_floor_iterations": noise_floor_iterations, "noise_floor_from_iteration": floor_at,
                   "first_iteration_with_a_rejected_trial": first_reject, "iterations_with_rejected_trials": n_reject_iters, "trials_per_iteration": trials_trace,
                   # the CPU oracle's own loop (the reference's full sparse LDL') over the same K iterations from the same start: a committed fixture
                   # (tests/golden/oracle_trials.json, tools/oracle_trials.py), not a run on this box
                   **oracle_loop_fixture(args.workload, args.steps)},
            "spread": spread,
            "lm_path": ("matrix-free LM trial: the cost blocks are evaluated inside the Schur elimination and the back-substitution, the eliminated rows of A.data are never formed (nlls_mf.hip)" if mf_loop
                        else "materialised: accumulate sweep -> A.data -> elimination"),
            "materialised": materialised, "hbm_bytes_per_lm_iteration": hbm_iter,
            "roofline": roofline, "roofline_solve": roofline_solve, "roofline_solve_dense": roofline_solve_dense, "cpu_baseline": cpu,
        }
        if world > 1 or force_dist:
            out["rccl"] = ls.ctx.comm_info()        # what the library's communicator reports (ncclCommCount / ncclCommUserRank), not the environment
        if weak:
            out["weak_scaling"] = weak
        if strong:
            out["strong_scaling_10x"] = strong
        if phases:
            out["phases_measured"] = phases
        if world > 1 and args.workload == "ba_1kx100k":
            out["predicted"] = predicted_scaling(world)
        sys.stdout.flush(); os.dup2(saved_stdout, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)
    ls.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
