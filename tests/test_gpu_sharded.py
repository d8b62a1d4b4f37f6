"""N > 1 path on real kernels: two ranks share the single GPU of the test box (gloo + host-staged reductions;
RCCL itself refuses two ranks on one device) and must reproduce the unsharded result."""
import os
import subprocess
import sys
import time

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


# (a string of seeds = several scenarios in ONE process group: most of a scenario's two seconds is the start of its ranks)
@pytest.mark.parametrize("world,backend,seed", [(1, "nccl", 21), (2, "gloo", "21,401,403,5001,6001,7001,5201,8001,9001,10001"), (3, "gloo", "21,402,5002,5102,7001,5301,9001,10002"),
                                                (2, "gloo", -1), (3, "gloo", -1), (2, "gloo", 3000), (3, "gloo", 3000), (2, "gloo", 4000), (3, "gloo", 4000)])
def test_sharded_matches_unsharded(world, backend, seed):
    """(1, "nccl"): the local / reduce / finish route over RCCL with device buffers, one rank -- the plumbing bench.py --gpus N uses.
    seed 3000: BASELINE config 3 at full size (100 cameras x 10k points, ~100k residual blocks) against the CPU oracle: cost, gradient,
    damped step and four Levenberg-Marquardt iterations.
    seed 4000: BASELINE config 4 at full size (1k cameras x 100k points x 1M residual blocks -- the configuration north_star shards over 1 / 2 / 4 / 8 GPUs) against the CPU oracle: cost 1e-11,
    gradient 1e-10, damped step 1e-7, four Levenberg-Marquardt iterations 1e-8 (host loop and the library's own), as seed 3000 does for config 3.
    seed -1: a zero pivot only one rank sees must be raised on every rank (no rank left behind in a collective).
    seed >= 6000: the cameras' labels permuted -- every rank re-orders the reduced system at upload (reverse Cuthill-McKee) and must arrive at the same order;
    seed 51xx: the same under NLLS_FLAG_PRESHARDED, where each rank sees only its own part of the camera graph (the union is taken collectively).
    seed 52xx: a pre-sharded upload whose reduced rows differ in layout between ranks (points listed before the cameras) is refused on every rank.
    seed 7001: a deadline (maxtime) only rank 0 crosses -- all ranks leave the loop in the same iteration (no rank left in a collective).
    seed 100xx: problems that do not shard (dense systems) run as REPLICAS -- whole on every rank, no collective, the unsharded result (round 5; refused before).
    seed 9001: a bundle adjustment with three dynamic-size variables beside it (round 4: dynamic-size blocks in a block-sparse system, also under sharding).
    seed 8001 / 5301: a 24 x 24 camera grid with permuted labels -- the reduced system goes to the tile-sparse solver (solve_mode 3); the nested dissection at upload
    must give every rank the same tiles (5301: pre-sharded, from the union of the ranks' camera graphs).
    5000 <= seed < 6000: NLLS_FLAG_PRESHARDED -- every rank uploads only its own share (all cameras + its points), the reduced system's layout is agreed on
    collectively, and the library's outer loop must reach the unsharded result."""
    seeds = [int(v) for v in str(seed).split(",")]
    port = str(29500 + world + seeds[0] % 50 + 7 * (len(seeds) > 1))
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "sharded_worker.py"), str(r), str(world), port, backend, str(seed)],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = []
    deadline = time.monotonic() + 150 + 20 * len(seeds) + (400 if 4000 in seeds else 0)          # all ranks share one deadline: a rank that died leaves the others in a collective
    for p in procs:
        try:
            out, _ = p.communicate(timeout=max(1.0, deadline - time.monotonic()))
        except subprocess.TimeoutExpired:
            for q in procs:
                if q.poll() is None:
                    q.kill()
            out, _ = p.communicate()
        outs.append(out)
    for r, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {r} failed:\n{out[-3000:]}"
        if seeds[0] < 0: assert "singular block raised on every rank" in out
        else: assert out.count("sharded == unsharded") == len(seeds), out[-3000:]


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` WITHOUT a launcher (the way the driver starts N = 1): bench.py starts one child process per rank before it
    touches the GPU, relays rank 0's JSON line and fails if a rank fails.  Rehearsed on the one GPU with gloo + host-staged reductions
    (RCCL refuses two ranks on one device); the sharded trials run in the library's own loop (collectives behind the C ABI)."""
    import json
    env = dict(os.environ, NLLS_BENCH_BACKEND="gloo")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(os.path.dirname(HERE), "bench.py"), "--gpus", "2", "--workload", "ba_100x10k", "--steps", "4", "--warmup", "1",
                        "--repeats", "3", "--no-cpu-baseline"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=240)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines                      # stdout carries exactly one line
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 4 and out["config"]["sharding"] == "by point over 2 ranks"
    assert out["lm"]["final_cost"] < out["lm"]["start_cost"] and out["value"] > 0
    # the N > 1 line diagnoses itself: measured microseconds per phase of the collective trial beside the prediction, and the strong-scaling leg on the ten-times workload
    ph = out["phases_measured"]
    assert ph["trials_counted"] >= 4 and all(ph["per_trial_us"][k] > 0 for k in ("elimination_us", "allreduce_S_us", "reduced_solve_us", "backsub_retraction_us", "trial_tail_us")), ph
    assert ph["gradient_sweep_us"] > 0 and 0.3 * 1e3 * ph["ms_per_step_of_this_loop"] < ph["sum_per_iteration_us"] < 1.5 * 1e3 * ph["ms_per_step_of_this_loop"], ph
    assert out["rccl"]["nranks"] == 2 and out["rccl"]["rank"] == 0 and out["spread"]["runs"] >= 3 and len(out["spread"]["trials_per_run"]) == out["spread"]["runs"]


def test_bench_fails_fast_when_a_rank_dies():
    """A rank that dies in the middle of the timed loop leaves its peers inside a collective.  bench.py's own launcher watches ALL its children: it must
    terminate the survivors and exit non-zero within seconds, not sit until the driver's limit (rank 1 exits with code 17 half way through the runs)."""
    env = dict(os.environ, NLLS_BENCH_BACKEND="gloo")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    t0 = time.monotonic()
    r = subprocess.run([sys.executable, os.path.join(os.path.dirname(HERE), "bench.py"), "--gpus", "2", "--workload", "ba_100x10k", "--steps", "4", "--warmup", "1",
                        "--no-cpu-baseline", "--die-rank", "1"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=200)
    took = time.monotonic() - t0
    assert r.returncode != 0 and "ranks failed (rank, exit code): [(1, 17)]" in r.stderr, (r.returncode, r.stderr[-2000:])
    assert not r.stdout.strip()                        # no result line from a failed run
    assert took < 90, took                             # (start-up of two torch processes included; the wait for the dead rank's peers is what must be short)
