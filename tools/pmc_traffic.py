#!/usr/bin/env python3
"""Turns the two rocprofv3 --pmc passes of tools/measure_round.sh (FETCH_SIZE, WRITE_SIZE over tools/sweep_only.py) into
profiles/pmc_traffic.json: HBM bytes per accumulate launch, corrected as MI355X_MICROARCH.md's HBM section prescribes (KiB units;
FETCH_SIZE doubled on gfx950), together with the hash of the sweep sources the counters were collected on -- bench.py refuses
the file when the hash does not match the build it runs."""
import csv, glob, json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import sweep_code_hash

PRIMARY = ("gh_fused_kernel", "gh_fold_kernel")                 # one of these per accumulate sweep; gh_fold_gather_kernel rides with gh_fold_kernel (three-slot kinds: config 5)
ALL = PRIMARY + ("gh_fold_gather_kernel",)

def per_launch(pattern, counter):
    f = sorted(glob.glob(pattern))[-1]
    rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == counter and any(k in r["Kernel_Name"] for k in ALL)]
    n = sum(1 for r in rows if any(k in r["Kernel_Name"] for k in PRIMARY))
    return sum(float(r["Counter_Value"]) for r in rows) / n * 1024.0, n, f

def trace_avg_ns(pattern):
    """rocprofv3 --kernel-trace --stats of the bench run of the same build: average dispatch duration of the accumulate launch(es), summed over its kernels"""
    fs = sorted(glob.glob(pattern))
    if not fs:
        return None
    tot = 0.0
    for r in csv.DictReader(open(fs[-1])):
        if any(k in r["Name"] for k in ALL):
            tot += float(r["AverageNs"])
    return tot or None

tag, workload = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "ba_1kx100k")
sfx = "" if workload == "ba_1kx100k" else "_" + workload
fetch, nf, ff = per_launch(f"gpurun_out/{tag}_fetch{sfx}/*/*counter_collection.csv", "FETCH_SIZE")
write, nw, fw = per_launch(f"gpurun_out/{tag}_write{sfx}/*/*counter_collection.csv", "WRITE_SIZE")
path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
rec = json.load(open(path)) if os.path.exists(path) else {}
rec[workload] = {"hbm_bytes_per_sweep": int(2 * fetch + write), "fetch_bytes_corrected": int(2 * fetch), "write_bytes": int(write), "launches_averaged": [nf, nw],
                 "sweep_code_hash": sweep_code_hash(),
                 "rocprof_avg_dispatch_ns": trace_avg_ns(f"gpurun_out/{tag}_stats{sfx}/*/*kernel_stats.csv"),
                 "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over tools/sweep_only.py (tools/measure_round.sh); KiB units; FETCH_SIZE doubled "
                         "(gfx950 reports half of wide coalesced reads, MI355X_MICROARCH.md HBM section); the accumulate launch(es) of one sweep (gh_fused_kernel, or gh_fold_kernel + gh_fold_gather_kernel), average per sweep; rocprof_avg_dispatch_ns: the same launches' average dispatch duration in the kernel trace of the bench run (loop and back-to-back launches together)"}
json.dump(rec, open(path, "w"), indent=1)
print(json.dumps(rec[workload]))
