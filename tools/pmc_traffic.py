#!/usr/bin/env python3
"""Turns the two rocprofv3 --pmc passes of tools/measure_round.sh (FETCH_SIZE, WRITE_SIZE over tools/sweep_only.py) into
profiles/pmc_traffic.json: HBM bytes per accumulate launch, corrected as MI355X_MICROARCH.md's HBM section prescribes (KiB units;
FETCH_SIZE doubled on gfx950), together with the hash of the sweep sources the counters were collected on -- bench.py refuses
the file when the hash does not match the build it runs."""
import csv, glob, json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import sweep_code_hash

def per_launch(pattern, counter, kernel_substr):
    f = sorted(glob.glob(pattern))[-1]
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if r["Counter_Name"] == counter and kernel_substr in r["Kernel_Name"]]
    return sum(vals) / len(vals) * 1024.0, len(vals), f

tag, workload = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "ba_1kx100k")
kern = "gh_fused"            # gh_fused_kernel (two-slot kinds) / gh_fused3_kernel (three-slot kinds: config 5)
sfx = "" if workload == "ba_1kx100k" else "_" + workload
fetch, nf, ff = per_launch(f"gpurun_out/{tag}_fetch{sfx}/*/*counter_collection.csv", "FETCH_SIZE", kern)
write, nw, fw = per_launch(f"gpurun_out/{tag}_write{sfx}/*/*counter_collection.csv", "WRITE_SIZE", kern)
path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
rec = json.load(open(path)) if os.path.exists(path) else {}
rec[workload] = {"hbm_bytes_per_sweep": int(2 * fetch + write), "fetch_bytes_corrected": int(2 * fetch), "write_bytes": int(write), "launches_averaged": [nf, nw],
                 "sweep_code_hash": sweep_code_hash(),
                 "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over tools/sweep_only.py (tools/measure_round.sh); KiB units; FETCH_SIZE doubled "
                         "(gfx950 reports half of wide coalesced reads, MI355X_MICROARCH.md HBM section); the fused accumulate launch (light + heavy tiles in one launch), average per launch"}
json.dump(rec, open(path, "w"), indent=1)
print(json.dumps(rec[workload]))
