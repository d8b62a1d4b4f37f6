#!/usr/bin/env python3
"""The two rocprofv3 --pmc passes per path of tools/pmc_iter.sh -> profiles/pmc_iter.json: HBM bytes per LM iteration (FETCH_SIZE doubled on gfx950, KiB units: MI355X_MICROARCH.md's
HBM section), summed over EVERY kernel of the loop and divided by the iterations, per kernel beside the total, with the hash of the sources the counters were collected on."""
import csv, glob, json, os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import loop_code_hash

tag, w = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "ba_1kx100k")
def total(pattern, counter):
    """bytes per kernel name over the dispatches of the LOOP: everything in front of the library's first own kernel -- the runtime's copy / fill kernels of the upload (the hot
    arena's compaction moves ~250 MB once) -- is not an LM iteration's traffic and is left out"""
    f = sorted(glob.glob(pattern))[-1]; per = collections.Counter()
    rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == counter]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    first = next((i for i, r in enumerate(rows) if "nlls::" in r["Kernel_Name"]), 0)
    for r in rows[first:]:
        per[r["Kernel_Name"].split("(")[0].replace("void nlls::", "").replace("nlls::", "")[:48]] += float(r["Counter_Value"]) * 1024.0
    return per
out = {}
for path, key in (("mf", "matrix_free"), ("mat", "materialised")):
    run = json.load(open(f"gpurun_out/{tag}_iter_{path}_{w}.json")); it = run["iterations"]
    fe = total(f"gpurun_out/{tag}_iter_fetch_{path}_{w}/*/*counter_collection.csv", "FETCH_SIZE"); wr = total(f"gpurun_out/{tag}_iter_write_{path}_{w}/*/*counter_collection.csv", "WRITE_SIZE")
    kern = {k: int((2 * fe.get(k, 0) + wr.get(k, 0)) / it) for k in sorted(set(fe) | set(wr), key=lambda k: -(2 * fe.get(k, 0) + wr.get(k, 0)))}
    out[key] = {"hbm_bytes_per_lm_iteration": int((2 * sum(fe.values()) + sum(wr.values())) / it), "fetch_bytes_corrected": int(2 * sum(fe.values()) / it), "write_bytes": int(sum(wr.values()) / it),
                "iterations": it, "linear_solves": run["linear_solves"], "per_kernel_bytes_per_iteration": {k: v for k, v in list(kern.items())[:14]}}
path = os.path.join(ROOT, "profiles", "pmc_iter.json")
rec = json.load(open(path)) if os.path.exists(path) else {}
rec[w] = dict(out, loop_code_hash=loop_code_hash(),
              note="rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes over tools/lm_iters.py (tools/pmc_iter.sh): every kernel from the library's first launch on -- 20 LM iterations, the loop's first full sweep included; the upload's copy kernels excluded --, "
                   "KiB units, FETCH_SIZE doubled (gfx950: MI355X_MICROARCH.md HBM section); the counters include hits of the memory-side cache")
json.dump(rec, open(path, "w"), indent=1)
json.dump(rec, open(os.path.join(ROOT, "gpurun_out", f"{tag}_pmc_iter.json"), "w"), indent=1)      # (gpurun merges gpurun_out/ back; copy to profiles/pmc_iter.json)
print(json.dumps({k: rec[w][k]["hbm_bytes_per_lm_iteration"] for k in ("matrix_free", "materialised")}))
