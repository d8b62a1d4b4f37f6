#!/usr/bin/env python3
"""rocprofv3 --pmc CSVs of tools/pmc_mfma.sh -> profiles/pmc_mfma.json: per kernel of the reduced solve and PER SOLVE, the f64 MFMA instructions the
hardware counted (SQ_INSTS_VALU_MFMA_F64), the matrix pipe's busy cycles and the shader's busy cycles, with the hash of the solver sources they were
collected on.  bench.py checks its launcher-side count of issued MFMAs (BcrSolver::mfma_issued) against the counter and refuses to quote an
issued-flops figure that the hardware does not confirm."""
import csv, glob, hashlib, json, os, sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def solver_code_hash():
    h = hashlib.sha256()
    for f in ("nlls_bcr.hip", "nlls_bcr.hpp", "nlls_solve.hip"):
        h.update(open(os.path.join(ROOT, "nllssolver.jl_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def collect(pattern, solve_marker):
    f = sorted(glob.glob(pattern))[-1]
    per = defaultdict(lambda: defaultdict(float)); calls = defaultdict(set)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("nlls::", "")
        per[k][r["Counter_Name"]] += float(r["Counter_Value"]); calls[k].add(r["Dispatch_Id"])
    nsolve = max([len(calls.get(m, ())) for m in solve_marker] + [1])       # a kernel launched exactly once per solve
    out = {}
    for k, c in per.items():
        if c.get("SQ_INSTS_VALU_MFMA_F64", 0) <= 0:
            continue
        out[k] = {"launches_per_solve": round(len(calls[k]) / nsolve, 2), "mfma_f64_instructions_per_solve": c["SQ_INSTS_VALU_MFMA_F64"] / nsolve,
                  "mfma_busy_cycles_per_solve": c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / nsolve, "sq_busy_cycles_per_solve": c.get("SQ_BUSY_CYCLES", 0) / nsolve,
                  "valu_instructions_per_solve": c.get("SQ_INSTS_VALU", 0) / nsolve}
    return out, nsolve, f


if __name__ == "__main__":
    tag = sys.argv[1]
    once = ("schur_elim_all_kernel<3>", "schur_backsub_fast_kernel<3>")
    band, nb, fb = collect(f"gpurun_out/{tag}_mfma_band/*/*counter_collection.csv", once)
    dense, nd_, fd = collect(f"gpurun_out/{tag}_mfma_dense/*/*counter_collection.csv", once)
    bcr = sum(v["mfma_f64_instructions_per_solve"] for k, v in band.items() if k.startswith("bcr_"))
    rec = {"solver_code_hash": solver_code_hash(), "workload": "ba_1kx100k (tools/solve_only.py)", "solves_counted": {"band": nb, "dense": nd_},
           "band": band, "dense": dense, "bcr_mfma_f64_instructions_per_solve": bcr,
           "note": "rocprofv3 --pmc SQ_INSTS_VALU_MFMA_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU (tools/pmc_mfma.sh); counter values summed over the dispatches of a "
                   "kernel and divided by the number of solves; one v_mfma_f64_16x16x4_f64 = 2048 flop"}
    json.dump(rec, open(os.path.join(ROOT, "profiles", "pmc_mfma.json"), "w"), indent=1)
    print(json.dumps({k: rec[k] for k in ("bcr_mfma_f64_instructions_per_solve", "solves_counted")}))
    for name, d in (("band", band), ("dense", dense)):
        for k, v in sorted(d.items(), key=lambda kv: -kv[1]["mfma_f64_instructions_per_solve"]):
            print(f"{name:5s} {k[:44]:44s} launches/solve {v['launches_per_solve']:6.1f}  mfma {v['mfma_f64_instructions_per_solve']:12.0f}  mfma busy cyc {v['mfma_busy_cycles_per_solve']:14.0f}  sq busy cyc {v['sq_busy_cycles_per_solve']:14.0f}")
