"""The bench line's contract on ONE GPU, run the way the driver runs it (a child process, one JSON line on stdout) on BASELINE config 3 so that it takes seconds."""
import json, os, subprocess, sys
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.gpu
def test_bench_line_carries_the_contract_and_the_round_6_fields():
    r = subprocess.run([sys.executable, os.path.join(os.path.dirname(HERE), "bench.py"), "--workload", "ba_100x10k", "--steps", "6", "--warmup", "2", "--repeats", "3"],
                       capture_output=True, text=True, timeout=600, env=dict(os.environ, NLLS_BENCH_NO_DENSE="1"))
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 6 and d["warmup"] == 2 and d["higher_is_better"] is True and d["dtype"] == "f64" and d["data"] == "synthetic"
    assert d["config"]["workload"] == "ba_100x10k" and d["vs_baseline"] is None and d["value"] > 0
    assert abs(d["value"] * d["ms_per_step"] * 1e-3 - 1.0) < 0.02                      # value = steps per second of the timed loop
    # which trial the timed loop ran, and the same iterations through the other one
    assert d["lm_path"].startswith("matrix-free") and d["materialised"]["value"] > 0
    assert abs(d["materialised"]["final_cost"] - d["lm"]["final_cost"]) <= 1e-9 * abs(d["lm"]["final_cost"])
    assert abs(d["lm"]["final_cost"] - d["lm"]["oracle_final_cost"]) <= 1e-9 * abs(d["lm"]["oracle_final_cost"]) if d["lm"].get("oracle_final_cost") and d["steps"] == 20 else True
    rf = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "frac_cold", "frac_hbm", "frac_of_achievable", "achievable_peak", "regime", "timed_loop_path", "algorithmic_bytes_per_launch"):
        assert k in rf, k
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0 and rf["achievable_peak"] == 6290.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3 and rf["frac_hbm"] == rf["frac_cold"] and 0 < rf["frac"] < 1
    assert "matrix-free" in rf["timed_loop_path"]
    # HBM bytes of one LM iteration: the PMC record of THIS build, or an explicit refusal -- never a stale number
    h = d["hbm_bytes_per_lm_iteration"]
    assert h is None or "refused" in h or (h["matrix_free"]["hbm_bytes_per_lm_iteration"] < h["materialised"]["hbm_bytes_per_lm_iteration"])
    cb = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"): assert k in cb, k
    assert cb["kind"] == "port" and cb["cores"] == 1 and cb["value"] > 0 and d["value"] > 100 * cb["value"]
