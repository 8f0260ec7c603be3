"""CPU test: the committed bench line (profiles/r03c_final_bench.json, produced by `python bench.py` on an MI355X) carries every field the
driver's contract names, and its derived numbers are consistent with each other."""
import glob
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def latest_line():
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_final_bench.json")))
    assert files, "no committed bench line under profiles/"
    with open(files[-1]) as fh:
        return json.loads(fh.read().strip().splitlines()[-1]), files[-1]


def test_committed_bench_line_has_the_contract_fields():
    d, path = latest_line()
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline"):
        assert k in d, (path, k)
    assert d["n_gpus"] == 1 and d["higher_is_better"] is True and d["vs_baseline"] is None and d["dtype"] == "f64"
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    # achieved = algorithmic bytes / measured kernel time; traffic (PMC) must not be below the bytes that physically move
    assert abs(r["achieved"] - r["algorithmic_bytes"] / (r["kernel_ms"] * 1e-3) / 1e9) / r["achieved"] < 0.01
    assert r["traffic"] is None or r["traffic"] >= r["physical_bytes"]
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("port", "reference") and c["cores"] == 1
    # the step cannot be shorter than its dominant kernel, and value = algorithmic bytes per step
    assert d["ms_per_step"] >= r["kernel_ms"] * 0.98
    assert abs(d["value"] - r["algorithmic_bytes"] / (d["ms_per_step"] * 1e-3) / 1e9) / d["value"] < 0.01


def test_committed_pmc_summary_matches_the_kernel_sources():
    import sys
    sys.path.insert(0, ROOT)
    import bench
    import pytest
    path, pm = bench.committed_pmc()
    if path is None:      # kernel sources edited since the last profile round: bench.py then reports traffic = null (never a stale number)
        pytest.skip("no profiles/*_pmc_summary.json was made from the current spmv.hip + rebalance.hip + dsa_dev.h: re-run tools/scripts/profile_round.sh")
    assert pm["kernel_source_sha"] == bench.kernel_source_sha()


def test_committed_kernel_trace_agrees_with_the_bench_line():
    """The rocprofv3 --kernel-trace --stats summary of the same command (profiles/<tag>_spmv_only_kernel_stats.csv) must give the
    duration bench.py measured with HIP events for the dominant kernel: within 4 % of each other."""
    import csv
    d, path = latest_line()
    stats = path.replace("_bench.json", "_spmv_only_kernel_stats.csv")
    assert os.path.exists(stats), stats
    rows = [r for r in csv.DictReader(open(stats)) if "k_spmv_gather" in r["Name"]]
    assert rows, "no k_spmv_gather row in " + stats
    top = max(rows, key=lambda r: int(r["Calls"]))
    trace_ms = float(top["AverageNs"]) / 1e6
    bench_ms = d["roofline"]["kernel_ms"]
    assert abs(trace_ms - bench_ms) / bench_ms < 0.04, (trace_ms, bench_ms)
