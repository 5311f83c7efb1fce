"""bench.py's contract, checked without a GPU: the algorithmic byte counts of SURVEY 8(d), the CPU-core detection,
and the shape of the JSON line (on the line committed next to the profiles of the same build)."""
import json
import os

import bench

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_algorithmic_bytes_match_the_survey_table():
    assert bench.algorithmic_bytes(14, 2, 40, 10) == 11060      # M2, N=40, K=10
    assert bench.algorithmic_bytes(8, 1, 40, 10) == 6332        # M1, N=40, K=10
    assert bench.algorithmic_bytes(5, 2, 20, 0) == 2460         # M0, N=20
    assert bench.algorithmic_bytes(14, 2, 20, 3) == 5772        # M2, N=20, K=3


def test_usable_cores_is_sane():
    n = bench.usable_cores()
    assert 1 <= n <= (os.cpu_count() or 1)


def test_committed_bench_line_has_the_contract_fields():
    d = json.load(open(os.path.join(ROOT, "profiles", "r01_g_bench.json")))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["unit"] == "solves/s" and d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "f64" and d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] == "port" and c["cores"] >= 1
    # value = units all ranks processed / time
    assert abs(d["value"] - d["config"]["instances_total"] * d["steps"] / (d["ms_per_step"] * 1e-3 * d["steps"])) < 1e-6 * d["value"]
