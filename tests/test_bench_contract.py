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


def test_line_names_its_own_config():
    assert bench.baseline_config("usv_model_pf_ca", 65536, 1, 40, 10, False) == "BASELINE.json configs[2]"
    assert bench.baseline_config("usv_model_pf_ca", 1024, 1, 20, 3, False) == "BASELINE.json configs[1]"
    assert bench.baseline_config("usv_model_pf_ca", 32768, 8, 40, 10, False) == "BASELINE.json configs[3]"
    assert bench.baseline_config("usv_model_pf_ca", 8192, 8, 80, 20, True) == "BASELINE.json configs[4]"
    assert "configs[2] per GPU x8" in bench.baseline_config("usv_model_pf_ca", 65536, 8, 40, 10, False)
    assert bench.baseline_config("usv_model_pf_ca", 4096, 1, 40, 10, False).startswith("custom")


def test_gpus_gt_visible_devices_is_refused():
    """`python bench.py --gpus 2` without a launcher on a box with fewer than 2 GPUs must fail loudly instead of printing
    a 1-GPU number labelled n_gpus = 2; and a launcher whose WORLD_SIZE disagrees with --gpus is refused too."""
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--batch", "8", "--cpu-sample", "0"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "refusing" in r.stderr and "{" not in r.stdout
    env["WORLD_SIZE"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "1", "--warmup", "0",
                        "--batch", "8", "--cpu-sample", "0"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "refusing" in (r.stderr + r.stdout) and '"metric"' not in r.stdout
