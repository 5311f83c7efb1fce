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


def test_global_batch_mode_shards_one_seed_1234_batch():
    """SURVEY.md 8(d) configs[3] / [4]: ONE batch from seed 1234, instance b on rank floor(b * world / B): the ranks' workloads are the
    contiguous slices of the single-rank workload - also for a batch that does not divide evenly."""
    import numpy as np
    for G, world in ((64, 8), (50, 4)):
        whole, B1, dt, steps, sigma, mask = bench.make_workload("usv_model_pf_ca", 10, 4, 9999, G, "survey", False, 0, 1)
        assert B1 == G and steps == 5 and dt == 0.05 and sigma == 1e-3
        parts = [bench.make_workload("usv_model_pf_ca", 10, 4, 9999, G, "survey", False, r, world) for r in range(world)]
        assert sum(p[1] for p in parts) == G
        for k in ("x0", "x_init", "u_init", "yref", "yref_e", "p", "lh"):
            assert np.array_equal(np.concatenate([p[0][k] for p in parts], axis=0), whole[k]), k
        # instance b sits on rank floor(b * world / G)
        off = 0
        for r, p in enumerate(parts):
            for b in range(off, off + p[1]):
                assert b * world // G == r, (b, r)
            off += p[1]
    # without --global-batch every rank draws its own batch (weak scaling)
    a = bench.make_workload("usv_model_pf_ca", 10, 4, 8, 0, "survey", False, 0, 2)
    b = bench.make_workload("usv_model_pf_ca", 10, 4, 8, 0, "survey", False, 1, 2)
    assert a[1] == b[1] == 8 and not np.array_equal(a[0]["x0"], b[0]["x0"])


def test_value_counts_converged_solves_and_the_line_states_its_departures():
    """round-4 lines: `value` is the rate of CONVERGED solves, the rate counting every solve sits beside it, and
    config.workload names the generator's departures (obstacle clip, initial guess)."""
    import glob
    lines = sorted(glob.glob(os.path.join(ROOT, "profiles", "r04_*bench*.json")))
    assert lines, "no round-4 bench line committed under profiles/"
    for fn in lines:
        d = json.load(open(fn))
        ws = d["workload_stats"]
        allv, unc = ws["solves_per_s_counting_unconverged_ones"], ws["unconverged_solves_in_timed_region"]
        total = d["config"]["instances_total"] * d["steps"]
        assert abs(d["value"] - allv * (1.0 - unc / total)) <= 1e-9 * allv and d["value"] <= allv
        if d["config"]["ocp"] == "usv_model_pf_ca":
            w = d["config"]["workload"]
            assert "obstacle clip" in w and "initial guess" in w and "RK4 steps" in w
        if d.get("parity"):
            p = d["parity"]
            for k in ("rel_err_x", "rel_err_u", "count_above_1e-5", "compared", "above_1e-5_without_kkt_certificate_or_beyond_5e-3"):
                assert k in p, k
            assert p["above_1e-5_without_kkt_certificate_or_beyond_5e-3"] == 0


def test_round5_lines_carry_the_median_and_the_survey_verbatim_line_exists():
    """every line reports the median of its step times beside the mean ("report median"), and the survey's
    generator to the letter has a committed line of its own next to the builder's variant."""
    import glob
    lines = sorted(glob.glob(os.path.join(ROOT, "profiles", "r05_g_bench*_plain.json")))
    assert len(lines) >= 8
    for fn in lines:
        d = json.load(open(fn))
        lo, hi = d["ms_per_step_min_max"]
        assert lo <= d["ms_per_step_median"] <= hi and lo <= d["ms_per_step"] * 1.001 and d["ms_per_step"] <= hi * 1.001, fn
        ws = d["workload_stats"]
        total = d["config"]["instances_total"] * d["steps"]
        assert abs(d["value"] - ws["solves_per_s_counting_unconverged_ones"] * (1.0 - ws["unconverged_solves_in_timed_region"] / total)) <= 1e-9 * d["value"]
        assert ws["unconverged_counted_over_steps"] == min(d["steps"], 64)
    v = json.load(open(os.path.join(ROOT, "profiles", "r05_g_bench_survey_verbatim_plain.json")))
    w = v["config"]["workload"]
    assert "survey_verbatim" in w and "no obstacle clip" in w and "x_k = x0" in w and "mask 0x3fff" in w
    assert v["parity"]["above_1e-5_without_kkt_certificate_or_beyond_5e-3"] == 0
    ws = v["workload_stats"]
    assert 0.03 < ws["status_nonzero_frac"] < 0.12 and 0.03 < ws["qp_not_converged_frac"] < 0.12 and len(ws["qp_iter_histogram"]) > 20
    b = json.load(open(os.path.join(ROOT, "profiles", "r05_g_bench_plain.json")))
    assert "obstacle clip" in b["config"]["workload"] and b["value"] > v["value"]
    # the bench's parity leg against the oracle with HPIPM's options
    it = json.load(open(os.path.join(ROOT, "profiles", "r05_g_bench_oracle_itref2_plain.json")))["parity"]
    assert it["oracle_options"] == {"itref_corr_max": 2} and it["count_above_1e-5"] == 0 and it["rel_err_per_instance"]["max"] < 1e-5
    one = json.load(open(os.path.join(ROOT, "profiles", "r05_g_bench_oracle_cpc_device_plain_plain.json")))["parity"]
    both = json.load(open(os.path.join(ROOT, "profiles", "r05_g_bench_cpc_both_sides_plain.json")))["parity"]
    assert one["count_above_1e-5"] > 20 and both["count_above_1e-5"] <= 3 and both["above_1e-5_without_kkt_certificate_or_beyond_5e-3"] == 0


def test_rank_binding_falls_back_to_an_even_share_and_is_recorded():
    """bench.py --bind-numa: every rank pins its host threads to its GPU's NUMA node (sysfs); without a GPU / NUMA information the usable CPUs
    are split evenly among the ranks.  Single-rank runs are left alone by default."""
    before = os.sched_getaffinity(0)
    try:
        assert bench.bind_to_gpu_numa(0, 1, "auto") == "not bound" and os.sched_getaffinity(0) == before
        assert bench.bind_to_gpu_numa(0, 2, "off") == "not bound"
        if len(before) >= 2:
            msg = bench.bind_to_gpu_numa(1, 2, "on")
            now = os.sched_getaffinity(0)
            assert now and now <= before and ("even share" in msg or "NUMA node" in msg), msg
            if "even share" in msg:
                assert len(now) == len(before) // 2 and min(now) > min(before)
    finally:
        os.sched_setaffinity(0, before)


def test_scale_run_script_names_every_sharded_config():
    """tools/scale_run.sh: N = 1, 2, 4, 8 x (weak scaling, BASELINE configs[3] as ONE batch of 262 144, configs[4]'s OCP and batch), the
    environment RCCL needs on this driver, the per-rank placement; the summary keeps ranks_seen / per-rank extremes / all-gather times."""
    sh = open(os.path.join(ROOT, "tools", "scale_run.sh")).read()
    for needle in ("for n in 1 2 4 8", "--global-batch 262144", "--global-batch 65536 --horizon 80 --obstacles 20 --moving", "HSA_ENABLE_IPC_MODE_LEGACY=0",
                   "ranks_seen", "per_rank_ms_per_step_min_max", "allgather_ms", "efficiency_vs_n1", "--showtoponuma"):
        assert needle in sh, needle
    assert os.access(os.path.join(ROOT, "tools", "scale_run.sh"), os.X_OK)


def test_round6_lines_carry_the_profile_its_spread_and_the_survey_verbatim_region():
    """the default line names the QP solver profile, compares against the oracle under that profile, prints what the
    profile choice moves on the device (parity.profile_spread) and the survey's generator to the letter as a second timed region."""
    d = json.load(open(os.path.join(ROOT, "profiles", "r06_g_bench_plain.json")))
    assert d["config"]["hpipm_mode"].startswith("BALANCE") and d["roofline"]["kernel"] == "usv_qp_rti"   # (one launch: no hand-over at 65 536)
    p = d["parity"]
    assert p["oracle_options"] == {"hpipm_mode": "BALANCE"} and p["count_above_1e-5"] == 0 and p["rel_err_per_instance"]["max"] < 1e-5
    assert p["above_1e-5_without_kkt_certificate_or_beyond_5e-3"] == 0 and p["kkt_certified_frac"] == 1.0
    sp = p["profile_spread"]
    assert sp["device_profile"] == "BALANCE" and sp["against_device_profile"] == "R04" and sp["compared"] > 2000 and 1e-3 < sp["max"] < 0.1
    sv = d["survey_verbatim"]
    assert 0.7 * d["value"] < sv["value"] < d["value"] and 0.03 < sv["qp_not_converged_frac"] < 0.12 and sv["steps"] == 10
    assert sv["parity"]["rule_violations"] == 0 and sv["parity"]["status_agreement_frac"] > 0.99
    ws = d["workload_stats"]
    total = d["config"]["instances_total"] * d["steps"]
    assert ws["unconverged_counted_over_steps"] == d["steps"]
    assert abs(d["value"] - ws["solves_per_s_counting_unconverged_ones"] * (1.0 - ws["unconverged_solves_in_timed_region"] / total)) <= 1e-9 * d["value"]
    assert len(d["config"]["per_rank_ms_per_step"]) == 1 and d["config"]["host_binding_rank0"] == "not bound"
    # the profile of rounds 1 - 5 on the same kernels, and the oracle without its refinement: the outliers the default no longer has
    r04 = json.load(open(os.path.join(ROOT, "profiles", "r06_g_bench_profile_r04_plain.json")))
    assert r04["config"]["hpipm_mode"].startswith("R04") and r04["parity"]["count_above_1e-5"] >= 1 and r04["value"] < d["value"]
    spd = json.load(open(os.path.join(ROOT, "profiles", "r06_g_bench_oracle_speed_plain.json")))["parity"]
    assert spd["oracle_options"]["hpipm_mode"] == "SPEED" and spd["count_above_1e-5"] >= 1
    # the mid-size batches run with the follow-up kernel beside the launch
    b = json.load(open(os.path.join(ROOT, "profiles", "r06_g_bench_b8192_plain.json")))
    assert "usv_qp_resume" in b["roofline"]["kernel_ms"] and b["value"] > 560e3


def test_scale_script_lines_have_the_plain_lines_keys():
    """tools/scale_run.sh run on the one GPU a builder's box has (`tools/scale_run.sh out 1`; profiles/r06_scale_n1_*): its N = 1 lines are bench
    lines like the plain one - same keys, same config keys -, the sharded configs name themselves, and the summary has one row per kind.  (The
    N = 2, 4, 8 rows are the driver's / an 8-GPU node's to fill: no scaling curve has been measured.)"""
    plain = json.load(open(os.path.join(ROOT, "profiles", "r06_g_bench_plain.json")))
    weak = json.load(open(os.path.join(ROOT, "profiles", "r06_scale_n1_weak.json")))
    # (the script's lines were taken before bench.py grew its third region: the plain line has that one key more)
    assert set(weak) <= set(plain) and set(plain) - set(weak) <= {"configs4_condensed"}
    assert set(weak["config"]) == set(plain["config"]) and set(weak["roofline"]) == set(plain["roofline"])
    assert weak["n_gpus"] == 1 and weak["config"]["ranks_seen"] == 1 and "configs[2]" in weak["config"]["workload"]
    c3 = json.load(open(os.path.join(ROOT, "profiles", "r06_scale_n1_cfg3.json")))
    c4 = json.load(open(os.path.join(ROOT, "profiles", "r06_scale_n1_cfg4.json")))
    assert c3["config"]["instances_total"] == 262144 and "ONE seed-1234 batch of 262144" in c3["config"]["workload"]
    assert c4["config"]["instances_total"] == 65536 and c4["config"]["horizon"] == 80 and c4["config"]["obstacles"] == 20 and "moving" in c4["config"]["workload"]
    summ = json.load(open(os.path.join(ROOT, "profiles", "r06_scale_n1_summary.json")))
    assert set(summ) == {"weak", "cfg3", "cfg4"} and all(len(v) == 1 and v[0]["n_gpus"] == 1 and v[0]["efficiency_vs_n1"] == 1.0 for v in summ.values())


def test_condensed_line_of_round_6_meets_the_bar_set_for_it():
    """usv_qp_cond at BASELINE configs[4]'s per-GPU share (8192 instances, N = 80 -> qp_cond_N = 10) at most 110 ms per launch
    (205 ms at the start of round 6): the committed line and the rocprofv3 statistics of the same command."""
    b = json.load(open(os.path.join(ROOT, "profiles", "r06_g_bench_cfg4_b8192_condN10_plain.json")))
    assert b["config"]["qp_solver_cond_N"] == 10 and b["config"]["horizon"] == 80 and b["config"]["obstacles"] == 20 and b["config"]["instances_total"] == 8192
    assert b["roofline"]["kernel"] == "usv_qp_cond" and b["roofline"]["kernel_ms"]["usv_qp_cond"] <= 110.0 and b["value"] >= 70e3
    stats = [l for l in open(os.path.join(ROOT, "profiles", "r06_g_cond_kernel_stats.csv")) if "usv_qp_cond<usv::ModelM2, 2, false, 256, 8, 7>" in l]
    assert len(stats) == 1 and float(stats[0].rsplit('"', 1)[1].split(",")[3]) <= 110e6      # (average duration, ns: the instantiation with the compile-time block shape)
    unc = json.load(open(os.path.join(ROOT, "profiles", "r06_g_bench_cfg4_b8192_per_gpu_plain.json")))
    assert unc["value"] > b["value"]     # (the uncondensed sweep is still the faster formulation at this shape: it stays the default)


def test_default_line_carries_configs4_with_its_condensing_applied():
    """The default line of the final evidence set (profiles/r06_g_bench_plain.json) has the third region: BASELINE configs[4] at its per-GPU share solved
    with qp_solver_cond_N = 10 and uncondensed, the two solutions compared on tick 0; and the PMC table has the traffic of that library."""
    d = json.load(open(os.path.join(ROOT, "profiles", "r06_g_bench_plain.json")))
    pmc = [e for e in json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json"))) if e["round"] == "r06_g"]
    assert len(pmc) == 1 and pmc[0]["lib_sha256"] == d["config"]["lib_sha256"] and abs(pmc[0]["hbm_bytes_per_launch"] - 353e9) < 5e9
    c = d["configs4_condensed"]
    assert c["condensed"]["kernel"] == "usv_qp_cond" and c["uncondensed"]["kernel"] == "usv_qp_rti"
    assert c["condensed"]["ms_per_step"] <= 110.0 and c["condensed"]["value"] >= 70e3 and c["uncondensed"]["value"] > c["condensed"]["value"]
    t = c["tick0_condensed_vs_uncondensed"]
    assert t["status_agreement_frac"] >= 0.999 and t["same_iteration_count_frac"] >= 0.999 and t["converged_on_both_sides_frac"] >= 0.9
    assert t["rel_err_per_instance"]["p50"] <= 1e-9 and t["rel_err_per_instance"]["p99"] <= 1e-5 and t["rel_err_per_instance"]["max"] <= 1e-3
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert "def configs4_condensed_leg" in src and '"configs4_condensed": configs4_condensed' in src
