#!/usr/bin/env python3
"""Do the placements / mappings of one library return the same BITS?  For each shape: the throughput mapping with the planes in HBM (the
reference point), the same with the planes in LDS, the aux plane in HBM, the latency mapping with one and with four waves per instance,
each solving the same closed loop from identical inputs tick by tick.  Prints, per variant and tick, whether x / u / pi / lam / t /
qp_iter equal the reference point's bit for bit (and the largest relative difference when they do not).
usage (GPU): python tools/bitident_probe.py [quick]"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from mpc_collisionavoidance_amd import BatchOcpSolver, scenario, usv_models

VARIANTS = [("hbm16", (("wide", 0), ("lds_workspace", 0))),
            ("hbm16_auxhbm", (("wide", 0), ("lds_workspace", 0), ("aux_in_lds", 0))),
            ("lds16", (("wide", 0), ("lds_workspace", 1))),
            ("wide1", (("wide", 1), ("wide_waves", 1))),
            ("wide4", (("wide", 1), ("wide_waves", 4)))]
SHAPES = [("usv_model_pf_ca", 20, 3, 256, 4), ("usv_model_pf_ca", 40, 10, 128, 3), ("usv_model_guidance_ca1", 20, 3, 256, 3),
          ("usv_model_guidance_ca1", 40, 10, 64, 3), ("usv_model", 20, 0, 128, 3), ("usv_model_guidance_ca1", 100, 8, 16, 2),
          ("usv_model_pf_ca", 100, 4, 8, 2),
          # two obstacle chunks (K = 17 .. 32; the four-wave form since r05_e)
          ("usv_model_pf_ca", 80, 20, 64, 2), ("usv_model_guidance_ca1", 40, 20, 128, 3), ("usv_model_pf_ca", 24, 32, 32, 3),
          ("usv_model_guidance_ca1", 60, 17, 16, 3)]
if len(sys.argv) > 1 and sys.argv[1] == "quick":
    SHAPES = SHAPES[:3]


def make(name, N, K, B, opts):
    wl = scenario.make_bench_batch(name, N, K, B, seed=1234)
    ocp = usv_models.make_ocp(name, N * scenario.BENCH_DT, N, K if name != "usv_model" else None)
    ocp.solver_options.sim_method_num_steps = scenario.BENCH_SIM_STEPS[name]
    s = BatchOcpSolver(ocp, B)
    scenario.load_into(s, wl)
    if K > 0:
        s.set_option("static_obstacles", 1)
    s.set_option("disturbance_mask", scenario.NOISE_MASK[name])
    for k, v in opts:
        s.set_option(k, v)
    return s


for name, N, K, B, ticks in SHAPES:
    sol = [(tag, make(name, N, K, B, opts)) for tag, opts in VARIANTS]
    print("== %s N=%d K=%d B=%d" % (name, N, K, B))
    for t in range(ticks):
        ref = None
        for tag, s in sol:
            st = s.solve()
            out = {f: s.get_all(f) for f in ("x", "u", "pi", "lam", "t")}
            out["qp_iter"] = s.get_int("qp_iter"); out["status"] = st.copy()
            if ref is None:
                ref = out
                print("  tick %d %-13s mapping %d  (reference point; mean iterations %.1f)" % (t, tag, s.last_mapping(), out["qp_iter"].mean()))
                continue
            bad = []
            for f in ("status", "qp_iter", "x", "u", "pi", "lam", "t"):
                if not np.array_equal(out[f], ref[f]):
                    d = np.abs(out[f].astype(float) - ref[f].astype(float)).max() / max(1.0, np.abs(ref[f].astype(float)).max())
                    bad.append("%s %.1e" % (f, d))
            print("  tick %d %-13s mapping %d  %s" % (t, tag, s.last_mapping(), "BIT-IDENTICAL" if not bad else "differs: " + ", ".join(bad)))
        # every variant continues from the reference point's state
        s0 = sol[0][1]
        s0.advance(1e-3, seed=50 + t)
        s0.sync()
        x0, xa, ua = s0.get("x0", 0), s0.get_all("x"), s0.get_all("u")
        for tag, s in sol[1:]:
            s.set("x0", 0, x0); s.set_all("x", xa); s.set_all("u", ua)
    for _, s in sol:
        s.close()
