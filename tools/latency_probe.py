"""Small-batch latency probe (GPU): ms per RTI solve of the whole batch for a range of batch sizes - the throughput mapping (four
instances per wavefront) with the workspace in HBM and in LDS, and the latency mapping (option "wide": one instance per wavefront;
"wide_waves" = 4: per workgroup of four), and what the library picks by default against the best of those.
python tools/latency_probe.py [model] [N] [K] [batch sizes, comma separated]"""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpc_collisionavoidance_amd import BatchOcpSolver, scenario, usv_models
name = sys.argv[1] if len(sys.argv) > 1 else "usv_model_pf_ca"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 20
K = int(sys.argv[3]) if len(sys.argv) > 3 else 3
sizes = [int(v) for v in sys.argv[4].split(",")] if len(sys.argv) > 4 else [1, 64, 256, 512, 1024, 2048, 4096]
TICKS = 20
for B in sizes:
    wl = scenario.make_bench_batch(name, N, K, B)
    ocp = usv_models.make_ocp(name, N * scenario.BENCH_DT, N, K)
    ocp.solver_options.sim_method_num_steps = scenario.BENCH_SIM_STEPS[name]
    res = {}
    for mode, opts in (("hbm", (("lds_workspace", 0), ("wide", 0))), ("lds", (("lds_workspace", 1), ("wide", 0))), ("wide", (("wide", 1), ("wide_waves", 1))),
                       ("wide4", (("wide", 1), ("wide_waves", 4))), ("default", ())):
        s = BatchOcpSolver(ocp, B)
        scenario.load_into(s, wl)
        s.set_option("static_obstacles", 1)
        s.set_option("disturbance_mask", scenario.NOISE_MASK[name])
        for k, v in opts:
            s.set_option(k, v)
        if mode == "wide" and os.environ.get("MAXW"):
            s.set_option("max_waves", float(os.environ["MAXW"]))   # (cap on the resident wide waves: what does a second round cost?)
        for t in range(3):
            s.solve_async(); s.advance(1e-3, seed=t)
        s.sync()
        t0 = time.perf_counter()
        for t in range(TICKS):
            s.solve_async(); s.advance(1e-3, seed=10 + t)
        s.sync()
        wall = (time.perf_counter() - t0) / TICKS * 1e3
        lin, qp = s.kernel_ms(TICKS)
        res[mode] = (wall, qp.mean(), lin.mean(), s.get_all("x"), s.get_int("qp_iter"), s.last_mapping())
        s.close()
    it = res["hbm"][4]
    best = min(res[m][0] for m in ("hbm", "lds", "wide", "wide4"))
    print("%s N=%d K=%d B %5d | HBM %.3f ms/tick (qp %.3f, lin %.3f) | LDS %.3f (qp %.3f) | wide[%d] %.3f (qp %.3f) | four waves[%d] %.3f (qp %.3f) | qp_iter mean %.1f max %d"
          " | default[%d] %.3f = %.2f x the best column"
          % (name, N, K, B, res["hbm"][0], res["hbm"][1], res["hbm"][2], res["lds"][0], res["lds"][1], res["wide"][5], res["wide"][0], res["wide"][1],
             res["wide4"][5], res["wide4"][0], res["wide4"][1], it.mean(), it.max(), res["default"][5], res["default"][0], res["default"][0] / best), flush=True)
