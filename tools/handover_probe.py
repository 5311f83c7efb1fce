#!/usr/bin/env python3
"""What does handing the long runners of the headline launch over to the latency mapping buy?  (GPU, development aid)
BASELINE configs[2] closed loop exactly as bench.py runs it, for a list of handover_iter values: ms per step (wall, 20 steps), the QP launch
+ follow-up launch as HIP events see them, instances handed over per tick.  usage: python tools/handover_probe.py [model] [values...]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mpc_collisionavoidance_amd import BatchOcpSolver, scenario, usv_models
name = sys.argv[1] if len(sys.argv) > 1 else "usv_model_pf_ca"
vals = [int(v) for v in sys.argv[2:]] or [0, 30, 24, 20, 16, 12, 0]
N, K, B = 40, 10, 65536
wl = scenario.make_bench_batch(name, N, K, B, seed=1234)
ocp = usv_models.make_ocp(name, N * scenario.BENCH_DT, N, K)
ocp.solver_options.sim_method_num_steps = scenario.BENCH_SIM_STEPS[name]
for hv in vals:
    s = BatchOcpSolver(ocp, B)
    scenario.load_into(s, wl)
    s.set_option("static_obstacles", 1)
    s.set_option("disturbance_mask", scenario.NOISE_MASK[name])
    s.set_option("handover_iter", hv)
    for w in range(3):
        s.solve_async(); s.advance(1e-3, seed=1000 + w)
    s.sync()
    t0 = time.perf_counter()
    for k in range(20):
        s.solve_async(); s.advance(1e-3, seed=2000 + k)
    s.sync()
    el = (time.perf_counter() - t0) / 20 * 1e3
    lin, qp = s.kernel_ms(20)
    ho = s.handover_counts(20)
    tk = s.tick_ms(20)
    print("handover_iter %3d: %.2f ms per step (median tick %.2f), QP launch + follow-up %.2f ms, lineariser on the main stream %.2f ms, handed over per tick mean %.0f max %d, %.0f solves/s"
          % (hv, el, np.median(tk), qp.mean(), lin.mean(), ho.mean(), ho.max(), B / el * 1e3), flush=True)
    s.close()
