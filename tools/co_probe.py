#!/usr/bin/env python3
"""A/B of the hand-over's follow-up kernel BESIDE the draining launch (option "handover_co") on the bench's closed loop:
tools/co_probe.py <model> <N> <K> <B> [ticks]  ->  ms per tick for (co off / on) x thresholds, how many instances were handed over
and how many of them the co-resident kernel finished."""
import sys, time
import numpy as np
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from mpc_collisionavoidance_amd import BatchOcpSolver, scenario, usv_models

name, N, K, B = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
ticks = int(sys.argv[5]) if len(sys.argv) > 5 else 20
cases = [("off", dict(handover_iter=0))]
for thr in (24, 20, 16):
    cases.append(("behind %d" % thr, dict(handover_iter=thr, handover_co=0)))
for thr in (24, 20, 16):
    for wgs in (64, 128, 256, 0):
        cases.append(("beside %d w%d" % (thr, wgs), dict(handover_iter=thr, handover_co=1, handover_co_wgs=wgs)))
extra = [kv.split("=") for kv in sys.argv[6:]]
res = {}
for rep in range(2):
    for tag, opts in cases:
        wl = scenario.make_bench_batch(name, N, K, B, seed=1234)
        ocp = usv_models.make_ocp(name, N * scenario.BENCH_DT, N, K if name != "usv_model" else None)
        ocp.solver_options.sim_method_num_steps = scenario.BENCH_SIM_STEPS[name]
        s = BatchOcpSolver(ocp, B)
        scenario.load_into(s, wl)
        if K > 0:
            s.set_option("static_obstacles", 1)
        s.set_option("disturbance_mask", scenario.NOISE_MASK[name])
        for k, v in opts.items():
            s.set_option(k, v)
        for k, v in extra:
            s.set_option(k, float(v))
        for t in range(4):
            s.solve_async(); s.advance(1e-3, seed=100 + t)
        s.sync()
        t0 = time.perf_counter()
        for t in range(ticks):
            s.solve_async(); s.advance(1e-3, seed=t)
        s.sync()
        ms = (time.perf_counter() - t0) * 1e3 / ticks
        n = min(ticks, 16)
        handed = s.handover_counts(n)
        fin, tmo = s.handover_co_counts(n)
        km = s.kernel_ms(n)
        fu = s.followup_ms(n)
        res.setdefault(tag, []).append(ms)
        print("%-16s rep %d  %.2f ms/tick  %.0f k solves/s  handed %.0f  beside %.0f  timeouts %.0f  qp %.2f ms  follow-up behind %.2f ms"
              % (tag, rep, ms, B / ms, handed.mean(), fin.mean(), tmo.mean(), np.mean(km[1]), fu.mean()), flush=True)
        s.close()
print({k: round(min(v), 2) for k, v in res.items()})
