"""Pipelined lineariser against the un-pipelined sequence on the headline workload itself (GPU): 65 536 instances, 60 closed-loop ticks,
every 10th tick's iterate, statuses, iteration counts and hand-over compared bit for bit; plus how many instances the ahead-of-time
pass had to leave to the fix-up pass (the fix-up lineariser's time says it) and the step time of both."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa
from mpc_collisionavoidance_amd import BatchOcpSolver, scenario, usv_models
name = sys.argv[1] if len(sys.argv) > 1 else "usv_model_pf_ca"
N, K, B, ticks = 40, 10, 65536, int(sys.argv[2]) if len(sys.argv) > 2 else 60
wl = scenario.make_bench_batch(name, N, K, B, seed=1234)
ocp = usv_models.make_ocp(name, N * scenario.BENCH_DT, N, K)
ocp.solver_options.sim_method_num_steps = scenario.BENCH_SIM_STEPS[name]
res = {}
for pipe in (1, 0):
    s = BatchOcpSolver(ocp, B)
    scenario.load_into(s, wl)
    s.set_option("static_obstacles", 1)
    s.set_option("disturbance_mask", scenario.NOISE_MASK[name])
    s.set_option("pipeline_linearize", pipe)
    snaps = []
    s.sync(); t0 = time.perf_counter()
    for t in range(ticks):
        s.solve_async(); s.advance(1e-3, seed=100 + t)
        if (t + 1) % 10 == 0:
            s.sync()
            snaps.append((s.get_all("x"), s.get_all("u"), s.get_int("status").copy(), s.get_int("qp_iter").copy(), s.get("x0", 0)))
    s.sync(); el = time.perf_counter() - t0
    lin, qp = s.kernel_ms(min(ticks, 50))
    print("pipeline_linearize=%d: %.2f ms per tick (snapshots included), lineariser on the main stream %.2f ms, QP %.2f ms" % (pipe, el / ticks * 1e3, lin.mean(), qp.mean()), flush=True)
    res[pipe] = snaps
    s.close()
ok = True
for i, (a, b) in enumerate(zip(res[1], res[0])):
    same = [bool(np.array_equal(p, q)) for p, q in zip(a, b)]
    print("tick %d: x u status qp_iter x0 identical:" % (10 * (i + 1)), same, flush=True)
    ok = ok and all(same)
print("BIT-IDENTICAL" if ok else "MISMATCH")
sys.exit(0 if ok else 1)
