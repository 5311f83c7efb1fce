"""Does splitting the batch over S independent handles (each with its own stream) hide the tail of the persistent QP launch?
(GPU, development aid)  python tools/two_stream_probe.py [S ...]
Every handle runs the closed loop of its own slice; the slices are stepped round-robin, so while one slice's launch drains its
last long-running instances the other slices' kernels fill the compute units that have become free."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa
from mpc_collisionavoidance_amd import BatchOcpSolver, scenario, sharding, usv_models
name, N, K, B, steps, warm = "usv_model_pf_ca", 40, 10, 65536, 20, 3
wl = scenario.make_bench_batch(name, N, K, B, seed=1234)
ocp = usv_models.make_ocp(name, N * scenario.BENCH_DT, N, K)
ocp.solver_options.sim_method_num_steps = scenario.BENCH_SIM_STEPS[name]
for S in [int(a) for a in sys.argv[1:]] or [1, 2, 4]:
    hs = []
    for r in range(S):
        lo, hi = sharding.shard_bounds(B, S, r)
        s = BatchOcpSolver(ocp, hi - lo)
        scenario.load_into(s, sharding.split_workload(wl, S, r))
        s.set_option("static_obstacles", 1)
        s.set_option("disturbance_mask", scenario.NOISE_MASK[name])
        if os.environ.get("MAXW"):   # resident waves per handle (e.g. 2048 / S: the handles share the device instead of queueing behind each other)
            s.set_option("max_waves", float(os.environ["MAXW"]) if float(os.environ["MAXW"]) > 0 else 2048 // S)
        hs.append(s)
    for w in range(warm):
        for s in hs:
            s.solve_async(); s.advance(1e-3, seed=1000 + w)
    for s in hs:
        s.sync()
    t0 = time.perf_counter()
    for k in range(steps):
        for s in hs:
            s.solve_async(); s.advance(1e-3, seed=2000 + k)
    for s in hs:
        s.sync()
    el = time.perf_counter() - t0
    qi = np.concatenate([s.get_int("qp_iter") for s in hs])
    st = np.concatenate([s.get_int("status") for s in hs])
    print("handles %d: %.2f ms per step, %.0f solves/s, qp_iter mean %.2f, status != 0: %.4f" % (S, el / steps * 1e3, B * steps / el, qi.mean(), (st != 0).mean()), flush=True)
    for s in hs:
        s.close()
