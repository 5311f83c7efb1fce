import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa
from mpc_collisionavoidance_amd import BatchOcpSolver, scenario, usv_models
name, B, N, K = "usv_model_pf_ca", int(sys.argv[1]), 40, 10
T = int(sys.argv[2])
P = lambda *a: print(*a, flush=True)
wl = scenario.make_bench_batch(name, N, K, B, seed=1234)
dt, steps = scenario.BENCH_DT, scenario.BENCH_SIM_STEPS[name]
ocp = usv_models.make_ocp(name, N * dt, N, K)
ocp.solver_options.sim_method_num_steps = steps
s = BatchOcpSolver(ocp, B)
scenario.load_into(s, wl)
s.set_option("static_obstacles", 1)
s.set_option("host_mirror", 0)
P("created")
t0 = time.time()
s.closed_loop(T, 1e-3, 77)
P("enqueued %.3f s" % (time.time() - t0))
import ctypes as C
for i in range(3):
    time.sleep(2.0)
    ctr = (C.c_int * 9)()
    s._lib.usvmpc_debug_counters(s._h, ctr)
    P("counters [linearised, lin tickets, handed over, waves, abort, -, -, -, qp tickets]:", list(ctr))
try:
    s.sync()
    P("synced %.3f s" % (time.time() - t0))
except Exception as e:
    P("sync error:", e)
P(s.get_int("qp_iter")[:8], s.get_int("status")[:8])
