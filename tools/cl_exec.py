import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa
from mpc_collisionavoidance_amd import BatchOcpSolver, scenario, usv_models
name, B, N, K = "usv_model_pf_ca", 4, 40, 10
wl = scenario.make_bench_batch(name, N, K, B, seed=1234)
dt, steps = scenario.BENCH_DT, scenario.BENCH_SIM_STEPS[name]
ocp = usv_models.make_ocp(name, N * dt, N, K)
ocp.solver_options.sim_method_num_steps = steps
s = BatchOcpSolver(ocp, B)
scenario.load_into(s, wl)
s.set_option("static_obstacles", 1)
s.set_option("host_mirror", 0)
s.set_option("aux_in_lds", 0)
s.closed_loop(1, 0.0, 1); s.sync()
r = s.get("nlp_res", 0).reshape(-1)
print("before: exec %08x%08x xlane(t0) %g xlane(t5) %g" % (int(r[0]), int(r[1]), r[2], r[3]))
print("after : exec %08x%08x xlane(t0) %g xlane(t5) %g ; t5: ulane %g valid %g isPX %g hasb %g N %g Kn %g nB %g itmax %g" % (int(r[4]), int(r[5]), r[6], r[7], *r[8:16]))
print(s.get_int("qp_iter"))
