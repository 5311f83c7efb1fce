import sys, os
import numpy as np
sys.path.insert(0, "/root/repo")
from mpc_collisionavoidance_amd import BatchOcpSolver, scenario, usv_models
name, N, K = "usv_model_pf_ca", 20, 3
for B in (1, 2, 3, 8, 64):
    wl = scenario.make_bench_batch(name, N, K, B)
    ocp = usv_models.make_ocp(name, N * 0.05, N, K); ocp.solver_options.sim_method_num_steps = 5
    outs = []
    for mode in (0, 1, 1):
        s = BatchOcpSolver(ocp, B); scenario.load_into(s, wl); s.set_option("static_obstacles", 1); s.set_option("lds_workspace", mode)
        st = s.solve()
        outs.append((s.get_all("x"), s.get_all("u"), st.copy(), s.get_int("qp_iter")))
        s.close()
    d01 = np.abs(outs[0][0] - outs[1][0]).max(); d12 = np.abs(outs[1][0] - outs[2][0]).max()
    print("B", B, "max|x hbm - x lds|", d01, "lds run-to-run", d12, "status", outs[0][2][:8], outs[1][2][:8], "iters", outs[0][3][:8], outs[1][3][:8])
