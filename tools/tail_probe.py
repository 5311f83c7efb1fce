"""How much of the QP launch is its tail (the last long-running instances draining while the rest of the device idles)?
(GPU, development aid)  Caps the IPM iterations (qp_solver_iter_max) and reports kernel time against the pass count: a launch
whose time is set by its tail shortens by (cap difference) x (lone-wave pass time) when the cap drops, a launch that is busy
throughout shortens only in proportion to the work removed."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa
from mpc_collisionavoidance_amd import BatchOcpSolver, scenario, usv_models
name, N, K, B = "usv_model_pf_ca", 40, 10, 65536
wl = scenario.make_bench_batch(name, N, K, B, seed=1234)
for cap in (50, 35, 25, 18):
    ocp = usv_models.make_ocp(name, N * scenario.BENCH_DT, N, K)
    ocp.solver_options.sim_method_num_steps = scenario.BENCH_SIM_STEPS[name]
    ocp.solver_options.qp_solver_iter_max = cap
    s = BatchOcpSolver(ocp, B)
    scenario.load_into(s, wl)
    s.set_option("static_obstacles", 1)
    s.set_option("disturbance_mask", scenario.NOISE_MASK[name])
    for w in range(3):
        s.solve_async(); s.advance(1e-3, seed=1000 + w)
    for k in range(8):
        s.solve_async(); s.advance(1e-3, seed=2000 + k)
    s.sync()
    lin, qp = s.kernel_ms(8)
    qi = s.get_int("qp_iter")
    passes = (qi + 1).sum() / 8192.0
    print("iter cap %2d: usv_qp_rti %.2f ms, mean iterations %.2f, at the cap %.4f, row-passes / 8192 rows = %.1f, ms per ideal pass %.3f"
          % (cap, qp.mean(), qi.mean(), (qi >= cap).mean(), passes, qp.mean() / passes), flush=True)
    s.close()
