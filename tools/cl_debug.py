import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa
from mpc_collisionavoidance_amd import BatchOcpSolver, scenario, usv_models
name = sys.argv[1] if len(sys.argv) > 1 else "usv_model_pf_ca"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4
T = int(sys.argv[3]) if len(sys.argv) > 3 else 1
N, K = 40, 10
wl = scenario.make_bench_batch(name, N, K, B, seed=1234)
dt, steps = scenario.BENCH_DT, scenario.BENCH_SIM_STEPS[name]
ocp = usv_models.make_ocp(name, N * dt, N, K)
ocp.solver_options.sim_method_num_steps = steps
def make(fused, **opt):
    s = BatchOcpSolver(ocp, B)
    scenario.load_into(s, wl)
    s.set_option("static_obstacles", 1)
    s.set_option("host_mirror", 0)
    s.set_option("fused_closed_loop", fused)
    for k, v in opt.items():
        s.set_option(k, v)
    return s
for opts in ({}, {"aux_in_lds": 0}, {"merge_box_rows": 0, "aux_in_lds": 0}):
    a, b = make(0, **opts), make(1, **opts)
    a.closed_loop(T, 1e-3, 77); a.sync()
    b.closed_loop(T, 1e-3, 77); b.sync()
    print(opts)
    for f in ("status", "qp_iter", "qp_status"):
        print("  ", f, a.get_int(f)[:8], b.get_int(f)[:8])
    print("   res", a.get("res", 0)[:4], b.get("res", 0)[:4])
    xa, xb = a.get_all("x"), b.get_all("x")
    print("   x max diff", np.abs(xa - xb).max(), "x0 diff", np.abs(a.get("x0", 0) - b.get("x0", 0)).max())
# planes: lineariser outputs of the two paths for B instances, T = 1
a, b = make(0), make(1)
a.solve_async(); a.sync()
b.closed_loop(1, 0.0, 1); b.sync()
wa, wb = a.debug_workspace(), b.debug_workspace()
npt = wa.shape[2]
print("npt", npt, "shape", wa.shape)
# the lineariser's planes are the last ones before soft-box planes: RB0, GQ, MAT.. ; compare every plane of group 0..B-1 (identity map at the first solve)
for e in range(npt):
    d = np.abs(wa[:, :B, e] - wb[:, :B, e])
    if d.max() > 0:
        k, g, l = np.unravel_index(np.argmax(d), d.shape)
        print("  plane %2d differs: max %.3e at stage %d group %d lane %d (%.6g vs %.6g); stages differing %s" % (e, d.max(), k, g, l, wa[k, g, e, l], wb[k, g, e, l], np.where(d.reshape(d.shape[0], -1).max(axis=1) > 0)[0][:12]))
