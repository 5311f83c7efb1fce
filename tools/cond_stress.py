"""Development aid: run-to-run / team-count determinism of the condensing kernel at scale.  usage: python tools/cond_stress.py"""
import sys, numpy as np
sys.path.insert(0, ".")
from mpc_collisionavoidance_amd import BatchOcpSolver, scenario, usv_models
name, N, K, B, N2 = "usv_model_pf_ca", 80, 20, 8192, 10
wl = scenario.make_bench_batch(name, N, K, B, moving=True, seed=99)
dt, steps = scenario.BENCH_DT, scenario.BENCH_SIM_STEPS[name]
def mk(extra=()):
    ocp = usv_models.make_ocp(name, N * dt, N, K); ocp.solver_options.sim_method_num_steps = steps; ocp.solver_options.qp_solver_cond_N = N2
    s = BatchOcpSolver(ocp, B); scenario.load_into(s, wl)
    for k, v in extra: s.set_option(k, v)
    return s
a, a2, b = mk(), mk(), mk((("max_waves", 311),))
for tick in range(5):
    st = [s.solve() for s in (a, a2, b)]
    xs = [s.get_all("x") for s in (a, a2, b)]
    qs = [s.get_int("qp_status") for s in (a, a2, b)]
    qi = [s.get_int("qp_iter") for s in (a, a2, b)]
    for lbl, j in (("same-config", 1), ("other-team-count", 2)):
        d = np.abs(xs[0] - xs[j]).reshape(B, -1).max(axis=1)
        bad = np.where((d > 0) | (qi[0] != qi[j]) | (qs[0] != qs[j]))[0]
        print("tick", tick, lbl, "differing instances", bad.size, "qp_status a", qs[0][bad][:8], "other", qs[j][bad][:8], "iters", qi[0][bad][:8], qi[j][bad][:8], "max |dx|", d[bad][:6])
    print("tick", tick, "status counts", {int(k): int((qs[0] == k).sum()) for k in np.unique(qs[0])})
    pis = [s.get_all("pi") for s in (a, a2, b)]
    print("tick", tick, "pi bitwise equal (NaN == NaN):", np.array_equal(pis[0], pis[2], equal_nan=True), "NaNs in pi:", int(np.isnan(pis[0]).sum()),
          "in x:", int(np.isnan(xs[0]).sum()))
    for s in (a, a2, b): s.advance(1e-3, seed=7 + tick)
