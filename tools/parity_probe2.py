"""Where does the one-step discrepancy on usv_model_pf_ca come from?  (GPU, development aid)
Per tick, from identical inputs: device (default tolerances) vs oracle (default) vs oracle converged to 1e-11 ("truth")."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpc_collisionavoidance_amd import BatchOcpSolver, scenario, usv_models
from oracle import binding as ob
name, B, ticks, N, K = "usv_model_pf_ca", int(sys.argv[1]) if len(sys.argv) > 1 else 512, int(sys.argv[2]) if len(sys.argv) > 2 else 12, 40, 10
wl = scenario.make_bench_batch(name, N, K, B)
dt, steps = scenario.BENCH_DT, scenario.BENCH_SIM_STEPS[name]
ocp = usv_models.make_ocp(name, N * dt, N, K)
ocp.solver_options.sim_method_num_steps = steps
s = BatchOcpSolver(ocp, B)
scenario.load_into(s, wl)
s.set_option("static_obstacles", 1)
s.set_option("disturbance_mask", scenario.NOISE_MASK[name])
spec = ob.spec(2, N, N * dt, K, sim_steps=steps)
tight = ob.spec(2, N, N * dt, K, sim_steps=steps, tol_stat=1e-11, tol_eq=1e-11, tol_ineq=1e-11, tol_comp=1e-12, qp_iter_max=80)
classic = ob.spec(2, N, N * dt, K, sim_steps=steps, riccati=ob.RICCATI_CLASSIC)
data = (wl["yref"], wl["yref_e"], wl["p"], wl["lh"])
x0 = wl["x0"].copy()
def err(a, b, ok):
    a, b = a[ok], b[ok]
    ax = tuple(range(a.ndim - 1))
    sc = np.maximum(1e-2, np.abs(b).max(axis=ax))
    per_inst = (np.abs(a - b) / sc).reshape(a.shape[0], -1).max(axis=1)
    return per_inst
for t in range(ticks):
    xp, up = s.get_all("x"), s.get_all("u")
    s.solve()
    xg, ug, qs = s.get_all("x"), s.get_all("u"), s.get_int("qp_status")
    xo, uo = xp.copy(), up.copy(); sto, ito = ob.rti_batch(spec, xo, uo, x0, *data, threads=8)
    xc, uc = xp.copy(), up.copy(); stc, itc = ob.rti_batch(classic, xc, uc, x0, *data, threads=8)
    xt, ut = xp.copy(), up.copy(); stt, itt = ob.rti_batch(tight, xt, ut, x0, *data, threads=8)
    ok = (qs == 0) & (sto == 0) & (ito < 50) & (stt == 0) & (itt < 80) & (stc == 0)
    e_do, e_dt, e_ot, e_dc = err(ug, uo, ok), err(ug, ut, ok), err(uo, ut, ok), err(ug, uc, ok)
    print("tick %2d ok %.3f | u: dev-oracle max %.1e p99 %.1e | dev-tight %.1e %.1e | oracle-tight %.1e %.1e | dev-classic %.1e | x dev-oracle %.1e dev-tight %.1e oracle-tight %.1e  (tight iters %.1f vs %.1f)" % (
        t, ok.mean(), e_do.max(), np.percentile(e_do, 99), e_dt.max(), np.percentile(e_dt, 99), e_ot.max(), np.percentile(e_ot, 99), e_dc.max(),
        err(xg, xo, ok).max(), err(xg, xt, ok).max(), err(xo, xt, ok).max(), itt[ok].mean(), ito[ok].mean()), flush=True)
    s.advance(1e-3, seed=2000 + t)
    x0 = s.get("x0", 0)
