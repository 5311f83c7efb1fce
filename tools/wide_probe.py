"""Development aid (GPU): the closed loop of one workload on the latency mapping and on the throughput mapping from identical inputs,
tick by tick, both against the oracle - where does an instance differ?   python tools/wide_probe.py [model] [N] [K] [B] [ticks]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpc_collisionavoidance_amd import BatchOcpSolver, scenario, usv_models
from oracle import binding as ob
from tests import util
name = sys.argv[1] if len(sys.argv) > 1 else "usv_model_guidance_ca1"
N, K, B, ticks = (int(sys.argv[i]) if len(sys.argv) > i else d for i, d in ((2, 40), (3, 10), (4, 256), (5, 25)))
wl = scenario.make_bench_batch(name, N, K, B, seed=1234)
dt, steps = scenario.BENCH_DT, scenario.BENCH_SIM_STEPS[name]
ocp = usv_models.make_ocp(name, N * dt, N, K)
ocp.solver_options.sim_method_num_steps = steps
spec = ob.spec(util.MODEL_ID[name], N, N * dt, K, sim_steps=steps)
def make(wide):
    s = BatchOcpSolver(ocp, B)
    scenario.load_into(s, wl)
    s.set_option("static_obstacles", 1)
    s.set_option("disturbance_mask", scenario.NOISE_MASK[name])
    s.set_option("wide", wide)
    return s
a, b = make(1), make(0)
x0 = wl["x0"].copy()
for t in range(ticks):
    xin, uin = a.get_all("x"), a.get_all("u")
    sa, sb = a.solve(), b.solve()
    xo, uo = xin.copy(), uin.copy()
    sto, ito = ob.rti_batch(spec, xo, uo, x0, wl["yref"], wl["yref_e"], wl["p"], wl["lh"], threads=0)
    qa, qb = a.get_int("qp_iter"), b.get_int("qp_iter")
    xa, ua, xb, ub = a.get_all("x"), a.get_all("u"), b.get_all("x"), b.get_all("u")
    ok = (sa == 0) & (sb == 0) & (sto == 0)
    eab = np.maximum(util.rel_err_per_instance(xa, xb), util.rel_err_per_instance(ua, ub))
    eao = np.maximum(util.rel_err_per_instance(xa, xo), util.rel_err_per_instance(ua, uo))
    ebo = np.maximum(util.rel_err_per_instance(xb, xo), util.rel_err_per_instance(ub, uo))
    i = int(np.argmax(np.where(ok, eao, 0)))
    print("tick %2d  wide-vs-throughput max %.2e | wide-vs-oracle max %.2e (instance %d: iters wide %d thr %d oracle %d, thr-vs-oracle there %.2e, wide-vs-thr there %.2e) | thr-vs-oracle max %.2e | iter diffs wide/thr %d"
          % (t, eab[ok].max(), eao[ok].max(), i, qa[i], qb[i], ito[i], ebo[i], eab[i], ebo[ok].max(), (qa != qb).sum()), flush=True)
    a.advance(1e-3, seed=2000 + t)
    a.sync()
    x0 = a.get("x0", 0)
    b.set("x0", 0, x0)
    b.set_all("x", xa)
    b.set_all("u", ua)
