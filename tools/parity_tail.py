"""Classify every device-vs-oracle difference above north_star's 1e-5 on the bench workload (GPU; writes the tracked note
profiles/r03_parity_tail.txt).

Per closed-loop tick and instance, all from IDENTICAL inputs (the iterate and x0 the stock device build starts the tick from):
  dev     the shipped library (estimate + Newton reciprocals, paired reciprocals)
  exact   the same kernels with tools/experiments/exact_div.patch applied (IEEE division / square root everywhere): build_ab/libusvmpc_exactdiv.so
  oracle  oracle/usv_oracle.c at the default IPM tolerances (square-root Riccati)
  tight   the oracle converged to 1e-11 ("the solution of the QP")
and the independent KKT certificate of tests/kkt.py for both device builds.  For every instance with dev-vs-oracle > 1e-5
the line shows the IPM iteration counts on the three sides, the distance of each to the tight solution, and whether the
exact-division build removes the difference.

usage: python tools/parity_tail.py [B=2048] [ticks=10] [out=profiles/r03_parity_tail.txt]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402,F401  (before the solver library: one HIP runtime)
from mpc_collisionavoidance_amd import BatchOcpSolver, scenario, usv_models  # noqa: E402
from oracle import binding as ob  # noqa: E402
from tests import kkt, util  # noqa: E402
from tests.test_kkt_certify import _pad_pi, _step  # noqa: E402

name, N, K = "usv_model_pf_ca", 40, 10
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
ticks = int(sys.argv[2]) if len(sys.argv) > 2 else 10
out_path = sys.argv[3] if len(sys.argv) > 3 else os.path.join(ROOT, "profiles", "r03_parity_tail.txt")
EXACT = os.path.join(ROOT, "build_ab", "libusvmpc_exactdiv.so")

wl = scenario.make_bench_batch(name, N, K, B, seed=1234)
dt, steps = scenario.BENCH_DT, scenario.BENCH_SIM_STEPS[name]
ocp = usv_models.make_ocp(name, N * dt, N, K)
ocp.solver_options.sim_method_num_steps = steps


def make(libpath):
    if libpath:
        os.environ["USVMPC_LIB"] = libpath
    else:
        os.environ.pop("USVMPC_LIB", None)
    s = BatchOcpSolver(ocp, B)
    scenario.load_into(s, wl)
    s.set_option("static_obstacles", 1)
    s.set_option("disturbance_mask", scenario.NOISE_MASK[name])
    return s


dev = make(None)
exact = make(EXACT) if os.path.exists(EXACT) else None
spec = ob.spec(2, N, N * dt, K, sim_steps=steps)
tight = ob.spec(2, N, N * dt, K, sim_steps=steps, tol_stat=1e-11, tol_eq=1e-11, tol_ineq=1e-11, tol_comp=1e-12, qp_iter_max=100)
data = (wl["yref"], wl["yref_e"], wl["p"], wl["lh"])
x0 = wl["x0"].copy()


def inst_err(xa, ua, xb, ub):
    """tests/util.rel_err_per_instance over (x, u): worst component, every component scaled by its own magnitude over the batch"""
    return np.maximum(util.rel_err_per_instance(xa, xb), util.rel_err_per_instance(ua, ub))


lines = []
tot = dict(n=0, conv=0, above=0, above_exact=0, cert=0, cert_exact=0, removed=0, same_iter_above=0)
for tk in range(ticks):
    xp, up = dev.get_all("x"), dev.get_all("u")
    dev.solve()
    xg, ug, qs, qi = dev.get_all("x"), dev.get_all("u"), dev.get_int("qp_status"), dev.get_int("qp_iter")
    qp = kkt.linearize_batch(ob, spec, xp, up, x0, *data)
    rd = kkt.kkt_batch(qp, _step(xg, ug, xp, up), _pad_pi(dev.get_all("pi")), dev.get_all("lam"), dev.get_all("t"))
    cd = kkt.certified(rd, 1.02e-6, 1.02e-8, 1.02e-8, 1.02e-8)
    if exact is not None:
        exact.set_all("x", xp); exact.set_all("u", up); exact.set("x0", 0, x0)
        exact.set_option("sort_by_difficulty", 0)
        exact.solve()
        xe, ue, qse, qie = exact.get_all("x"), exact.get_all("u"), exact.get_int("qp_status"), exact.get_int("qp_iter")
        re_ = kkt.kkt_batch(qp, _step(xe, ue, xp, up), _pad_pi(exact.get_all("pi")), exact.get_all("lam"), exact.get_all("t"))
        ce = kkt.certified(re_, 1.02e-6, 1.02e-8, 1.02e-8, 1.02e-8)
    xo, uo = xp.copy(), up.copy()
    sto, ito = ob.rti_batch(spec, xo, uo, x0, *data, threads=0)
    xt, ut = xp.copy(), up.copy()
    stt, itt = ob.rti_batch(tight, xt, ut, x0, *data, threads=0)
    ok = (qs == 0) & (sto == 0) & (ito < spec.opts.qp_iter_max)
    e_do = inst_err(xg, ug, xo, uo)
    tot["n"] += B
    tot["conv"] += int(ok.sum())
    tot["cert"] += int((ok & cd).sum())
    above = ok & (e_do > 1e-5)
    tot["above"] += int(above.sum())
    tot["same_iter_above"] += int((above & (qi == ito)).sum())
    if exact is not None:
        oke = ok & (qse == 0)
        e_eo = inst_err(xe, ue, xo, uo)
        tot["cert_exact"] += int((oke & ce).sum())
        tot["above_exact"] += int((oke & (e_eo > 1e-5)).sum())
        tot["removed"] += int((above & oke & (e_eo <= 1e-5)).sum())
    tt_ok = (stt == 0) & (itt < 100)
    e_dt, e_ot = inst_err(xg, ug, xt, ut), inst_err(xo, uo, xt, ut)
    lines.append("tick %2d: converged on both sides %5d / %d   dev-vs-oracle p50 %.1e p99 %.1e max %.1e   > 1e-5: %d   KKT-certified (dev) %d / %d"
                 "   distance to the tight solution (p50 / max): dev %.1e / %.1e, oracle %.1e / %.1e"
                 % (tk, ok.sum(), B, np.percentile(e_do[ok], 50), np.percentile(e_do[ok], 99), e_do[ok].max(), above.sum(),
                    (ok & cd).sum(), ok.sum(), np.percentile(e_dt[ok & tt_ok], 50), e_dt[ok & tt_ok].max(),
                    np.percentile(e_ot[ok & tt_ok], 50), e_ot[ok & tt_ok].max()))
    for b in np.where(above)[0]:
        lines.append("    instance %5d  dev-vs-oracle %.2e  qp_iter dev %2d / oracle %2d%s  to tight: dev %.2e, oracle %.2e (tight: %d iterations%s)"
                     "  KKT dev stat %.1e eq %.1e ineq %.1e comp %.1e %s%s"
                     % (b, e_do[b], qi[b], ito[b], (" / exact-div %2d" % qie[b]) if exact is not None else "", e_dt[b], e_ot[b], itt[b],
                        "" if tt_ok[b] else ", NOT converged", rd["stat"][b], rd["eq"][b], rd["ineq"][b], rd["comp"][b],
                        "certified" if cd[b] else "NOT CERTIFIED",
                        ("  | exact-div build vs oracle %.2e (%s)" % (e_eo[b], "removed" if e_eo[b] <= 1e-5 else "stays")) if exact is not None else ""))
    print(lines[-1 - int(above.sum())], flush=True)
    dev.advance(1e-3, seed=2000 + tk)
    dev.sync()
    x0 = dev.get("x0", 0)

head = ["parity tail on BASELINE configs[2] (usv_model_pf_ca, N=40, Tf=2 s, 10 obstacles, SURVEY 8(d) generator, seed 1234), "
        "%d instances x %d closed-loop ticks, every solve from identical inputs on all sides" % (B, ticks),
        "error norm: tests/util.rel_err_per_instance over x and u (worst component, each scaled by its own magnitude over the batch, floor 1e-2)",
        "",
        "solves %d, converged on both sides %d" % (tot["n"], tot["conv"]),
        "KKT-certified by tests/kkt.py (stat <= 1e-6, eq / ineq / comp <= 1e-8, lam, t >= 0): shipped build %d / %d = %.6f%s"
        % (tot["cert"], tot["conv"], tot["cert"] / max(1, tot["conv"]),
           (", exact-division build %d" % tot["cert_exact"]) if exact is not None else ""),
        "dev-vs-oracle above 1e-5: %d of %d (%.2e); of those with the SAME iteration count on both sides: %d"
        % (tot["above"], tot["conv"], tot["above"] / max(1, tot["conv"]), tot["same_iter_above"]),
        ("exact-division build: above 1e-5 vs oracle %d; of the shipped build's outliers it removes %d of %d"
         % (tot["above_exact"], tot["removed"], tot["above"])) if exact is not None else "exact-division build not present",
        ""]
open(out_path, "w").write("\n".join(head + lines) + "\n")
print("\n".join(head))
