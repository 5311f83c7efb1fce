"""Edge-case sweep of the device path against the oracle (development aid; run on the GPU box)."""
import sys, itertools
import numpy as np
sys.path.insert(0, ".")
import torch  # noqa
from mpc_collisionavoidance_amd import BatchOcpSolver, scenario, usv_models
from oracle import binding as ob
from tests import util

bad = 0
cases = []
for name in ("usv_model", "usv_model_guidance_ca1", "usv_model_pf_ca"):
    Ks = [0] if name == "usv_model" else [1, 2, 6, 9, 10, 12, 15, 16, 17, 25, 31, 32]
    for K in Ks:
        for N, B in ((2, 5), (3, 1), (17, 7), (60, 4)):   # (N = 1 is refused by usvmpc_create: include/usvmpc.h)
            cases.append((name, N, K, B, False))
    if name != "usv_model":
        cases.append((name, 12, 20, 6, True))
        cases.append((name, 12, 7, 6, True))
for name, N, K, B, moving in cases:
    try:
        ocp, wl = util.make(name, N, K, B, seed=N * 100 + K, moving=moving)
        s = BatchOcpSolver(ocp, B)
        scenario.load_into(s, wl)
        spec = util.oracle_spec(ob, name, N, scenario.DT[name], K)
        xo, uo = wl["x_init"].copy(), wl["u_init"].copy()
        worst = 0.0
        for it in range(3):
            st = s.solve()
            xo, uo, sto, ito = util.oracle_rti(ob, spec, wl, xo, uo)
            qs = s.get_int("qp_status")
            ok = (st == 0) & (sto == 0) & (qs == 0) & (ito < 50)
            if not np.array_equal(st == 0, sto == 0):
                print("STATUS MISMATCH", name, N, K, B, moving, it, st, sto); bad += 1
            x, u = s.get_all("x"), s.get_all("u")
            if ok.any():
                worst = max(worst, util.rel_err(x[ok], xo[ok]), util.rel_err(u[ok], uo[ok]))
            # keep both sides on the same iterate where one of them failed
            s.set_all("x", xo); s.set_all("u", uo)
        flag = "" if worst < 1e-6 else "  <<<<<<"
        if flag: bad += 1
        print("%-24s N=%-3d K=%-2d B=%d moving=%d  rel err %.2e  ok %d/%d%s" % (name, N, K, B, moving, worst, ok.sum(), B, flag), flush=True)
        s.close()
    except Exception as e:
        bad += 1
        print("EXCEPTION", name, N, K, B, moving, repr(e), flush=True)
print("bad:", bad)
