#!/usr/bin/env python3
"""Diagnostic for the co-resident follow-up kernel: a run without hand-over against runs with it under several settings;
prints per tick how many instances differ and whether those are the handed-over ones (iteration count past the threshold)."""
import sys
import numpy as np
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from mpc_collisionavoidance_amd import BatchOcpSolver, scenario, usv_models

name, N, K, B, hand = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
ticks = int(sys.argv[6]) if len(sys.argv) > 6 else 4


def make(opts):
    wl = scenario.make_bench_batch(name, N, K, B, seed=1234)
    ocp = usv_models.make_ocp(name, N * scenario.BENCH_DT, N, K if name != "usv_model" else None)
    ocp.solver_options.sim_method_num_steps = scenario.BENCH_SIM_STEPS[name]
    s = BatchOcpSolver(ocp, B)
    scenario.load_into(s, wl)
    if K > 0:
        s.set_option("static_obstacles", 1)
    s.set_option("disturbance_mask", scenario.NOISE_MASK[name])
    for k, v in (("wide", 0), ("lds_workspace", 0)) + tuple(opts):
        s.set_option(k, v)
    return s


variants = [("sequential, alone, tickets", (("handover_iter", hand), ("handover_co", 1), ("handover_co_mode", 12))),
            ("sequential, alone, by position", (("handover_iter", hand), ("handover_co", 1), ("handover_co_mode", 76))),
            ("seq, alone, by position, 16 wgs", (("handover_iter", hand), ("handover_co", 1), ("handover_co_mode", 76), ("handover_co_wgs", 16))),
            ("beside, tickets", (("handover_iter", hand), ("handover_co", 1), ("handover_co_mode", 0)))]
for tag, opts in variants:
    a, b = make((("handover_iter", 0),)), make(opts)
    for t in range(ticks):
        sa, sb = a.solve(), b.solve()
        qa, qb = a.get_int("qp_iter"), b.get_int("qp_iter")
        xa, xb = a.get_all("x"), b.get_all("x")
        diff = (np.abs(xa - xb).reshape(B, -1).max(axis=1) > 0) | (sa != sb) | (qa != qb)
        fin, tmo = b.handover_co_counts(1)
        print("%-34s tick %d: differ %d (of them with iter >= %d on the reference side: %d; status differs %d)  handed %d  beside %d  timeouts %d"
              % (tag, t, diff.sum(), hand, (diff & (qa >= hand)).sum(), (sa != sb).sum(), b.handover_counts(1)[0], fin[0], tmo[0]), flush=True)
        a.advance(1e-3, seed=5 + t); b.advance(1e-3, seed=5 + t)
        # (continue both from the reference side's state so that every tick is compared from identical inputs)
        b.set_all("x", a.get_all("x")); b.set_all("u", a.get_all("u")); b.set("x0", 0, a.get("x0", 0))
    a.close(); b.close()
