#!/usr/bin/env python3
"""Generates tests/golden/*.npz: small seeded inputs of the hot path with the expected outputs.

The reference's own solver (acados + HPIPM + CasADi-generated code) is not in /root/reference and
cannot be built or imported here, and the reference holds no golden vectors for this path, so the
expected outputs are produced by the CPU oracle (oracle/usv_oracle.c) - whose QP solutions are
verified against the dense KKT conditions by tests/test_oracle_qp.py - and are therefore a
regression pin of the restated algorithm, not of acados itself ("parity unpinned").
Inputs follow the reference scenarios / the synthetic generator (mpc_collisionavoidance_amd.scenario).

Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import binding as ob  # noqa: E402
from mpc_collisionavoidance_amd import scenario  # noqa: E402
from tests import util  # noqa: E402

CASES = [  # (tag, model name, N, K, B, seed, RTI iterations, generator)
    # the reference's own step sizes (usv_model_pf_ca: dt = 0.01 s), obstacles beside the roll-out: rows mostly inactive
    ("m0_n20", "usv_model", 20, 0, 4, 11, 3, "beside"),
    ("m1_n20_k3", "usv_model_guidance_ca1", 20, 3, 6, 12, 3, "beside"),
    ("m2_n20_k3", "usv_model_pf_ca", 20, 3, 6, 13, 3, "beside"),
    ("m1_n40_k10", "usv_model_guidance_ca1", 40, 10, 4, 14, 2, "beside"),
    ("m2_n40_k10", "usv_model_pf_ca", 40, 10, 4, 15, 2, "beside"),
    ("m1_n12_k20", "usv_model_guidance_ca1", 12, 20, 3, 16, 2, "beside"),
    # the benchmark workloads of SURVEY.md 8(d) (dt = 0.05 s, 5 RK4 steps per interval for usv_model_pf_ca, obstacles in the
    # look-ahead: ACTIVE obstacle rows) - BASELINE configs[1] and configs[2] shapes and the two-chunk shape of configs[4]
    # These run CLOSED LOOP (x0 <- x_1 between the ticks, no disturbance, as scripts/usv_guidance_ca1/main.py:169-175): the
    # vehicle has to arrive at the obstacles before their rows bind.  x_out / u_out of tick t-1 and x0_in[t] are tick t's inputs.
    ("m2s_n20_k3", "usv_model_pf_ca", 20, 3, 8, 21, 12, "survey"),
    ("m2s_n40_k10", "usv_model_pf_ca", 40, 10, 8, 22, 5, "survey"),
    ("m1s_n20_k3", "usv_model_guidance_ca1", 20, 3, 8, 23, 12, "survey"),
    ("m1s_n40_k10", "usv_model_guidance_ca1", 40, 10, 8, 24, 5, "survey"),
    ("m2s_n40_k20", "usv_model_pf_ca", 40, 20, 4, 25, 4, "survey"),
]


PROFILE = "BALANCE"   # the QP solver profile the fixtures are made under (oracle/usv_oracle.c usv_opts_profile): the default since round 6


def active_rows(name, wl, x, K):
    """[B] bool: some obstacle row of the iterate x sits on its bound (hard rows: h - lh < 1e-3; soft rows, whose slack rests at
    lsh = -0.2: h - lh < 0.2 + 1e-3)."""
    if not K:
        return np.zeros(x.shape[0], dtype=bool)
    ipx, ipy = (5, 6) if name == "usv_model_guidance_ca1" else (10, 11)
    N = wl["lh"].shape[1]
    px, py = x[:, 1:N, ipx, None], x[:, 1:N, ipy, None]
    h = np.hypot(px - wl["p"][:, 1:N, 0::2], py - wl["p"][:, 1:N, 1::2])
    margin = 0.2 if name == "usv_model_guidance_ca1" else 0.0
    return ((h - wl["lh"][:, 1:N]) < margin + 1e-3).any(axis=(1, 2))


def main():
    out_dir = os.path.dirname(os.path.abspath(__file__))
    for tag, name, N, K, B, seed, iters, gen in CASES:
        if gen == "survey":
            wl = scenario.make_bench_batch(name, N, K, B, seed=seed)
            dt, steps = scenario.BENCH_DT, scenario.BENCH_SIM_STEPS[name]
        else:
            ocp, wl = util.make(name, N, K, B, seed=seed)
            dt, steps = scenario.DT[name], 1
        path = os.path.join(out_dir, tag + ".npz")
        if gen != "survey" and os.path.exists(path):
            continue   # the round-1 fixtures stay byte-for-byte what they were: made under the QP solver profile "R04" (no `profile` key)
        spec = util.oracle_spec(ob, name, N, dt, K, sim_steps=steps, hpipm_mode=PROFILE)
        x, u = wl["x_init"].copy(), wl["u_init"].copy()
        xs, us, sts, its, x0s = [], [], [], [], []
        x0 = wl["x0"].copy()
        act = np.zeros(B, dtype=bool)
        for _ in range(iters):
            x0s.append(x0.copy())
            x, u, st, it = util.oracle_rti(ob, spec, wl, x, u, x0=x0)
            xs.append(x.copy()); us.append(u.copy()); sts.append(st.copy()); its.append(it.copy())
            act |= active_rows(name, wl, x, K)
            if gen == "survey":
                x0 = x[:, 1].copy()
        np.savez_compressed(path, name=name, N=N, K=K, B=B, dt=dt, seed=seed, sim_steps=steps, generator=gen, active=act, profile=PROFILE,
                            x0=wl["x0"], yref=wl["yref"], yref_e=wl["yref_e"], p=wl["p"], lh=wl["lh"],
                            x_init=wl["x_init"], u_init=wl["u_init"], x0_in=np.stack(x0s),
                            x_out=np.stack(xs), u_out=np.stack(us), status=np.stack(sts), qp_iter=np.stack(its))
        print(tag, "status max", int(np.max(sts)), "qp_iter", [int(v.max()) for v in its], "active_row_frac", float(act.mean()))


if __name__ == "__main__":
    main()
