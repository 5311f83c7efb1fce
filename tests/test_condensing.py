"""Partial condensing (BASELINE.json configs[4]: N = 80 -> N2 = 10, 20 moving obstacles).

The product solves the QP on its uncondensed stages: partial condensing is a reformulation of the same QP (HPIPM
d_part_cond_qp eliminates the intermediate states of a block), chosen by the reference only by solver NAME
(qp_solver = PARTIAL_CONDENSING_HPIPM with qp_solver_cond_N left at N: scripts/usv_pf_ca/acados_settings.py:172).
These tests hold the product against an oracle that REALLY condenses: oracle/condense.py (numpy) builds the condensed QP
- dense block Hessians, every intermediate-stage inequality as a dense general row in (u_hat, x_k0) - solves it with the
oracle's own Mehrotra IPM on the N2 dense stages and expands the solution.

CPU: the condensing oracle against the C oracle (uncondensed) - same solution, and from a point on the linearised
dynamics (b = 0) the same IPM iterates.  GPU (`-m gpu`): the HIP path at BASELINE configs[4]'s shape, B = 256, against the
condensing oracle:
  * tick 0 (the rolled-out guess satisfies the dynamics, so both cold starts coincide): status of every instance,
    median error <= 1e-9, 90 % of the instances <= 1e-6, all <= 1e-3 (per-component norm of tests/util.py; measured
    median 4e-10, 90th percentile 2e-7, worst 3e-5 - 80 stages of a model whose controls are weakly determined);
  * tick 1 (b != 0: the two IPMs start from different slacks and stop at different points of the same tolerance ball,
    which for this R = 0 model is ~3e-2 wide in the controls - tests/test_gpu_closed_loop.py): same status, iterate
    within 5e-2; and, independent of that ball, the device's step is a feasible point of the CONDENSED QP (violation of
    its dense rows <= 1e-6, block dynamics <= 1e-8) with the condensing oracle's optimal objective to 1e-6 relative.
"""
import numpy as np
import pytest

from mpc_collisionavoidance_amd import BatchOcpSolver, scenario, usv_models
from oracle import condense
from tests import util

NAME = "usv_model_pf_ca"


def _spec(oracle, N, K, **kw):
    return oracle.spec(2, N, N * scenario.BENCH_DT, K, sim_steps=scenario.BENCH_SIM_STEPS[NAME], **kw)


def test_condensed_qp_reproduces_the_uncondensed_solution(oracle):
    N, K, B, N2 = 16, 4, 6, 4
    wl = scenario.make_bench_batch(NAME, N, K, B, moving=True, seed=5)
    spec = _spec(oracle, N, K)
    x, u = wl["x_init"].copy(), wl["u_init"].copy()
    for tick in range(2):
        for b in range(B):
            args = (wl["x0"][b], wl["yref"][b], wl["yref_e"][b], wl["p"][b], wl["lh"][b])
            r = oracle.rti(spec, x[b], u[b], *args)
            c = condense.rti_condensed(oracle, spec, x[b], u[b], *args, N2)
            assert r["status"] == c["status"] == 0 and r["qp_status"] == c["qp_status"] == 0
            assert abs(r["qp_iter"] - c["qp_iter"]) <= (0 if tick == 0 else 1)
            # tick 0: b = 0, identical IPM iterates; tick 1: b != 0, different cold starts, the same solution up to the
            # IPM tolerances
            tol = 1e-11 if tick == 0 else 1e-5
            assert util.rel_err(c["x"], r["x"]) < tol and util.rel_err(c["u"], r["u"]) < tol
            x[b], u[b] = r["x"], r["u"]


def test_condensed_qp_has_dense_rows_and_block_dimensions(oracle):
    N, K, N2 = 16, 4, 4
    wl = scenario.make_bench_batch(NAME, N, K, 1, moving=True, seed=5)
    qp, _ = oracle.linearize_and_solve(_spec(oracle, N, K), wl["x_init"][0], wl["u_init"][0], wl["x0"][0], wl["yref"][0],
                                       wl["yref_e"][0], wl["p"][0], wl["lh"][0], solve=False)
    cq = condense.part_cond(qp, N2)
    M = N // N2
    assert len(cq["stages"]) == N2 + 1
    s0, s1 = cq["stages"][0], cq["stages"][1]
    assert s0["H"].shape == (M * 2 + 14, M * 2 + 14) and s0["B"].shape == (14, M * 2) and s0["A"].shape == (14, 14)
    # block 0: M input-bound pairs + (M-1) x (5 state bounds + K obstacle rows); later blocks: M x (2 + 5 + K)
    assert s0["C"].shape[0] == M * 2 + (M - 1) * (5 + K) and s1["C"].shape[0] == M * (2 + 5 + K)
    # an obstacle row of an intermediate stage has become dense in the block's inputs
    last = s1["C"][-1]
    assert np.count_nonzero(last[:M * 2]) >= 2 * (M - 1) - 2 and np.count_nonzero(last[M * 2:]) >= 5


@pytest.mark.gpu
def test_config4_shape_hip_vs_condensing_oracle(oracle):
    N, K, B, N2 = 80, 20, 256, 10
    wl = scenario.make_bench_batch(NAME, N, K, B, moving=True, seed=1234)
    assert not np.array_equal(wl["p"][:, 0], wl["p"][:, N])
    dt, steps = scenario.BENCH_DT, scenario.BENCH_SIM_STEPS[NAME]

    def solver():
        ocp = usv_models.make_ocp(NAME, N * dt, N, K)
        ocp.solver_options.sim_method_num_steps = steps
        ocp.solver_options.qp_solver_cond_N = N2          # accepted: the same QP
        s = BatchOcpSolver(ocp, B)
        scenario.load_into(s, wl)
        return s

    def condensed(spec, x, u, **kw):
        out = [condense.rti_condensed(oracle, spec, x[b], u[b], wl["x0"][b], wl["yref"][b], wl["yref_e"][b], wl["p"][b],
                                      wl["lh"][b], N2, **kw) for b in range(B)]
        return (np.stack([o["x"] for o in out]), np.stack([o["u"] for o in out]), np.array([o["status"] for o in out]),
                np.array([o["qp_status"] for o in out]))

    slack = max(1, int(0.02 * B))
    s = solver()
    spec = _spec(oracle, N, K)
    # ---- tick 0
    x0_, u0_ = wl["x_init"], wl["u_init"]
    st = s.solve()
    xg, ug, qs = s.get_all("x"), s.get_all("u"), s.get_int("qp_status")
    xc, uc, stc, qsc = condensed(spec, x0_, u0_)
    assert (st != stc).sum() <= slack, np.where(st != stc)[0]
    ok = (qs == 0) & (qsc == 0)
    assert ok.mean() > 0.9
    e = np.maximum(util.rel_err_per_instance(xg[ok], xc[ok]), util.rel_err_per_instance(ug[ok], uc[ok]))
    assert np.median(e) <= 1e-9 and np.percentile(e, 90) <= 1e-6 and e.max() <= 1e-3, (np.median(e), np.percentile(e, 90), e.max())
    # ---- tick 1, default tolerances: same statuses, iterates inside the tolerance ball
    st1 = s.solve()
    xg1, ug1, qs1 = s.get_all("x"), s.get_all("u"), s.get_int("qp_status")
    xc1, uc1, stc1, qsc1 = condensed(spec, xg, ug)
    assert (st1 != stc1).sum() <= slack
    ok1 = (qs1 == 0) & (qsc1 == 0)
    assert ok1.mean() > 0.9
    e1 = np.maximum(util.rel_err_per_instance(xg1[ok1], xc1[ok1]), util.rel_err_per_instance(ug1[ok1], uc1[ok1]))
    assert e1.max() <= 5e-2, e1.max()
    s.close()
    # ---- tick 1 once more, judged inside the condensed QP itself: the device's step must be a feasible point of the
    # CONDENSED problem (dense rows, block dynamics) whose objective equals the condensing oracle's optimum - a statement
    # that does not depend on where in the tolerance ball either IPM stopped
    worst_obj = worst_viol = worst_eq = 0.0
    for b in np.where(ok1)[0][:64]:
        c = condense.rti_condensed(oracle, spec, xg[b], ug[b], wl["x0"][b], wl["yref"][b], wl["yref_e"][b], wl["p"][b],
                                   wl["lh"][b], N2)
        dz_hip = np.zeros((N + 1, 16))
        dz_hip[:, 2:] = xg1[b] - xg[b]
        dz_hip[:N, :2] = ug1[b] - ug[b]
        jh, vh, eh = condense.evaluate(c["cq"], dz_hip)
        jc, vc, ec = condense.evaluate(c["cq"], c["dz"])
        worst_obj = max(worst_obj, abs(jh - jc) / max(1.0, abs(jc)))
        worst_viol, worst_eq = max(worst_viol, vh), max(worst_eq, eh)
    assert worst_obj <= 1e-6 and worst_viol <= 1e-6 and worst_eq <= 1e-8, (worst_obj, worst_viol, worst_eq)
