"""Partial condensing (BASELINE.json configs[4]: N = 80 -> N2 = 10, 20 moving obstacles).

The reference selects partial condensing only by solver NAME (qp_solver = PARTIAL_CONDENSING_HPIPM with qp_solver_cond_N
left at N - blocks of one stage: scripts/usv_pf_ca/acados_settings.py:172); for that setting the Riccati sweep over the N
stages (csrc/qp_ipm.hpp) is the solver.  With qp_solver_cond_N = N2 < N the product condenses on the device
(csrc/cond_ipm.hpp, option "qp_cond_N"): HPIPM d_part_cond_qp restated - block Hessians, dense block dynamics, every
intermediate-stage inequality a general row in (u_hat, x_k0) -, the IPM on the N2 dense stages, expansion of (x, u, pi).
The oracle for it is oracle/condense.py (numpy: part_cond + the oracle's Mehrotra IPM on the dense stages + expand).

CPU: the condensing oracle against the C oracle (uncondensed) - same solution, and from a point on the linearised
dynamics (b = 0) the same IPM iterates; the condensing KERNEL BODY (cond_ipm.hpp compiled serially into the emulator
library, one CPU thread playing the team of an instance) against the condensing oracle - same statuses, same iteration
counts, iterates to 1e-9 - and its expanded (x, u, pi, lam, t) against the KKT conditions of the UNCONDENSED QP.
GPU (`-m gpu`): at BASELINE configs[4]'s shape, B = 256, (a) the condensing kernel against the condensing oracle
(identical iteration path: iterates to 1e-8), against the uncondensed kernel (same solution inside the IPM tolerance ball)
and KKT-certified on the uncondensed QP; (b) the uncondensed kernel against the condensing oracle as before:
  * tick 0 (the rolled-out guess satisfies the dynamics, so both cold starts coincide): status of every instance,
    median error <= 1e-9, 90 % of the instances <= 1e-6, all <= 1e-3 (per-component norm of tests/util.py; measured
    median 4e-10, 90th percentile 2e-7, worst 3e-5 - 80 stages of a model whose controls are weakly determined);
  * tick 1 (b != 0: the two IPMs start from different slacks and stop at different points of the same tolerance ball,
    which for this R = 0 model is ~3e-2 wide in the controls - tests/test_gpu_closed_loop.py): same status, iterate
    within 5e-2; and, independent of that ball, the device's step is a feasible point of the CONDENSED QP (violation of
    its dense rows <= 1e-6, block dynamics <= 1e-8) with the condensing oracle's optimal objective to 1e-6 relative.
"""
import ctypes as C

import numpy as np
import pytest

from mpc_collisionavoidance_amd import BatchOcpSolver, _capi, scenario, usv_models
from oracle import condense
from tests import kkt, util

NAME = "usv_model_pf_ca"


def _spec(oracle, N, K, **kw):
    return oracle.spec(2, N, N * scenario.BENCH_DT, K, sim_steps=scenario.BENCH_SIM_STEPS[NAME], **kw)


@pytest.mark.parametrize("N2", [4, 3, 7])   # (3: blocks of 6, 5, 5 stages; 7: of 3, 3, 2, 2, 2, 2, 2 - HPIPM's partition when N2 does not divide N)
def test_condensed_qp_reproduces_the_uncondensed_solution(oracle, N2):
    N, K, B = 16, 4, 6
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


# ------------------------------------------------------------------------------ the condensing kernel body on the CPU
def _emu_cond_rti(emu, desc, wl, x, u, N2, export=False):
    from tests.test_emu_kernels import emu_rti
    emu.usv_emu_set_cond.argtypes = [C.c_int]
    emu.usv_emu_set_export.argtypes = [_capi._dp, _capi._dp]
    emu.usv_emu_set_export.restype = None
    B, N = x.shape[0], desc.N
    nlam = 2 * (desc.nbu + desc.nbx + desc.K)
    lam, t = np.zeros((B, N + 1, nlam)), np.zeros((B, N + 1, nlam))
    emu.usv_emu_set_cond(N2)
    if export:
        emu.usv_emu_set_export(lam.ctypes.data_as(_capi._dp), t.ctypes.data_as(_capi._dp))
    try:
        r = emu_rti(emu, desc, wl, x, u)
    finally:
        emu.usv_emu_set_cond(0)
        emu.usv_emu_set_export(None, None)
    r["lam"], r["t"] = lam, t
    return r


# (blocks of 4 stages; two obstacle chunks; a model without obstacle rows; blocks of 2)
@pytest.mark.parametrize("name,N,K,B,N2", [("usv_model_pf_ca", 16, 4, 4, 4), ("usv_model_pf_ca", 12, 18, 2, 3),
                                           ("usv_model", 8, 0, 3, 2), ("usv_model_pf_ca", 12, 3, 2, 6),
                                           ("usv_model_pf_ca", 10, 3, 2, 1),    # (N2 = 1: the whole horizon in one block - full condensing)
                                           ("usv_model_pf_ca", 16, 4, 3, 3), ("usv_model_pf_ca", 16, 4, 2, 7),   # (blocks of 6 5 5 / 3 3 2 2 2 2 2 stages)
                                           ("usv_model", 11, 0, 2, 4)])
def test_condensing_kernel_body_matches_condensing_oracle(oracle, emu, name, N, K, B, N2):
    dt, steps = scenario.BENCH_DT, scenario.BENCH_SIM_STEPS[name]
    wl = scenario.make_bench_batch(name, N, K, B, moving=K > 0, seed=5)
    ocp = usv_models.make_ocp(name, N * dt, N, None if name == "usv_model" else K)
    ocp.solver_options.sim_method_num_steps = steps
    desc = _capi.desc_from_ocp(ocp, batch=B)
    spec = oracle.spec(util.MODEL_ID[name], N, N * dt, K, sim_steps=steps)
    x, u = wl["x_init"].copy(), wl["u_init"].copy()
    for tick in range(3):
        r = _emu_cond_rti(emu, desc, wl, x, u, N2)
        for b in range(B):
            c = condense.rti_condensed(oracle, spec, x[b], u[b], wl["x0"][b], wl["yref"][b], wl["yref_e"][b], wl["p"][b], wl["lh"][b], N2)
            assert r["status"][b] == c["status"] and r["qp_status"][b] == c["qp_status"] and r["qp_iter"][b] == c["qp_iter"]
            e = max(util.rel_err(r["x"][b], c["x"]), util.rel_err(r["u"][b], c["u"]))
            assert e < 1e-9, (tick, b, e)
            assert np.allclose(r["res"][b], c["res"], rtol=1e-3, atol=1e-12)
        x, u = r["x"], r["u"]
        wl["x0"] = x[:, 1].copy()   # closed loop: the dynamics residual is non-zero from the second tick on


@pytest.mark.parametrize("K,N2", [(5, 4), (20, 5), (5, 3)])   # (one / two obstacle chunks; blocks of 6 5 5)
def test_condensing_kernel_body_with_soft_rows(oracle, emu, K, N2):
    """usv_model_guidance_ca1 (soft obstacle rows: the slacks are eliminated row by row).  oracle/condense.py has hard rows only, so the
    checks are: tick 0, where the rolled-out guess satisfies the dynamics and the condensed and the uncondensed IPM coincide iterate by
    iterate - the uncondensed C oracle: same iteration counts, iterates and slacks to 1e-9; every tick - the expanded
    (x, u, pi, lam, t, sl, su) against the KKT conditions of the original QP (tests/kkt.py, independent of any iteration path)."""
    name, N, B = "usv_model_guidance_ca1", 16, 3
    dt, steps = scenario.BENCH_DT, scenario.BENCH_SIM_STEPS[name]
    wl = scenario.make_bench_batch(name, N, K, B, seed=11)
    ocp = usv_models.make_ocp(name, N * dt, N, K)
    ocp.solver_options.sim_method_num_steps = steps
    desc = _capi.desc_from_ocp(ocp, batch=B)
    spec = oracle.spec(util.MODEL_ID[name], N, N * dt, K, sim_steps=steps)
    from tests.test_emu_kernels import emu_rti
    emu.usv_emu_set_cond.argtypes = [C.c_int]
    emu.usv_emu_set_export.argtypes = [_capi._dp, _capi._dp]
    emu.usv_emu_set_export.restype = None
    nlam = 2 * (desc.nbu + desc.nbx + 2 * K)
    x, u = wl["x_init"].copy(), wl["u_init"].copy()
    active = 0
    for tick in range(3):
        lam, t = np.zeros((B, N + 1, nlam)), np.zeros((B, N + 1, nlam))
        emu.usv_emu_set_cond(N2)
        emu.usv_emu_set_export(lam.ctypes.data_as(_capi._dp), t.ctypes.data_as(_capi._dp))
        try:
            r = emu_rti(emu, desc, wl, x, u)
        finally:
            emu.usv_emu_set_cond(0)
            emu.usv_emu_set_export(None, None)
        assert (r["status"] == 0).all() and (r["qp_status"] == 0).all()
        if tick == 0:
            xo, uo, sto, ito = util.oracle_rti(oracle, spec, wl, x.copy(), u.copy())
            assert np.array_equal(r["qp_iter"], ito) and np.array_equal(r["status"], sto)
            assert util.rel_err(r["x"], xo) < 1e-9 and util.rel_err(r["u"], uo) < 1e-9
        qp = kkt.linearize_batch(oracle, spec, x, u, wl["x0"], wl["yref"], wl["yref_e"], wl["p"], wl["lh"])
        dz = np.zeros((B, N + 1, 9))
        dz[:, :, 1:] = r["x"] - x
        dz[:, :N, :1] = r["u"] - u
        pad = lambda a: np.concatenate([a, np.zeros_like(a[:, :1])], axis=1)   # noqa: E731
        res = kkt.kkt_batch(qp, dz, np.concatenate([np.zeros((B, 1, 8)), r["pi"]], axis=1), lam, t, pad(r["sl"]), pad(r["su"]))
        assert kkt.certified(res).all(), (tick, res)
        nrow = desc.nbu + desc.nbx + K
        active += int((lam[:, :, desc.nbu + desc.nbx:nrow] > 1e-3).any(axis=(1, 2)).sum())
        x, u = r["x"], r["u"]
        wl["x0"] = x[:, 1].copy()
    assert active > 0, "no obstacle row carried a multiplier: the case does not exercise the inequality path"


def test_condensing_kernel_body_solution_is_certified_on_the_uncondensed_qp(oracle, emu):
    """Expansion (x, u by the original dynamics, pi by the adjoint recursion inside a block) and the multiplier read-back
    of the condensed solve: together they must satisfy the KKT conditions of the ORIGINAL N-stage QP."""
    name, N, K, B, N2 = "usv_model_pf_ca", 40, 10, 3, 5
    dt, steps = scenario.BENCH_DT, scenario.BENCH_SIM_STEPS[name]
    wl = scenario.make_bench_batch(name, N, K, B, seed=11)
    ocp = usv_models.make_ocp(name, N * dt, N, K)
    ocp.solver_options.sim_method_num_steps = steps
    desc = _capi.desc_from_ocp(ocp, batch=B)
    spec = oracle.spec(util.MODEL_ID[name], N, N * dt, K, sim_steps=steps)
    x, u = wl["x_init"].copy(), wl["u_init"].copy()
    active = 0
    for it in range(2):
        r = _emu_cond_rti(emu, desc, wl, x, u, N2, export=True)
        qp = kkt.linearize_batch(oracle, spec, x, u, wl["x0"], wl["yref"], wl["yref_e"], wl["p"], wl["lh"])
        ok = r["qp_status"] == 0
        assert ok.all()
        dz = np.zeros((B, N + 1, 16))
        dz[:, :, 2:] = r["x"] - x
        dz[:, :N, :2] = r["u"] - u
        pi = np.concatenate([np.zeros((B, 1, 14)), r["pi"]], axis=1)
        res = kkt.kkt_batch(qp, dz, pi, r["lam"], r["t"])
        assert kkt.certified(res).all(), res
        nrow = desc.nbu + desc.nbx + K
        active += int((r["lam"][:, :, desc.nbu + desc.nbx:nrow] > 1e-3).any(axis=(1, 2)).sum())
        x, u = r["x"], r["u"]
    assert active > 0


# ------------------------------------------------------------------------------ the condensing kernel on the device
def _cond_solver(name, N, K, B, wl, N2, extra=()):
    dt, steps = scenario.BENCH_DT, scenario.BENCH_SIM_STEPS[name]
    ocp = usv_models.make_ocp(name, N * dt, N, None if name == "usv_model" else K)
    ocp.solver_options.sim_method_num_steps = steps
    if N2:
        ocp.solver_options.qp_solver_cond_N = N2
    s = BatchOcpSolver(ocp, B)
    scenario.load_into(s, wl)
    for k, v in extra:
        s.set_option(k, v)
    return s


@pytest.mark.gpu
@pytest.mark.parametrize("name,N,K,B,N2", [("usv_model_pf_ca", 80, 20, 96, 10), ("usv_model_pf_ca", 40, 10, 160, 8),
                                           ("usv_model", 20, 0, 70, 4), ("usv_model_pf_ca", 10, 3, 40, 1),
                                           ("usv_model_pf_ca", 40, 10, 64, 6)])   # (blocks of 7 7 7 7 6 6 stages)
def test_condensing_kernel_on_the_device_vs_condensing_oracle(oracle, name, N, K, B, N2):
    """qp_solver_cond_N = N2 is APPLIED: the device condenses, solves the N2 dense stages and expands.  Same iteration path as
    oracle/condense.py: statuses, iteration counts (more than one off for at most 2 % of the instances) and
    iterates (median 1e-9, 90 % 1e-6, all 1e-3); the uncondensed kernel reaches the same solution inside the IPM's tolerance ball."""
    wl = scenario.make_bench_batch(name, N, K, B, moving=K > 0, seed=1234)
    dt, steps = scenario.BENCH_DT, scenario.BENCH_SIM_STEPS[name]
    spec = oracle.spec(util.MODEL_ID[name], N, N * dt, K, sim_steps=steps)
    sc, su_ = _cond_solver(name, N, K, B, wl, N2), _cond_solver(name, N, K, B, wl, 0)
    x, u = wl["x_init"].copy(), wl["u_init"].copy()
    x0 = wl["x0"].copy()
    for tick in range(2):
        st = sc.solve()
        stu = su_.solve()
        xg, ug, qs, qi = sc.get_all("x"), sc.get_all("u"), sc.get_int("qp_status"), sc.get_int("qp_iter")
        xu, uu, qsu = su_.get_all("x"), su_.get_all("u"), su_.get_int("qp_status")
        errs, off = [], 0
        for b in range(B):
            c = condense.rti_condensed(oracle, spec, x[b], u[b], x0[b], wl["yref"][b], wl["yref_e"][b], wl["p"][b], wl["lh"][b], N2)
            assert st[b] == c["status"] and qs[b] == c["qp_status"], (tick, b, st[b], c["status"], qs[b], c["qp_status"])
            off += int(abs(int(qi[b]) - int(c["qp_iter"])) > 1)   # (an ill-conditioned instance may leave the common path by a few iterations)
            if c["qp_status"] == 0 and qi[b] == c["qp_iter"]:
                errs.append(max(util.rel_err(xg[b], c["x"]), util.rel_err(ug[b], c["u"])))
        assert off <= max(1, int(0.02 * B)), off
        # (measured at N = 80 / 96 instances: median 2e-10, worst 5e-5 - 80 stages of a model whose controls are weakly determined
        # (R = 0): the same gates as for the uncondensed kernel below)
        worst = max(errs)
        assert np.median(errs) <= 1e-9 and np.percentile(errs, 90) <= 1e-6 and worst <= 1e-3, (np.median(errs), np.percentile(errs, 90), worst)
        # against the uncondensed kernel: same statuses (up to the few instances that sit on a tolerance), same solution
        both = (qs == 0) & (qsu == 0)
        assert (st != stu).sum() <= max(1, int(0.02 * B)) and both.mean() > 0.9
        e = np.maximum(util.rel_err_per_instance(xg[both], xu[both]), util.rel_err_per_instance(ug[both], uu[both]))
        # (tick 0: the guess satisfies the dynamics, both cold starts coincide; later the two formulations start from different slacks
        # and stop at different points of the tolerance ball, which for this R = 0 model is ~3e-2 wide in the controls)
        assert (np.median(e) <= 1e-7 or tick > 0) and e.max() <= 5e-2, (tick, np.median(e), e.max())
        print("cond vs oracle", name, tick, "worst", worst, "iter off", off, "| vs uncondensed kernel: median", np.median(e), "max", e.max())
        # next tick: both solvers continue from the condensed solver's iterate, x0 <- x1
        x, u = xg, ug
        x0 = xg[:, 1].copy()
        for s in (sc, su_):
            s.set_all("x", x); s.set_all("u", u); s.set("x0", 0, x0)
    sc.close(); su_.close()


@pytest.mark.gpu
def test_condensing_kernel_solutions_are_certified_on_the_uncondensed_qp(oracle):
    """(x, u, pi, lam, t) of the condensed device solve against the KKT conditions of the ORIGINAL N-stage QP (tests/kkt.py),
    closed loop on the bench workload's generator: expansion and multiplier read-back included."""
    from tests.test_kkt_certify import _certify_closed_loop
    out = _certify_closed_loop(oracle, "usv_model_pf_ca", 40, 10, 512, 4, options=(("keep_multipliers", 1), ("qp_cond_N", 5)))
    assert out["kkt_certified_frac"] == 1.0 and out["active_row_frac"] > 0.3, out


@pytest.mark.gpu
def test_condensing_kernel_is_deterministic():
    """The team-parallel loops of cond_ipm.hpp are only executed serially in the CPU suite; a missing barrier on the device would show as
    run-to-run differences.  Two handles, same inputs, different numbers of resident teams: bit-identical results (every reduction has a
    fixed order - no atomics on floating-point data)."""
    name, N, K, B, N2 = "usv_model_pf_ca", 40, 10, 700, 10
    wl = scenario.make_bench_batch(name, N, K, B, seed=77)
    a, b = _cond_solver(name, N, K, B, wl, N2), _cond_solver(name, N, K, B, wl, N2, extra=(("max_waves", 97),))
    for tick in range(3):
        sa, sb = a.solve(), b.solve()
        assert np.array_equal(sa, sb) and np.array_equal(a.get_int("qp_iter"), b.get_int("qp_iter"))
        assert np.array_equal(a.get_all("x"), b.get_all("x")) and np.array_equal(a.get_all("u"), b.get_all("u"))
        assert np.array_equal(a.get_all("pi"), b.get_all("pi"), equal_nan=True)   # (a QP that ended in NaNs leaves NaN multipliers, on both)
        for s in (a, b):
            s.advance(1e-3, seed=5 + tick)
    a.close(); b.close()


@pytest.mark.gpu
def test_condensed_rti_then_full_sqp_and_multiplier_read_back(oracle):
    """Mixed use on one handle: RTI solves on the condensed QP, "lam" / "t" only once their buffers exist, then a full SQP - which
    runs on the uncondensed stages and must start from zero multipliers (the condensed solve leaves none in the workspace) - ends
    where the full SQP of a handle that never condensed ends."""
    name, N, K, B, N2 = "usv_model_pf_ca", 20, 4, 48, 5
    wl = scenario.make_bench_batch(name, N, K, B, seed=21)
    sc, su_ = _cond_solver(name, N, K, B, wl, N2), _cond_solver(name, N, K, B, wl, 0)
    assert (sc.solve() == su_.solve()).all()
    with pytest.raises(Exception):
        sc.get_all("lam")                      # the buffers did not exist during that solve (they do from now on)
    st = sc.solve(); su_.solve()
    lam, t = sc.get_all("lam"), sc.get_all("t")
    ok = st == 0
    assert ok.mean() > 0.9 and (lam[ok] >= 0).all() and (t[ok] >= 0).all() and float((lam[ok] * t[ok]).max()) <= 1e-7
    assert np.abs(lam[ok][:, 1:N]).max() > 0 and float(np.abs(lam[:, N]).max()) == 0.0      # rows exist on stages 0 .. N-1 only
    # both handles continue from the condensed handle's iterate
    for s in (sc, su_):
        s.set_all("x", sc.get_all("x")); s.set_all("u", sc.get_all("u"))
    a, b = sc.solve_sqp(), su_.solve_sqp()
    assert (a == b).all()
    both = (a == 0)
    assert both.mean() > 0.8
    assert util.rel_err(sc.get_all("x")[both], su_.get_all("x")[both]) < 1e-6 and util.rel_err(sc.get_all("u")[both], su_.get_all("u")[both]) < 1e-5
    sc.close(); su_.close()


@pytest.mark.gpu
@pytest.mark.parametrize("N,K,B,N2", [(40, 10, 256, 8), (100, 8, 64, 20), (40, 20, 96, 6)])   # (the bench shape; the ROS node's N = 100, K = 8; two chunks, blocks of 7 7 7 7 6 6)
def test_condensing_kernel_with_soft_rows_on_the_device(oracle, N, K, B, N2):
    """usv_model_guidance_ca1 - the model of the reference's ROS node - with qp_solver_cond_N applied.  No condensing oracle for soft
    rows, so: tick 0 (both cold starts coincide) against the uncondensed kernel and the C oracle - statuses, iteration counts, iterates and
    slacks; closed loop: every converged solve KKT-certified on the original QP (tests/test_kkt_certify.py machinery)."""
    name = "usv_model_guidance_ca1"
    wl = scenario.make_bench_batch(name, N, K, B, seed=1234)
    dt, steps = scenario.BENCH_DT, scenario.BENCH_SIM_STEPS[name]
    spec = oracle.spec(util.MODEL_ID[name], N, N * dt, K, sim_steps=steps)
    sc, su_ = _cond_solver(name, N, K, B, wl, N2), _cond_solver(name, N, K, B, wl, 0)
    st, stu = sc.solve(), su_.solve()
    xo, uo, sto, ito = util.oracle_rti(oracle, spec, wl, wl["x_init"].copy(), wl["u_init"].copy())
    assert np.array_equal(st, stu) and np.array_equal(st, sto)
    assert (np.abs(sc.get_int("qp_iter") - ito) > 1).sum() <= max(1, int(0.02 * B))
    ok = (st == 0) & (sc.get_int("qp_iter") == ito)
    assert ok.mean() > 0.9
    assert util.rel_err(sc.get_all("x")[ok], xo[ok]) < 1e-7 and util.rel_err(sc.get_all("u")[ok], uo[ok]) < 1e-7
    assert util.rel_err(sc.get_all("x")[ok], su_.get_all("x")[ok]) < 1e-7
    assert np.abs(sc.get_all("sl")[ok] - su_.get_all("sl")[ok]).max() < 1e-7 and np.abs(sc.get_all("su")[ok] - su_.get_all("su")[ok]).max() < 1e-7
    sc.close(); su_.close()
    from tests.test_kkt_certify import _certify_closed_loop
    out = _certify_closed_loop(oracle, name, N, K, B, 3, options=(("keep_multipliers", 1), ("qp_cond_N", N2)))
    assert out["kkt_certified_frac"] == 1.0, out


@pytest.mark.gpu
def test_condensing_option_is_refused_where_it_is_not_built():
    wl = scenario.make_bench_batch("usv_model_pf_ca", 20, 4, 8, seed=3)
    wl40 = scenario.make_bench_batch("usv_model_pf_ca", 40, 4, 8, seed=3)
    with pytest.raises(Exception, match="at most 64"):
        _cond_solver("usv_model_pf_ca", 40, 4, 8, wl40, 1)    # 14 + 40 * 2 = 94 variables in the one block
    s = _cond_solver("usv_model_pf_ca", 20, 4, 8, wl, 20)   # N2 = N: no condensing, the default path
    assert (s.solve() == 0).all()
    s.close()


@pytest.mark.gpu
def test_config4_shape_hip_vs_condensing_oracle(oracle):
    N, K, B, N2 = 80, 20, 256, 10
    wl = scenario.make_bench_batch(NAME, N, K, B, moving=True, seed=1234)
    assert not np.array_equal(wl["p"][:, 0], wl["p"][:, N])
    dt, steps = scenario.BENCH_DT, scenario.BENCH_SIM_STEPS[NAME]

    def solver():
        ocp = usv_models.make_ocp(NAME, N * dt, N, K)
        ocp.solver_options.sim_method_num_steps = steps
        # (qp_solver_cond_N left at N: this test holds the UNCONDENSED kernel against the condensing oracle)
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


def test_condensing_kernels_in_the_library_have_the_resources_the_design_counts_on():
    """docs/rounds/r06.md section 7 / DESIGN.md section 6: the hard-row condensing kernels are compiled for four waves per SIMD (128 registers;
    with the thread index opaque per block they spill a few dozen at most, 172 before), the soft-row ones for three (168), and the instantiation with
    the compile-time block shape of BASELINE configs[4] (8 stages per block, 7 touched states) exists next to the run-time-shape one."""
    import os, subprocess, sys
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = _capi.lib_path()
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "so_resources.py"), lib, "usv_qp_cond"], capture_output=True, text=True).stdout
    rows = {}
    for line in out.splitlines():
        name = line[:line.index(" vgpr ")].strip()
        f = line[line.index(" vgpr "):].split()
        rows[name] = {f[i]: f[i + 1] for i in range(0, len(f) - 1, 2)}
    fixed = [n for n in rows if n.endswith("256, 8, 7>")]
    assert sorted(fixed) == ["usv_qp_cond<ModelM2, 1, false, 256, 8, 7>", "usv_qp_cond<ModelM2, 2, false, 256, 8, 7>"], sorted(rows)
    for n, r in rows.items():
        soft = ", true, " in n
        assert int(r["vgpr"]) <= (168 if soft else 128), (n, r)
        assert int(r["spill"]) <= 64, (n, r)          # (reloads of spilled registers were what set the pace of this kernel)
    assert all(n.endswith(", 0, 0>") or n in fixed for n in rows), sorted(rows)


@pytest.mark.gpu
def test_condensed_solve_that_converged_is_not_reported_failed():
    """The backward sweep factorises while it forms the residuals; at a converged iterate that (unneeded) factorisation can meet a pivot that is
    not positive.  Until round 6 the kernel tested that flag before the convergence test - 10 of 8192 solves of BASELINE configs[4]'s workload
    came back with QP status 3 and residuals inside the tolerances, where oracle/condense.py and the uncondensed kernel say converged.  Now: no
    condensed solve with status 3 has all four residuals inside the tolerances, and the condensed kernel fails no more often than the uncondensed."""
    name, N, K, B, N2 = "usv_model_pf_ca", 80, 20, 4096, 10
    wl = scenario.make_bench_batch(name, N, K, B, seed=1234, moving=True)
    sc, su_ = _cond_solver(name, N, K, B, wl, N2), _cond_solver(name, N, K, B, wl, 0)
    for tick in range(2):
        sc.solve(); su_.solve()
        qc, qu, res = sc.get_int("qp_status"), su_.get_int("qp_status"), sc.get("res", 0)
        s3 = qc == 3
        inside = s3 & (res[:, 0] <= 1e-6) & (res[:, 1] <= 1e-8) & (res[:, 2] <= 1e-8) & (res[:, 3] <= 1e-8)
        assert s3.sum() > 0 and inside.sum() == 0, (tick, int(s3.sum()), int(inside.sum()))   # (the workload has infeasible instances: status 3 occurs)
        assert (qc == 0).sum() >= (qu == 0).sum() - B // 200, (tick, int((qc == 0).sum()), int((qu == 0).sum()))
        both = (qc == 0) & (qu == 0)
        if tick == 0:   # (both cold starts coincide; from the second tick on the two formulations' multiplier warm starts differ)
            assert both.mean() > 0.9 and (sc.get_int("qp_iter")[both] == su_.get_int("qp_iter")[both]).mean() > 0.995
        sc.advance(1e-3, seed=7 + tick); su_.advance(1e-3, seed=7 + tick)
    sc.close(); su_.close()
