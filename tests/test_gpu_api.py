"""C-ABI behaviour on the device: edge sizes, error codes, failure statuses, zero-copy views."""
import ctypes as C

import numpy as np
import pytest

from mpc_collisionavoidance_amd import BatchOcpSolver, scenario, sharding, usv_models
from tests import util

pytestmark = pytest.mark.gpu


def _mk(name, N, K, B, seed=1):
    ocp, wl = util.make(name, N, K, B, seed=seed)
    s = BatchOcpSolver(ocp, B)
    scenario.load_into(s, wl)
    return ocp, wl, s


@pytest.mark.parametrize("name", ["usv_model_guidance_ca1", "usv_model_pf_ca"])
def test_no_obstacles_and_max_obstacles(oracle, name):
    for K in (0, 32):
        ocp, wl, s = _mk(name, 8, K, 6, seed=K + 3)
        st = s.solve()
        spec = util.oracle_spec(oracle, name, 8, scenario.DT[name], K)
        xo, uo, sto, _ = util.oracle_rti(oracle, spec, wl, wl["x_init"], wl["u_init"])
        ok = (sto == 0) & (s.get_int("qp_status") == 0)
        assert ok.sum() >= 4 and np.array_equal(st[ok], sto[ok])
        assert util.rel_err(s.get_all("x")[ok], xo[ok]) < 1e-7 and util.rel_err(s.get_all("u")[ok], uo[ok]) < 1e-7
        s.close()
    with pytest.raises(Exception):
        usv_models.make_ocp(name, 0.4, 8, 33) and BatchOcpSolver(usv_models.make_ocp(name, 0.4, 8, 33), 2)


def test_minimal_sizes(oracle):
    ocp, wl, s = _mk("usv_model", 2, 0, 1)
    st = s.solve()
    spec = util.oracle_spec(oracle, "usv_model", 2, 0.05, 0)
    xo, uo, sto, _ = util.oracle_rti(oracle, spec, wl, wl["x_init"], wl["u_init"])
    assert st[0] == sto[0] == 0 and util.rel_err(s.get_all("x"), xo) < 1e-9
    s.close()
    with pytest.raises(RuntimeError):
        BatchOcpSolver(usv_models.make_ocp("usv_model", 0.05, 1), 1)  # N must be >= 2


def test_error_codes_and_messages():
    ocp, wl, s = _mk("usv_model_pf_ca", 6, 3, 4)
    lib, h = s._lib, s._h
    buf = np.zeros(4 * 64)
    p = buf.ctypes.data_as(C.POINTER(C.c_double))
    assert lib.usvmpc_set(h, b"bogus", 0, p, 14) == -2
    assert b"unknown field" in lib.usvmpc_last_error(h)
    assert lib.usvmpc_set(h, b"x", 7, p, 14) == -3          # stages 0..N
    assert lib.usvmpc_set(h, b"u", 6, p, 2) == -3           # stages 0..N-1
    assert lib.usvmpc_set(h, b"x", 0, p, 13) == -4
    assert b"mismatching dimension" in lib.usvmpc_last_error(h)
    assert lib.usvmpc_get(h, b"pi", 0, p, 14) == -3         # pi lives on stages 1..N
    assert lib.usvmpc_get(h, b"pi", 6, p, 14) == 0
    assert lib.usvmpc_set(h, b"pi", 1, p, 14) == -2         # outputs are not settable
    assert lib.usvmpc_set_option(h, b"nope", 1.0) == -2
    assert lib.usvmpc_get_int(h, b"nope", buf.ctypes.data_as(C.POINTER(C.c_int))) == -2
    with pytest.raises(Exception, match="mismatching dimension"):
        s.set("yref", 0, np.zeros((4, 3)))
    s.close()


def test_nan_input_gives_status_4_and_leaves_iterate_untouched():
    ocp, wl, s = _mk("usv_model_guidance_ca1", 8, 4, 8)
    x0 = wl["x0"].copy()
    x0[3, 2] = np.nan
    s.set("x0", 0, x0)
    st = s.solve()
    assert st[3] == 4 and s.get_int("qp_status")[3] == 3
    assert np.array_equal(s.get_all("x")[3], wl["x_init"][3]) and np.array_equal(s.get_all("u")[3], wl["u_init"][3])
    good = np.arange(8) != 3
    assert (st[good] == 0).all() and np.isfinite(s.get_all("x")[good]).all()
    s.close()


def test_repeated_rti_iterations_converge_and_are_deterministic():
    outs = []
    for rep in range(2):
        ocp, wl, s = _mk("usv_model_pf_ca", 20, 5, 64, seed=9)
        steps = []
        for it in range(6):
            before = s.get_all("x")
            s.solve()
            steps.append(np.abs(s.get_all("x") - before).max())
        outs.append((s.get_all("x"), s.get_all("u"), steps))
        s.close()
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])
    steps = outs[0][2]
    assert steps[-1] < 1e-3 * steps[0]  # fixed inputs: the SQP iteration contracts


def test_zero_copy_torch_views_and_u0_gather_shape():
    import torch
    ocp, wl, s = _mk("usv_model_pf_ca", 10, 3, 16)
    s.solve()
    xt = sharding.device_tensor(s.device_ptr("x"), (s.B, s.N + 1, s.nx))
    assert xt.is_cuda and np.array_equal(xt.cpu().numpy(), s.get_all("x"))
    u0 = sharding.first_controls_view(s)
    assert tuple(u0.shape) == (16, 2) and np.array_equal(u0.cpu().numpy(), s.get("u", 0))
    st = torch.as_tensor(np.zeros(1))  # keep torch import used
    assert st.numel() == 1
    s.close()


def test_closed_loop_advance_matches_host_round_trip():
    """usvmpc_advance == x0 = get(1,'x'); set(0,'lbx',x0) (scripts/usv_guidance_ca1/main.py:169-175)."""
    ocp, wl, a = _mk("usv_model_guidance_ca1", 12, 4, 32, seed=4)
    _, _, b = _mk("usv_model_guidance_ca1", 12, 4, 32, seed=4)
    for it in range(3):
        a.solve(); b.solve()
        a.advance()
        b.set("x0", 0, b.get("x", 1))
    a.solve(); b.solve()
    assert np.array_equal(a.get_all("x"), b.get_all("x")) and np.array_equal(a.get_all("u"), b.get_all("u"))
    a.close(); b.close()


def test_stage0_bounds_move_x0_and_must_coincide():
    """set(0, "lbx") / set(0, "ubx") are the x0 embedding: every write moves x0 (a caller that only rewrites one of the two
    each tick is not silently ignored), and solve() refuses to run while the two disagree."""
    from mpc_collisionavoidance_amd import usv_models
    constraint, model, solver = usv_models.acados_settings(1.0, 10, name="usv_model", n_obstacles=None)
    xa = np.array([0.5, 0.0, 0.0, 1.0, 1.0])
    xb = np.array([0.6, 0.01, 0.0, 2.0, 2.0])
    solver.set(0, "lbx", xa)
    solver.set(0, "ubx", xa)
    assert solver.solve() == 0 and np.allclose(solver.get(0, "x"), xa, atol=1e-9)
    solver.set(0, "lbx", xb)                       # only one of the two: x0 follows, but the pair is inconsistent
    xkeep = solver.get(0, "x")
    with pytest.warns(UserWarning, match="lbx and ubx of stage 0 differ"):
        assert solver.solve() == 4                 # solve() never raises: status 4, iterate untouched
    assert np.array_equal(solver.get(0, "x"), xkeep)
    solver.set(0, "ubx", xb)
    assert solver.solve() == 0 and np.allclose(solver.get(0, "x"), xb, atol=1e-9)


@pytest.mark.parametrize("name,N,K,B", [("usv_model_pf_ca", 20, 3, 96), ("usv_model_guidance_ca1", 20, 8, 33), ("usv_model_pf_ca", 40, 10, 40)])
def test_workspace_in_lds_matches_workspace_in_hbm(name, N, K, B):
    """Option lds_workspace: the same solve with the per-stage planes in LDS (small batches) - a separate instantiation of the same
    sweeps: the same bits (qp_ipm.hpp contracts multiply-adds by the language rule, not by instantiation: round 5)."""
    from mpc_collisionavoidance_amd import usv_models
    wl = scenario.make_bench_batch(name, N, K, B, seed=21)
    ocp = usv_models.make_ocp(name, N * scenario.BENCH_DT, N, K)
    ocp.solver_options.sim_method_num_steps = scenario.BENCH_SIM_STEPS[name]
    out = []
    for mode in (0, 1):
        s = BatchOcpSolver(ocp, B)
        scenario.load_into(s, wl)
        s.set_option("wide", 0)
        s.set_option("lds_workspace", mode)
        st = s.solve()
        out.append((s.get_all("x"), s.get_all("u"), st.copy(), s.get_int("qp_iter"), s.get_all("pi"), s.get_all("lam"), s.get_all("t")))
        s.close()
    for a, b in zip(out[0], out[1]):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("name,N,K,B", [("usv_model_pf_ca", 20, 3, 96), ("usv_model_guidance_ca1", 20, 10, 33), ("usv_model_pf_ca", 12, 20, 17),
                                        ("usv_model_pf_ca", 40, 9, 40)])
def test_one_row_pass_matches_two(name, N, K, B):
    """Option merge_box_rows (default on whenever every box row rides in an idle lane of the last obstacle chunk): the box rows
    processed in place, as rows of that chunk, against the two-pass form - same statuses and iteration counts, iterates equal
    to rounding (the lane in which a row's share of the complementarity sums is accumulated differs)."""
    from mpc_collisionavoidance_amd import usv_models
    wl = scenario.make_bench_batch(name, N, K, B, seed=22)
    ocp = usv_models.make_ocp(name, N * scenario.BENCH_DT, N, K)
    ocp.solver_options.sim_method_num_steps = scenario.BENCH_SIM_STEPS[name]
    out = []
    for mode in (0, 1):
        s = BatchOcpSolver(ocp, B)
        scenario.load_into(s, wl)
        s.set_option("merge_box_rows", mode)
        st = s.solve()
        out.append((s.get_all("x"), s.get_all("u"), st.copy(), s.get_int("qp_iter")))
        s.close()
    assert np.array_equal(out[0][2], out[1][2]) and np.abs(out[0][3] - out[1][3]).max() <= 1
    ok = out[0][2] == 0
    assert util.rel_err(out[1][0][ok], out[0][0][ok]) < 1e-7 and util.rel_err(out[1][1][ok], out[0][1][ok]) < 1e-6


def test_host_mirror_staging_is_invisible(oracle):
    """Small handles keep a pinned host mirror of the caller-visible arrays: setters write it, the next launch uploads the dirty
    fields, x / u / status come back with the solve (include/usvmpc.h, option host_mirror).  Every interleaving of per-stage /
    whole-field set, get, solve, advance and device-pointer access must behave exactly like the direct path (mirror off)."""
    name, N, K, B = "usv_model_guidance_ca1", 12, 4, 3
    ocp, wl = util.make(name, N, K, B, seed=41)
    rng = np.random.default_rng(3)

    def script(mirror):
        s = BatchOcpSolver(ocp, B)
        if not mirror:
            s.set_option("host_mirror", 0)
        out = []
        scenario.load_into(s, wl)
        out.append(s.get_all("x"))                       # get of a field that is dirty in the mirror, before any solve
        for k in range(N):                               # the reference protocol: per-stage setters
            s.set("yref", k, wl["yref"][:, k])
            s.set("p", k, wl["p"][:, k])
            s.set("lh", k, wl["lh"][:, k])
        s.set("p", N, wl["p"][:, N])
        s.set("yref", N, wl["yref_e"])
        s.set("x0", 0, wl["x0"])
        out.append(s.solve().copy())
        out += [s.get("x", 1), s.get("u", 0), s.get_all("x"), s.get_all("u"), s.get("pi", 3)]
        xm = s.get_all("x")
        xm[:, 5] += 0.01
        s.set("x", 5, xm[:, 5])                           # partial write of an output field, read back whole and per stage
        out += [s.get_all("x"), s.get("x", 5), s.get("x", 6)]
        s.advance(0.0)                                    # device-side write of x0 behind the mirror
        out.append(s.get("x0", 0))
        s.set("lh", 2, wl["lh"][:, 2] * 0.9)              # one stage dirty, the rest of the field untouched
        out.append(s.solve().copy())
        out += [s.get_all("x"), s.get_all("u"), s.get_all("lh")]
        s.set("x0", 0, wl["x0"] + 0.01)
        s.advance(0.0)                                    # the pending x0 is uploaded, then overwritten by the hand-over
        out.append(s.get("x0", 0))
        t = sharding.device_tensor(s.device_ptr("u"), (B, N, s.nu))   # zero-copy access: the mirror stops vouching for x / u
        t += 0.125
        import torch
        torch.cuda.synchronize()
        out.append(s.get_all("u"))
        s.solve_async()
        s.set("yref", 0, wl["yref"][:, 0] + 0.001)        # a set while the solve (and its read-back) is in flight
        s.sync()
        out += [s.get_all("x"), s.get_int("status").copy()]
        out.append(s.solve().copy())
        out += [s.get_all("x"), s.get_all("u")]
        s.close()
        return out

    a, b = script(True), script(False)
    assert len(a) == len(b)
    for i, (p, q) in enumerate(zip(a, b)):
        assert np.array_equal(p, q), i


def test_batch_beyond_the_buffer_window_is_refused():
    """One stage of the workspace is addressed through a 32-bit buffer window: a batch whose stage exceeds 4 GiB is refused at
    creation (it used to wrap silently)."""
    ocp = usv_models.make_ocp("usv_model_pf_ca", 0.4, 8, 10)
    with pytest.raises(RuntimeError):
        BatchOcpSolver(ocp, 1400000)
    s = BatchOcpSolver(ocp, 8)
    assert s.device_bytes() > 0
    s.close()


def test_host_mirror_small_batch_reference_loop(oracle):
    """The mirror with more than one instance (per-stage setters are strided over the batch) and the closed-loop hand-over on the
    host as the reference does it: x0 = get(1, "x"); set(0, "lbx" / "ubx", x0) - against the oracle, 6 ticks."""
    name, N, K, B = "usv_model_guidance_ca1", 16, 5, 5
    ocp, wl = util.make(name, N, K, B, seed=12, dt=0.05)
    s = BatchOcpSolver(ocp, B)
    scenario.load_into(s, wl)
    spec = util.oracle_spec(oracle, name, N, 0.05, K)
    xo, uo, x0 = wl["x_init"].copy(), wl["u_init"].copy(), wl["x0"].copy()
    for t in range(6):
        for k in range(N):
            s.set("yref", k, wl["yref"][:, k]); s.set("p", k, wl["p"][:, k]); s.set("lh", k, wl["lh"][:, k])
        s.set("p", N, wl["p"][:, N])
        s.set("x0", 0, x0)
        st = s.solve()
        xo, uo, sto, _ = util.oracle_rti(oracle, spec, wl, xo, uo, x0=x0)
        assert np.array_equal(st, sto) and (st == 0).all()
        assert util.rel_err(s.get_all("x"), xo) < 1e-7 and util.rel_err(s.get("u", 0), uo[:, 0]) < 1e-7
        x0 = s.get("x", 1).copy()
    s.close()


@pytest.mark.parametrize("name,N,K", [("usv_model_pf_ca", 12, 4), ("usv_model_guidance_ca1", 12, 6), ("usv_model_guidance_ca1", 70, 3)])
def test_pipelined_lineariser_is_scheduling_only(name, N, K):
    """Option pipeline_linearize (default on for >= 16384 instances): the next tick's lineariser runs on a second stream in the tail
    of the QP launch, instance by instance as results become final, with a fix-up pass for the rest.  Against the un-pipelined
    sequence, tick by tick: iterates, statuses, iteration counts and the hand-over must be bit-identical - including ticks in
    front of which the caller replaced the iterate or the reference (the ahead-of-time linearisation must then be discarded) and
    a multiplier read-back while the second stream is busy."""
    from mpc_collisionavoidance_amd import usv_models
    B = 16384 + 7
    wl = scenario.make_bench_batch(name, N, K, B, seed=5)
    ocp = usv_models.make_ocp(name, N * scenario.BENCH_DT, N, K)
    ocp.solver_options.sim_method_num_steps = scenario.BENCH_SIM_STEPS[name]
    rng = np.random.default_rng(1)
    bump = 0.01 * rng.standard_normal(wl["x_init"].shape)

    def run(pipe):
        s = BatchOcpSolver(ocp, B)
        scenario.load_into(s, wl)
        s.set_option("static_obstacles", 1)
        s.set_option("disturbance_mask", scenario.NOISE_MASK[name])
        s.set_option("pipeline_linearize", pipe)
        out = []
        for t in range(14):
            if t == 5:      # the caller replaces the iterate between two ticks
                s.sync()
                s.set_all("x", s.get_all("x") + bump)
            if t == 8:      # ... or the reference of one stage
                s.set("yref", 3, wl["yref"][:, 3] * 1.01)
            s.solve_async()
            s.advance(1e-3, seed=50 + t)
            if t in (2, 9):
                s.sync()
                out.append(s.get_all("lam")[:64].copy())
            if t % 3 == 0 or t >= 10:
                s.sync()
                out += [s.get_all("x"), s.get_all("u"), s.get_int("status").copy(), s.get_int("qp_iter").copy(), s.get("x0", 0)]
        s.close()
        return out

    a, b = run(1), run(0)
    assert len(a) == len(b)
    for i, (p, q) in enumerate(zip(a, b)):
        assert np.array_equal(p, q), i


def test_lineariser_runs_ahead_only_for_callers_that_do_not_write_between_ticks():
    """a linearisation made ahead of time is thrown away by every caller write of x / u / yref, so the lineariser only runs ahead
    after two solves in a row without one.  A solve + advance loop (the bench's flow) uses every pass made ahead; the reference's protocol -
    yref set on every stage every tick (scripts/usv_guidance_ca1/main.py:123-130) - discards exactly the one that was in flight when it
    started writing and never causes another (usvmpc_pipeline_stats).  Results equal the un-pipelined sequence either way."""
    from mpc_collisionavoidance_amd import usv_models
    name, N, K, B = "usv_model_guidance_ca1", 10, 3, 16384
    wl = scenario.make_bench_batch(name, N, K, B, seed=9)
    ocp = usv_models.make_ocp(name, N * scenario.BENCH_DT, N, K)

    def run(pipe):
        s = BatchOcpSolver(ocp, B)
        scenario.load_into(s, wl)
        s.set_option("pipeline_linearize", pipe)
        stats = []
        for t in range(6):          # nothing written between the ticks
            s.solve_async()
            s.advance(0.0)
        s.sync()
        stats.append(s.pipeline_stats())
        for t in range(5):          # the reference's protocol: the reference goes in again before every solve
            s.set_all("yref", wl["yref"])
            s.solve_async()
            s.advance(0.0)
        s.sync()
        stats.append(s.pipeline_stats())
        out = (s.get_all("x"), s.get_all("u"), s.get_int("qp_iter").copy())
        s.close()
        return stats, out

    (quiet, writing), a = run(1)
    _, b = run(0)
    for p, q in zip(a, b):
        assert np.array_equal(p, q)
    assert quiet[0] >= 3 and quiet[1] == 0, quiet              # solves 3 .. 6 ran on a linearisation made ahead of time
    assert writing[1] == 1 and writing[0] == quiet[0], writing  # one pass in flight when the writes began, none launched afterwards
