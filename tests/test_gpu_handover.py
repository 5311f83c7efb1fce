"""Hand-over of long runners on the device (option "handover_iter"; qp_ipm.hpp QpIpm::suspend / solve phase 3, usvmpc.hip usv_qp_resume).
A launch of more instances than the device holds rows hands them out through a queue and ends with a few rows finishing instances of 30 - 50
IPM iterations while the device idles.  With the option, a row of a drained launch whose instance has passed that many iterations leaves it
to a follow-up launch on the latency mapping (one instance per wavefront over the same workspace planes).  Scheduling only: statuses,
iteration counts, iterates, multipliers and slacks are the same BITS as without it (the mappings agree bit for bit: tests/test_gpu_wide.py;
the lane emulator runs the same hand-over: tests/test_wide_emu.py)."""
import numpy as np
import pytest

from mpc_collisionavoidance_amd import BatchOcpSolver, scenario, usv_models

pytestmark = pytest.mark.gpu


def _make(name, N, K, B, seed, opts):
    wl = scenario.make_bench_batch(name, N, K, B, seed=seed)
    ocp = usv_models.make_ocp(name, N * scenario.BENCH_DT, N, K if name != "usv_model" else None)
    ocp.solver_options.sim_method_num_steps = scenario.BENCH_SIM_STEPS[name]
    s = BatchOcpSolver(ocp, B)
    scenario.load_into(s, wl)
    if K > 0:
        s.set_option("static_obstacles", 1)
    s.set_option("disturbance_mask", scenario.NOISE_MASK[name])
    for k, v in opts:
        s.set_option(k, v)
    return s


@pytest.mark.parametrize("name,N,K,B,hand,opts", [
    ("usv_model_pf_ca", 20, 3, 12000, 8, ()),                          # one row pass (box rows in idle obstacle lanes), aux plane in LDS
    ("usv_model_pf_ca", 40, 10, 10000, 12, ()),                        # the headline layout: one box row in the aux plane, two row passes
    ("usv_model_pf_ca", 40, 10, 10000, 12, (("aux_in_lds", 0),)),      # ... aux plane streamed
    ("usv_model_guidance_ca1", 20, 8, 12000, 6, ()),                   # soft rows
    ("usv_model_guidance_ca1", 40, 16, 9000, 6, ()),                   # soft rows, two row passes
    ("usv_model", 20, 0, 12000, 5, (("max_waves", 1024),)),            # no obstacle rows (three waves per SIMD hold 12 288 rows: fewer, so that the queue is used)
    ("usv_model_pf_ca", 20, 3, 12000, 8, (("max_waves", 512),)),       # a quarter of the rows: most of the batch through the queue
    # (above: the follow-up launch copies the planes into LDS - its default where the horizon fits; below: over the planes in HBM)
    ("usv_model_pf_ca", 40, 10, 10000, 12, (("handover_lds", 0),)),
    ("usv_model_guidance_ca1", 20, 8, 12000, 6, (("handover_lds", 0),)),
    ("usv_model_guidance_ca1", 100, 8, 3000, 5, ()),                   # a horizon that does not fit LDS; every instance resident from the start (no queue)
    ("usv_model_pf_ca", 40, 10, 3000, 12, ()),                         # a mid-size batch: no queue, the launch is as long as its hardest instances
    ("usv_model", 20, 0, 5000, 5, ()),
])
def test_handover_does_not_change_a_bit(name, N, K, B, hand, opts):
    a = _make(name, N, K, B, 1234, (("wide", 0), ("lds_workspace", 0), ("handover_iter", 0)) + tuple(opts))
    b = _make(name, N, K, B, 1234, (("wide", 0), ("lds_workspace", 0), ("handover_iter", hand)) + tuple(opts))
    handed = 0
    for t in range(3):
        sa, sb = a.solve(), b.solve()
        assert a.last_mapping() == 0 and b.last_mapping() == 0
        assert int(a.handover_counts(1)[0]) == 0
        handed += int(b.handover_counts(1)[0])
        assert np.array_equal(sa, sb), t
        for f in ("qp_status", "qp_iter"):
            assert np.array_equal(a.get_int(f), b.get_int(f)), (t, f)
        fields = ("x", "u", "pi", "lam", "t", "res") + (("sl", "su") if name == "usv_model_guidance_ca1" else ())
        for f in fields:
            fa, fb = (a.get(f, 0), b.get(f, 0)) if f == "res" else (a.get_all(f), b.get_all(f))
            assert np.array_equal(fa, fb, equal_nan=True), (t, f, float(np.nanmax(np.abs(fa - fb))))
        assert int(a.unconverged_counts(1)[0]) == int(b.unconverged_counts(1)[0]) and int(a.fail_counts(1)[0]) == int(b.fail_counts(1)[0])
        a.advance(1e-3, seed=5 + t)
        b.advance(1e-3, seed=5 + t)
    assert handed > 0, "no instance was handed over: the test did not exercise the follow-up launch"
    assert np.array_equal(a.get("x0", 0), b.get("x0", 0))
    print("handed over", name, N, K, B, handed)
    a.close()
    b.close()


def test_handover_with_the_pipelined_lineariser():
    """Large handles run the next tick's lineariser in the tail of the QP launch, instance by instance as results become final
    (pipeline_linearize): the follow-up launch publishes its instances' epochs like the main launch does."""
    name, N, K, B = "usv_model_pf_ca", 20, 3, 20000
    a = _make(name, N, K, B, 7, (("handover_iter", 0),))
    b = _make(name, N, K, B, 7, (("handover_iter", 10),))
    for t in range(6):
        a.solve_async(); a.advance(1e-3, seed=t)
        b.solve_async(); b.advance(1e-3, seed=t)
    a.sync(); b.sync()
    assert b.pipeline_stats()[0] >= 3                 # the lineariser did run ahead
    assert int(b.handover_counts(4).sum()) > 0
    for f in ("x", "u", "pi"):
        assert np.array_equal(a.get_all(f), b.get_all(f), equal_nan=True), f
    assert np.array_equal(a.get_int("qp_iter"), b.get_int("qp_iter")) and np.array_equal(a.get("x0", 0), b.get("x0", 0))
    a.close()
    b.close()


def test_default_hands_over_where_the_follow_up_works_in_lds():
    """Default ("handover_iter" = -1): past 20 iterations when the horizon's planes fit a CU's LDS (the follow-up copies them in) and the batch is
    at most three times what the device holds at once - with the follow-up kernel beside the launch ("handover_co") -, never when it would have
    to work over the planes in HBM, never for the large batches (re-measured in round 6: profiles/r06_handover_co.txt); the time of the
    follow-up launch behind the main one is reported apart (usvmpc_followup_ms)."""
    for name, N, K, B, want in (("usv_model_pf_ca", 40, 10, 6000, True), ("usv_model_pf_ca", 100, 4, 3000, False), ("usv_model_pf_ca", 40, 20, 3000, False),
                                ("usv_model_pf_ca", 40, 10, 30000, False)):
        s = _make(name, N, K, B, 11, (("wide", 0), ("lds_workspace", 0)))
        for t in range(3):
            s.solve(); s.advance(1e-3, seed=t)
        s.sync()
        handed, fu, it = int(s.handover_counts(3).sum()), s.followup_ms(3), s.get_int("qp_iter")
        assert (handed > 0) == want and (fu.max() > 0.0) == want, (name, N, K, handed, fu)
        if want:
            assert handed < B and (it >= 20).any() and fu.max() < s.kernel_ms(3)[1].max()   # (the cold first ticks run longer than the last one)
            assert int(s.handover_co_counts(3)[0].sum()) > 0                                  # (the kernel beside the launch did finish some)
        s.close()


@pytest.mark.parametrize("name,N,K,B,hand,opts", [
    ("usv_model_pf_ca", 40, 10, 20000, 12, ()),                        # the headline layout, most of the batch through the queue
    ("usv_model_pf_ca", 40, 10, 20000, 24, ()),                        # ... at the default threshold
    ("usv_model_pf_ca", 20, 3, 12000, 8, (("max_waves", 512),)),
    ("usv_model_guidance_ca1", 20, 8, 12000, 6, ()),                   # soft rows
    ("usv_model", 20, 0, 12000, 5, (("max_waves", 1024),)),
    ("usv_model_pf_ca", 40, 10, 3000, 12, ()),                         # a mid-size batch: no queue, every instance resident from the start
    ("usv_model_pf_ca", 40, 10, 20000, 12, (("handover_co_wgs", 16),)),   # few co-resident workgroups: most entries are left to the launch behind
    ("usv_model_pf_ca", 40, 10, 20000, 12, (("handover_co_spin", 1),)),   # waits that give up at once: the launch behind does (nearly) everything
])
def test_co_resident_follow_up_does_not_change_a_bit(name, N, K, B, hand, opts):
    """Option "handover_co" (usvmpc.hip usv_qp_resume_co): the follow-up kernel runs BESIDE the draining launch - entries published one by one
    with an agent-scope release, claimed by compare-and-swap, finished on the latency mapping while the main launch still runs; what it does not
    get to is done by the follow-up launch behind the main one.  Scheduling only - but this is the one hand-over whose producer and consumer run
    at the same time on different XCDs: every output must equal the run without any hand-over, bit for bit, tick after tick."""
    a = _make(name, N, K, B, 1234, (("wide", 0), ("lds_workspace", 0), ("handover_iter", 0)))
    b = _make(name, N, K, B, 1234, (("wide", 0), ("lds_workspace", 0), ("handover_iter", hand), ("handover_co", 1)) + tuple(opts))
    handed = co = 0
    for t in range(5):
        sa, sb = a.solve(), b.solve()
        handed += int(b.handover_counts(1)[0])
        fin, tmo = b.handover_co_counts(1)
        co += int(fin[0])
        assert fin[0] <= b.handover_counts(1)[0]
        assert np.array_equal(sa, sb), (t, "statuses differ on %d instances; handed %d, beside %d, timeouts %d" % ((sa != sb).sum(), b.handover_counts(1)[0], fin[0], tmo[0]))
        for f in ("qp_status", "qp_iter"):
            assert np.array_equal(a.get_int(f), b.get_int(f)), (t, f)
        fields = ("x", "u", "pi", "lam", "t", "res") + (("sl", "su") if name == "usv_model_guidance_ca1" else ())
        for f in fields:
            fa, fb = (a.get(f, 0), b.get(f, 0)) if f == "res" else (a.get_all(f), b.get_all(f))
            assert np.array_equal(fa, fb, equal_nan=True), (t, f, float(np.nanmax(np.abs(fa - fb))))
        a.advance(1e-3, seed=5 + t)
        b.advance(1e-3, seed=5 + t)
    assert handed > 0
    print("handed over", name, N, K, B, handed, "of which beside the launch", co)
    a.close()
    b.close()


def test_co_resident_follow_up_with_the_pipelined_lineariser():
    name, N, K, B = "usv_model_pf_ca", 20, 3, 20000
    a = _make(name, N, K, B, 7, (("handover_iter", 0),))
    b = _make(name, N, K, B, 7, (("handover_iter", 10), ("handover_co", 1)))
    for t in range(8):
        a.solve_async(); a.advance(1e-3, seed=t)
        b.solve_async(); b.advance(1e-3, seed=t)
    a.sync(); b.sync()
    assert b.pipeline_stats()[0] >= 3
    assert int(b.handover_counts(4).sum()) > 0
    for f in ("x", "u", "pi"):
        assert np.array_equal(a.get_all(f), b.get_all(f), equal_nan=True), f
    assert np.array_equal(a.get_int("qp_iter"), b.get_int("qp_iter")) and np.array_equal(a.get("x0", 0), b.get("x0", 0))
    a.close()
    b.close()


@pytest.mark.gpu
def test_default_policy_does_not_stall_over_many_ticks():
    """docs/rounds/r06.md section 8: with two follow-up workgroups per CU (the default until the end of round 6) about one tick in 1500 of this loop
    took 410 ms instead of 9 - the main launch did not end while follow-up workgroups that had found no room waited out their bound.  With one per CU
    (the default since): no tick anywhere near that, and no workgroup of the follow-up kernel runs into its bound."""
    import time
    name, N, K, B, ticks = "usv_model_pf_ca", 40, 10, 4096, 1500
    wl = scenario.make_bench_batch(name, N, K, B, seed=1234)
    ocp = usv_models.make_ocp(name, N * scenario.BENCH_DT, N, K)
    ocp.solver_options.sim_method_num_steps = scenario.BENCH_SIM_STEPS[name]
    s = BatchOcpSolver(ocp, B)
    scenario.load_into(s, wl)
    s.set_option("static_obstacles", 1)
    s.set_option("disturbance_mask", scenario.NOISE_MASK[name])
    for w in range(5):
        s.solve_async(); s.advance(1e-3, seed=100 + w)
    s.sync()
    t, finished, timeouts = np.zeros(ticks), 0, 0
    for k in range(ticks):
        t0 = time.perf_counter()
        s.solve_async(); s.advance(1e-3, seed=1000 + k)
        s.sync()
        t[k] = time.perf_counter() - t0
        fin, to = s.handover_co_counts(1)
        finished += int(fin[0]); timeouts += int(to[0])
    s.close()
    assert finished > 20 * ticks         # (the follow-up kernel beside the launch is at work in this loop: hundreds of instances per tick)
    assert timeouts == 0 and t.max() < 0.1 and t.max() < 8 * np.median(t), (timeouts, float(np.median(t)), float(t.max()))
