"""The latency mapping on the device (option "wide": ONE instance per wavefront, qp_ipm.hpp WIDE / usvmpc_last_mapping) against the
throughput mapping (four instances per wavefront).  The wide sweeps take every sum in the order of the 16-lane sweeps: on the lane
emulator, where no multiply-add is contracted, the two are bit-identical (tests/test_wide_emu.py).  On the device they are two
instantiations the compiler contracts differently, so - as for the LDS / HBM workspace pair - the comparison is: statuses and
iteration counts equal, iterates and multipliers equal to rounding, tick by tick FROM IDENTICAL INPUTS over a closed loop, with the
work queue (more instances than resident waves) and without, hard-row and soft-row model, down to ONE instance.
Parity of the wide mapping against the oracle at BASELINE configs[1]'s full size: tests/test_gpu_parity.py (its default there)."""
import numpy as np
import pytest

from mpc_collisionavoidance_amd import BatchOcpSolver, scenario, usv_models
from tests import util

pytestmark = pytest.mark.gpu


def _make(name, N, K, B, seed, opts):
    wl = scenario.make_bench_batch(name, N, K, B, seed=seed)
    ocp = usv_models.make_ocp(name, N * scenario.BENCH_DT, N, K)
    ocp.solver_options.sim_method_num_steps = scenario.BENCH_SIM_STEPS[name]
    s = BatchOcpSolver(ocp, B)
    scenario.load_into(s, wl)
    if K > 0:
        s.set_option("static_obstacles", 1)
    s.set_option("disturbance_mask", scenario.NOISE_MASK[name])
    for k, v in opts:
        s.set_option(k, v)
    return s


def _compare(name, N, K, B, ticks, opts_a=(("wide", 1), ("wide_waves", 1)), opts_b=(("wide", 0),), seed=1234, map_a=1, map_b=0):
    a, b = _make(name, N, K, B, seed, opts_a), _make(name, N, K, B, seed, opts_b)
    worst = 0.0
    for t in range(ticks):
        sa, sb = a.solve(), b.solve()
        assert a.last_mapping() == map_a and b.last_mapping() == map_b
        qa, qb = a.get_int("qp_iter"), b.get_int("qp_iter")
        assert np.array_equal(sa, sb) and np.array_equal(a.get_int("qp_status"), b.get_int("qp_status"))
        # (an instance whose exit test is passed by a hair's breadth on one side may take one iteration more on the other)
        assert (qa != qb).sum() <= max(1, B // 200) and np.abs(qa - qb).max() <= 1, (t, np.where(qa != qb)[0])
        ok = (sa == 0) & (qa == qb)
        assert ok.mean() > 0.9 or B < 8
        xa, ua = a.get_all("x"), a.get_all("u")
        e = np.maximum(util.rel_err_per_instance(xa[ok], b.get_all("x")[ok]), util.rel_err_per_instance(ua[ok], b.get_all("u")[ok]))
        # Rounding differences of a few ulp.  The soft-row model keeps them; the hard-row model (control weight R = 0) amplifies them on
        # its rounding-sensitive QPs exactly as it does between the device and its own emulator (tests/test_parity_outliers.py): median
        # and 90th percentile tight, every instance inside the parity rule's cap
        if name == "usv_model_pf_ca" and e.size < 20:
            assert e.max() <= 1e-6, (t, e)
            tight = e <= 1e-7
        elif name == "usv_model_pf_ca":
            assert np.median(e) <= 1e-10 and np.percentile(e, 90) <= 1e-7 and e.max() <= 5e-3, (t, np.median(e), np.percentile(e, 90), e.max())
            tight = e <= 1e-7
        else:
            assert e.max() <= 1e-7, (t, e.max())
            tight = np.ones(e.shape, bool)
        for f in ("pi", "lam", "t"):
            fa, fb = a.get_all(f)[ok][tight], b.get_all(f)[ok][tight]
            assert np.abs(fa - fb).max() <= 1e-5 * max(1.0, np.abs(fb).max()), (t, f)
        worst = max(worst, float(np.percentile(e, 99)))
        a.advance(1e-3, seed=77 + t)
        a.sync()
        # both continue from the wide side's state
        b.set("x0", 0, a.get("x0", 0))
        b.set_all("x", xa)
        b.set_all("u", ua)
    a.close()
    b.close()
    return worst


@pytest.mark.parametrize("name,N,K,B,ticks", [("usv_model_pf_ca", 20, 3, 1024, 6),          # BASELINE configs[1]
                                               ("usv_model_guidance_ca1", 20, 3, 1024, 4),
                                               ("usv_model_pf_ca", 40, 9, 300, 3),
                                               ("usv_model_guidance_ca1", 30, 8, 200, 3),
                                               ("usv_model_pf_ca", 40, 10, 256, 3),           # (the headline layout: one box row in the aux plane, two row passes)
                                               ("usv_model_guidance_ca1", 20, 16, 100, 3),
                                               ("usv_model_pf_ca", 40, 10, 1, 4),
                                               ("usv_model_pf_ca", 20, 3, 1, 5),              # the reference's shape: one instance
                                               ("usv_model_pf_ca", 21, 3, 7, 3),              # (horizon not a multiple of the block of four)
                                               # horizons whose planes do not fit a CU's LDS: the wide sweeps over planes in HBM
                                               ("usv_model_guidance_ca1", 100, 8, 1, 4),      # the reference node's own shape (nmpc_guidance_ca1.cpp:64)
                                               ("usv_model_guidance_ca1", 100, 8, 40, 3),
                                               ("usv_model_guidance_ca1", 70, 16, 24, 2),     # (two row passes)
                                               ("usv_model_pf_ca", 100, 4, 16, 3),
                                               ("usv_model_pf_ca", 99, 10, 8, 2),
                                               ("usv_model", 20, 0, 1, 4),                    # BASELINE configs[0]'s shape: no obstacle rows
                                               ("usv_model", 20, 0, 500, 3), ("usv_model", 150, 0, 6, 2)])
def test_wide_mapping_equals_the_throughput_mapping(name, N, K, B, ticks):
    w = _compare(name, N, K, B, ticks)
    print("wide vs throughput mapping", name, N, K, B, "99th percentile of the relative difference %.2e" % w)


@pytest.mark.parametrize("name,N,K,B,ticks", [("usv_model_pf_ca", 20, 3, 200, 4), ("usv_model_guidance_ca1", 20, 3, 1, 4),
                                               ("usv_model_pf_ca", 40, 10, 64, 3), ("usv_model_guidance_ca1", 40, 10, 256, 3),
                                               ("usv_model_pf_ca", 21, 9, 5, 3), ("usv_model_guidance_ca1", 100, 8, 1, 4),
                                               ("usv_model_guidance_ca1", 100, 8, 30, 2), ("usv_model_pf_ca", 99, 10, 8, 2),
                                               ("usv_model_pf_ca", 20, 3, 600, 2),      # (more instances than CUs: the rest through the queue)
                                               ("usv_model", 20, 0, 3, 3)])
def test_four_waves_per_instance_equal_the_throughput_mapping(name, N, K, B, ticks):
    """Option wide_waves = 4 (usvmpc_last_mapping = 4): a workgroup of four wavefronts - a whole CU - per instance, the row work of 16
    consecutive stages at once; planes in LDS or (N = 99 / 100) in HBM."""
    w = _compare(name, N, K, B, ticks, opts_a=(("wide", 1), ("wide_waves", 4)), map_a=4)
    print("four waves per instance vs throughput mapping", name, N, K, B, "99th percentile of the relative difference %.2e" % w)


def test_wide_mapping_with_the_work_queue_and_without():
    """4096 instances are more than the device holds waves of the wide kernel: the rest comes through the queue.  dynamic_rows = 0:
    one workgroup per instance instead (the same kernel, so these two ARE bit-identical)."""
    name, N, K, B = "usv_model_pf_ca", 20, 3, 4096
    _compare(name, N, K, B, 2)
    a, b = _make(name, N, K, B, 9, (("wide", 1), ("wide_waves", 1))), _make(name, N, K, B, 9, (("wide", 1), ("wide_waves", 1), ("dynamic_rows", 0)))
    sa, sb = a.solve(), b.solve()
    assert a.last_mapping() == 1 and b.last_mapping() == 1
    assert np.array_equal(sa, sb)
    for f in ("x", "u", "pi", "lam", "t"):
        assert np.array_equal(a.get_all(f), b.get_all(f)), f
    assert np.array_equal(a.get_int("qp_iter"), b.get_int("qp_iter"))
    a.close()
    b.close()


def test_default_takes_the_wide_mapping_for_small_batches_only():
    name, N, K = "usv_model_pf_ca", 20, 3
    for B, want in ((64, 1), (1024, 1), (16384, 0)):
        s = _make(name, N, K, B, 5, ())
        s.solve()
        assert s.last_mapping() == want, (B, s.last_mapping())
        s.close()
    for B, want in ((1, 4), (256, 4), (300, 1)):   # (soft-row OCP, up to one instance per CU: four waves each)
        s = _make("usv_model_guidance_ca1", N, K, B, 5, ())
        s.solve()
        assert s.last_mapping() == want, (B, s.last_mapping())
        s.close()
    # a layout the wide sweeps do not cover (two obstacle chunks) stays on the throughput mapping
    s = _make(name, 20, 20, 64, 5, (("wide", 1),))
    s.solve()
    assert s.last_mapping() == 0
    s.close()
