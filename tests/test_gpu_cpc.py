"""HPIPM's conditional predictor-corrector on the device (option "cond_pred_corr"; qp_ipm.hpp QpIpm::solve; oracle
usv_opts.cond_pred_corr): an IPM iteration whose corrected step leaves the duality measure above cpc_factor x the predictor's is redone with
the centring-only step.  On in the default QP solver profile (every mode acados can select has it: include/usvmpc.h USVMPC_HPIPM_*), and since
round 6 built into EVERY kernel: the throughput mapping (planes in HBM / LDS, aux plane in LDS), the latency mapping (one / four waves, planes
in LDS / HBM), the hand-over's follow-up launch, the launches of a full SQP, the partially condensed solve.  Here:
* with a factor low enough that steps are refused all the time the mappings still return the same bits (the emulator's version:
  tests/test_cpc_emu.py);
* switched off - or with a factor nothing exceeds - the same kernels return the bits of the profile "R04";
* the device follows the oracle with the option as closely as it follows it without;
* "hpipm_mode" re-applies a profile to a live handle."""
import numpy as np
import pytest

from mpc_collisionavoidance_amd import BatchOcpSolver, _capi, scenario, usv_models
from tests import parity_rule, util
from tests.test_gpu_wide import _compare

pytestmark = pytest.mark.gpu

FORCED = (("cpc_factor", 0.6),)   # (HPIPM: 2) - most corrected steps are refused


def _make(name, N, K, B, opts, mode=None):
    wl = scenario.make_bench_batch(name, N, K, B, seed=1234)
    ocp = usv_models.make_ocp(name, N * scenario.BENCH_DT, N, K if name != "usv_model" else None)
    ocp.solver_options.sim_method_num_steps = scenario.BENCH_SIM_STEPS[name]
    if mode is not None:
        ocp.solver_options.hpipm_mode = mode
    s = BatchOcpSolver(ocp, B)
    scenario.load_into(s, wl)
    if K > 0:
        s.set_option("static_obstacles", 1)
    s.set_option("disturbance_mask", scenario.NOISE_MASK[name])
    for k, v in opts:
        s.set_option(k, v)
    return s, wl


@pytest.mark.parametrize("name,N,K,B,ticks,wa,ma", [
    ("usv_model_pf_ca", 20, 3, 1024, 4, 1, 1),                 # BASELINE configs[1], planes in LDS
    ("usv_model_pf_ca", 40, 10, 256, 3, 1, 1),                 # the headline layout (two row passes)
    ("usv_model_guidance_ca1", 30, 8, 200, 3, 1, 1),
    ("usv_model_guidance_ca1", 40, 10, 256, 3, 4, 4),          # four waves per instance
    ("usv_model_pf_ca", 40, 10, 64, 3, 4, 4),
    ("usv_model_guidance_ca1", 100, 8, 40, 3, 1, 1),           # planes in HBM (the node's own horizon)
    ("usv_model_guidance_ca1", 100, 8, 1, 4, 4, 4),
    ("usv_model_pf_ca", 99, 10, 8, 2, 1, 1),
    ("usv_model_pf_ca", 80, 20, 64, 3, 4, 4),                  # two obstacle chunks, BASELINE configs[4]'s OCP
    ("usv_model_pf_ca", 40, 20, 200, 3, 1, 1),
    ("usv_model_pf_ca", 20, 15, 100, 3, 1, 1),                 # box rows in planes of their own
    ("usv_model", 20, 0, 500, 3, 1, 1),
    ("usv_model_pf_ca", 20, 3, 4096, 2, 1, 1),                 # more instances than resident wide waves: through the queue
])
def test_mappings_return_the_same_bits_with_refusals_forced(name, N, K, B, ticks, wa, ma):
    _compare(name, N, K, B, ticks, opts_a=(("wide", 1), ("wide_waves", wa)) + FORCED, opts_b=(("wide", 0),) + FORCED, map_a=ma)


@pytest.mark.parametrize("name,N,K,B", [("usv_model_pf_ca", 20, 3, 400), ("usv_model_guidance_ca1", 20, 8, 300)])
def test_workspace_in_lds_with_refusals_forced(name, N, K, B):
    a, _ = _make(name, N, K, B, (("wide", 0), ("lds_workspace", 1)) + FORCED)
    b, _ = _make(name, N, K, B, (("wide", 0), ("lds_workspace", 0), ("aux_in_lds", 0)) + FORCED)
    for t in range(3):
        sa, sb = a.solve(), b.solve()
        assert np.array_equal(sa, sb) and np.array_equal(a.get_int("qp_iter"), b.get_int("qp_iter"))
        for f in ("x", "u", "pi", "lam", "t"):
            assert np.array_equal(a.get_all(f), b.get_all(f), equal_nan=True), (t, f)
        a.advance(1e-3, seed=t); b.advance(1e-3, seed=t)
    a.close(); b.close()


@pytest.mark.parametrize("name,N,K,B,hand,opts", [
    ("usv_model_pf_ca", 40, 10, 10000, 6, ()),
    ("usv_model_pf_ca", 20, 3, 12000, 4, (("max_waves", 512),)),
    ("usv_model_guidance_ca1", 20, 8, 12000, 4, ()),
    ("usv_model_pf_ca", 40, 10, 10000, 6, (("handover_lds", 0),)),
    ("usv_model", 20, 0, 5000, 3, ()),
])
def test_handover_with_refusals_forced(name, N, K, B, hand, opts):
    """The pending step of a suspended solve may be a centring-only one: its flag rides in the hand-over record (QpIpm::suspend)."""
    a, _ = _make(name, N, K, B, (("wide", 0), ("lds_workspace", 0), ("handover_iter", 0)) + FORCED + tuple(opts))
    b, _ = _make(name, N, K, B, (("wide", 0), ("lds_workspace", 0), ("handover_iter", hand)) + FORCED + tuple(opts))
    handed = 0
    for t in range(3):
        sa, sb = a.solve(), b.solve()
        handed += int(b.handover_counts(1)[0])
        assert np.array_equal(sa, sb) and np.array_equal(a.get_int("qp_iter"), b.get_int("qp_iter")), t
        for f in ("x", "u", "pi", "lam", "t"):
            assert np.array_equal(a.get_all(f), b.get_all(f), equal_nan=True), (t, f)
        a.advance(1e-3, seed=5 + t); b.advance(1e-3, seed=5 + t)
    assert handed > 0
    a.close(); b.close()


@pytest.mark.parametrize("name,N,K,B,opts", [("usv_model_pf_ca", 40, 10, 600, ()), ("usv_model_pf_ca", 20, 3, 300, ()), ("usv_model_guidance_ca1", 20, 8, 200, ()),
                                             ("usv_model_pf_ca", 40, 10, 600, (("wide", 0), ("lds_workspace", 0)))])
def test_switched_off_the_kernels_return_the_bits_of_a_factor_nothing_exceeds(name, N, K, B, opts):
    """cond_pred_corr = 0 skips the test; cpc_factor = 1e30 runs it and never refuses: every so stays 1 and the results are the same bits
    (1 * x is x) - on whatever mapping the batch takes by default, and on the throughput mapping over planes in HBM."""
    a, _ = _make(name, N, K, B, (("cond_pred_corr", 0),) + tuple(opts))
    b, _ = _make(name, N, K, B, (("cpc_factor", 1e30),) + tuple(opts))
    c, _ = _make(name, N, K, B, tuple(opts))
    differs = False
    for t in range(3):
        sa, sb = a.solve(), b.solve()
        c.solve()
        assert a.last_mapping() == b.last_mapping()
        assert np.array_equal(sa, sb) and np.array_equal(a.get_int("qp_iter"), b.get_int("qp_iter"))
        for f in ("x", "u", "pi", "lam", "t"):
            assert np.array_equal(a.get_all(f), b.get_all(f), equal_nan=True), (t, f)
        differs = differs or not np.array_equal(a.get_all("u"), c.get_all("u"))
        for s in (a, b, c):
            s.advance(1e-3, seed=t)
    assert differs or name != "usv_model_pf_ca"   # (HPIPM's factor 2 does refuse steps on the hard-row workload)
    a.close(); b.close(); c.close()


def test_device_follows_the_oracle_with_the_option(oracle):
    name, N, K, B = "usv_model_pf_ca", 40, 10, 714
    dev, wl = _make(name, N, K, B, ())                              # the default profile: the option is on
    ref, _ = _make(name, N, K, B, (("cond_pred_corr", 0),))
    dt, steps = scenario.BENCH_DT, scenario.BENCH_SIM_STEPS[name]
    spec = oracle.spec(util.MODEL_ID[name], N, N * dt, K, sim_steps=steps)
    assert spec.opts.cond_pred_corr == 1 and spec.opts.mu0 == 1.0
    data = (wl["yref"], wl["yref_e"], wl["p"], wl["lh"])
    x0 = wl["x0"].copy()
    fired = agree = total = above = 0
    for t in range(4):
        xin, uin = dev.get_all("x"), dev.get_all("u")
        ref.set_all("x", xin); ref.set_all("u", uin); ref.set("x0", 0, x0)
        dev.solve(); ref.solve()
        xo, uo = xin.copy(), uin.copy()
        sto, ito = oracle.rti_batch(spec, xo, uo, x0, *data, threads=8)
        xg, ug, qs, qi = dev.get_all("x"), dev.get_all("u"), dev.get_int("qp_status"), dev.get_int("qp_iter")
        assert (dev.get_int("status") != sto).sum() <= 2
        ok = (qs == 0) & (sto == 0) & (ito < spec.opts.qp_iter_max)
        assert ok.mean() > 0.9
        agree += int((qi[ok] == ito[ok]).sum()); total += int(ok.sum())
        e = np.maximum(util.rel_err_per_instance(xg[ok], xo[ok]), util.rel_err_per_instance(ug[ok], uo[ok]))
        assert np.median(e) <= 1e-9 and np.percentile(e, 90) <= 1e-7, (t, np.median(e), np.percentile(e, 90))
        r = parity_rule.check(oracle, spec, dev, ok, e, xin, uin, x0, data, max_frac=0.02)   # (instances above 1e-5 carry the KKT certificate)
        above += r["above"]
        assert not r["violations"], (t, r)
        # the option did something: the device with it differs from the device without on some instances
        both = ok & (ref.get_int("qp_status") == 0)
        d = np.maximum(util.rel_err_per_instance(xg[both], ref.get_all("x")[both]), util.rel_err_per_instance(ug[both], ref.get_all("u")[both]))
        fired += int((d > 1e-9).sum())
        dev.advance(1e-3, seed=30 + t)
        dev.sync()
        x0 = dev.get("x0", 0)
    print("cond_pred_corr on the device: fired on %d instance-solves of %d, same iteration count as the oracle on %d of %d, above 1e-5 (certified) %d"
          % (fired, total, agree, total, above))
    assert fired >= 10 and agree >= 0.97 * total
    dev.close(); ref.close()


def test_profile_option_on_a_live_handle_and_refusals():
    """usvmpc_set_option "hpipm_mode" re-applies a profile (mu0, alpha_min, cond_pred_corr): a default handle switched to R04 returns the
    bits of a handle created under R04, and back."""
    name, N, K, B = "usv_model_pf_ca", 20, 3, 64
    a, wl = _make(name, N, K, B, (), mode="R04")
    b, _ = _make(name, N, K, B, ())
    c, _ = _make(name, N, K, B, ())
    b.set_option("hpipm_mode", _capi.HPIPM_MODES["R04"])
    sa, sb, sc = a.solve(), b.solve(), c.solve()
    assert np.array_equal(sa, sb) and np.array_equal(a.get_int("qp_iter"), b.get_int("qp_iter"))
    for f in ("x", "u", "pi", "lam", "t"):
        assert np.array_equal(a.get_all(f), b.get_all(f)), f
    assert not np.array_equal(a.get_int("qp_iter"), c.get_int("qp_iter"))      # (mu0 = 10 against 1: other iteration counts)
    b.set_option("hpipm_mode", _capi.HPIPM_MODES["SPEED"])                      # (on the device SPEED, BALANCE and ROBUST are one profile)
    b.set_all("x", wl["x_init"])
    b.set_all("u", wl["u_init"])
    sb = b.solve()
    assert np.array_equal(sb, sc) and np.array_equal(b.get_all("x"), c.get_all("x")) and np.array_equal(b.get_int("qp_iter"), c.get_int("qp_iter"))
    for bad in (("hpipm_mode", 7), ("hpipm_mode", 0.5), ("cpc_factor", 0.0)):
        with pytest.raises(Exception):
            b.set_option(*bad)
    # the partially condensed solve takes the option too (cond_ipm.hpp): accepted in both orders
    b.set_option("qp_cond_N", 5)
    b.set_option("cond_pred_corr", 1)
    b.solve()
    a.close(); b.close(); c.close()
