"""Option "cond_pred_corr" on the device (HPIPM's conditional predictor-corrector: DESIGN.md section 2; qp_ipm.hpp QpIpm<.., CPC>::solve;
oracle usv_opts.cond_pred_corr): an IPM iteration whose corrected step leaves the duality measure above cpc_factor x the predictor's is
redone with the centring-only step.  Off by default on both sides; here both sides run with it, on the hard-row bench workload in closed
loop, where it fires on a few per cent of the instances and moves them by up to 1e-2 (another path into the tolerance ball of QPs with
control weight R = 0) - the device must follow the oracle WITH the option as closely as it follows the plain oracle without."""
import numpy as np
import pytest

from mpc_collisionavoidance_amd import BatchOcpSolver, scenario, usv_models
from tests import parity_rule, util

pytestmark = pytest.mark.gpu


def _make(name, N, K, B, opts):
    wl = scenario.make_bench_batch(name, N, K, B, seed=1234)
    ocp = usv_models.make_ocp(name, N * scenario.BENCH_DT, N, K)
    ocp.solver_options.sim_method_num_steps = scenario.BENCH_SIM_STEPS[name]
    s = BatchOcpSolver(ocp, B)
    scenario.load_into(s, wl)
    s.set_option("static_obstacles", 1)
    s.set_option("disturbance_mask", scenario.NOISE_MASK[name])
    for k, v in opts:
        s.set_option(k, v)
    return s, wl


@pytest.mark.parametrize("name,N,K,B", [("usv_model_pf_ca", 40, 10, 600), ("usv_model_pf_ca", 20, 3, 300), ("usv_model_guidance_ca1", 20, 8, 200)])
def test_a_factor_nothing_exceeds_leaves_the_plain_bits(name, N, K, B):
    """The CPC instantiations with the fallback never firing: the plain kernels' results bit for bit (the corrector target with so = 1)."""
    a, _ = _make(name, N, K, B, (("wide", 0), ("lds_workspace", 0)))
    b, _ = _make(name, N, K, B, (("cond_pred_corr", 1), ("cpc_factor", 1e30)))
    for t in range(3):
        sa, sb = a.solve(), b.solve()
        assert b.last_mapping() == 0        # (small batch: the option keeps the solve on the throughput mapping)
        assert np.array_equal(sa, sb) and np.array_equal(a.get_int("qp_iter"), b.get_int("qp_iter"))
        for f in ("x", "u", "pi", "lam", "t"):
            assert np.array_equal(a.get_all(f), b.get_all(f), equal_nan=True), (t, f)
        a.advance(1e-3, seed=t); b.advance(1e-3, seed=t)
    a.close(); b.close()


def test_device_follows_the_oracle_with_the_option(oracle):
    name, N, K, B = "usv_model_pf_ca", 40, 10, 714
    dev, wl = _make(name, N, K, B, (("cond_pred_corr", 1),))
    ref, _ = _make(name, N, K, B, (("wide", 0),))
    dt, steps = scenario.BENCH_DT, scenario.BENCH_SIM_STEPS[name]
    spec = oracle.spec(util.MODEL_ID[name], N, N * dt, K, sim_steps=steps, cond_pred_corr=1)
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
    print("cond_pred_corr on the device: fired on %d instance-solves of %d, same iteration count as the oracle with the option on %d of %d, above 1e-5 (certified) %d"
          % (fired, total, agree, total, above))
    assert fired >= 10 and agree >= 0.97 * total
    dev.close(); ref.close()


def test_refusals():
    s, _ = _make("usv_model_pf_ca", 20, 3, 8, ())
    s.set_option("qp_cond_N", 5)
    with pytest.raises(Exception):
        s.set_option("cond_pred_corr", 1)
    s.set_option("qp_cond_N", 0)
    s.set_option("cond_pred_corr", 1)
    with pytest.raises(Exception):
        s.set_option("qp_cond_N", 5)
    with pytest.raises(Exception):
        s.set_option("cpc_factor", 0.0)
    s.close()
