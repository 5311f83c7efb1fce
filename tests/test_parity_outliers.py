"""The classified device-vs-oracle parity outliers of the bench workload, as a committed fixture (tests/golden/parity_outliers_pf_ca.npz,
made on an MI355X by tools/outlier_fixture.py: BASELINE configs[2] closed loop, 2048 instances x 10 ticks; every instance above
north_star's 1e-5, the largest ones below it, and a few ordinary instances - inputs of the solve and BOTH sides' outputs).

What they show (CPU suite):
* the fixture replays: the oracle (square-root Riccati, its default) reproduces its stored outputs;
* the gap is the factorisation FORM, not the hardware: the oracle's own classical-Riccati mode (USV_RICCATI_CLASSIC - the form
  the kernels use) lands on the device's point for the largest outlier (same iteration count, <= 1e-5), and over the set the
  oracle's two forms differ from EACH OTHER by more than the device differs from the oracle;
* the kernel bodies on the lane emulator (tests/emu: the same C++ as the device, IEEE division, no fused-multiply-add contraction)
  take the device's iteration counts on every instance, sit 1000x closer to the device than the oracle on the largest outlier -
  and differ from the device by 1e-6 .. 5e-5 on every other outlier / near instance (1e-13 on ordinary ones): these QPs amplify
  rounding by ten orders of magnitude whoever solves them, so bit-for-bit equality of two builds of one source does not exist here.
GPU suite: the device run again on the fixture's inputs against its own emulator and its stored outputs.

The fixture was taken under the QP solver profile "R04" (the defaults up to round 5: no conditional predictor-corrector, mu0 = 10, no iterative
refinement in the oracle - include/usvmpc.h USVMPC_HPIPM_R04) and every side below runs that profile: it is the record of what the unpinned
profile choice was worth, and it keeps R04 reachable bit for bit.  Under the default profile since round 6 (BALANCE: acados' overwrites,
cond_pred_corr, the oracle with HPIPM's two rounds of iterative refinement) the same closed loop has ONE instance of 20 448 above 1e-5, at
3.4e-5 (tests/golden/parity_tail_balance.npz, tests/test_parity_tail_balance.py).
"""
import os

import numpy as np
import pytest

from mpc_collisionavoidance_amd import _capi, scenario, usv_models
from tests.test_emu_kernels import emu_rti

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIX = os.path.join(ROOT, "tests", "golden", "parity_outliers_pf_ca.npz")


def _load():
    f = np.load(FIX)
    d = {k: f[k] for k in f.files}
    d["N"], d["K"], d["steps"], d["dt"] = int(f["N"]), int(f["K"]), int(f["sim_steps"]), float(f["dt"])
    return d


def _err(f, xa, ua, xb, ub):
    """tests/util.rel_err_per_instance with the component scales of the set's stored oracle outputs"""
    n = xa.shape[0]
    sx = np.maximum(1e-2, np.abs(f["x_orc"]).max(axis=(0, 1)))
    su = np.maximum(1e-2, np.abs(f["u_orc"]).max(axis=(0, 1)))
    return np.maximum((np.abs(xa - xb) / sx).reshape(n, -1).max(axis=1), (np.abs(ua - ub) / su).reshape(n, -1).max(axis=1))


def _oracle(ob, f, **opts):
    spec = ob.spec(2, f["N"], f["N"] * f["dt"], f["K"], sim_steps=f["steps"], hpipm_mode="R04", **opts)
    x, u = f["x_in"].copy(), f["u_in"].copy()
    st, it = ob.rti_batch(spec, x, u, *[np.ascontiguousarray(f[k]) for k in ("x0", "yref", "yref_e", "p", "lh")], threads=0)
    return x, u, st, it


def _ocp(f):
    ocp = usv_models.make_ocp("usv_model_pf_ca", f["N"] * f["dt"], f["N"], f["K"])
    ocp.solver_options.sim_method_num_steps = f["steps"]
    ocp.solver_options.hpipm_mode = "R04"
    return ocp


def _emu(emu, f):
    n = f["x_in"].shape[0]
    desc = _capi.desc_from_ocp(_ocp(f), batch=n)
    wl = {k: np.ascontiguousarray(f[k]) for k in ("x0", "yref", "yref_e", "p", "lh")}
    return emu_rti(emu, desc, wl, f["x_in"], f["u_in"])


def test_fixture_is_what_it_says():
    f = _load()
    kinds = list(f["kind"])
    assert kinds.count("outlier") >= 1 and kinds.count("near") >= 4 and kinds.count("ordinary") >= 2
    e = _err(f, f["x_dev"], f["u_dev"], f["x_orc"], f["u_orc"])
    out = f["kind"] == "outlier"
    assert (e[out] > 1e-5).all() and (e[~out] <= 1e-5).all()
    assert e.max() <= 5e-3                              # (the cap of the documented rule, tests/parity_rule.py)
    assert (e[f["kind"] == "ordinary"] <= 1e-11).all()  # what an ordinary instance looks like


def test_oracle_replays_and_the_gap_is_the_factorisation_form(oracle):
    f = _load()
    xs, us, sts, its = _oracle(oracle, f)
    # the stored oracle outputs were computed on the GPU box's host: same sources, possibly other compiler flags
    assert (sts == 0).all() and np.array_equal(its, f["it_orc"])
    assert _err(f, xs, us, f["x_orc"], f["u_orc"]).max() <= 1e-9
    xc, uc, stc, itc = _oracle(oracle, f, riccati=oracle.RICCATI_CLASSIC)
    # (the classical form in plain C is the less robust of the two: on one of the outlier QPs it stops at the step-length floor -
    # status 4, iterate untouched - where the square-root form and the device both converge; only outliers may do that)
    assert (stc[f["kind"] != "outlier"] == 0).all()
    dev_sqrt = _err(f, f["x_dev"], f["u_dev"], xs, us)
    dev_cls = _err(f, f["x_dev"], f["u_dev"], xc, uc)
    cls_sqrt = _err(f, xc, uc, xs, us)
    print("device vs oracle (sqrt)   ", dev_sqrt)
    print("device vs oracle (classic)", dev_cls, itc, f["it_dev"])
    print("oracle classic vs sqrt    ", cls_sqrt)
    worst = int(np.argmax(dev_sqrt))
    assert f["kind"][worst] == "outlier"
    # the largest outlier: the oracle run in the kernels' own Riccati form takes the device's iteration count and lands on the
    # device's point - three orders of magnitude closer than its square-root form does
    assert itc[worst] == f["it_dev"][worst] != its[worst]
    assert dev_cls[worst] <= 1e-5 and dev_cls[worst] <= 1e-2 * dev_sqrt[worst]
    # and over the set the oracle's two forms differ from each other by MORE than the device differs from the oracle: at this
    # model's conditioning (control weight R = 0) an iterate that passes the exit test is not determined to 1e-5
    both = stc == 0
    assert cls_sqrt[both].max() >= 0.99 * dev_sqrt.max()   # (on the largest outlier the classical form IS the device's point: the two gaps coincide)
    assert (cls_sqrt > 1e-5).sum() >= (dev_sqrt > 1e-5).sum()
    # ordinary instances: all three agree to rounding
    o = f["kind"] == "ordinary"
    assert dev_sqrt[o].max() <= 1e-11 and dev_cls[o].max() <= 1e-11 and cls_sqrt[o].max() <= 1e-11


def test_hpipm_options_not_adopted_by_default_on_the_outliers(oracle):
    """The two HPIPM options of acados' modes that the restatement leaves off (usv_opts): what they do to THIS set.
    * cond_pred_corr (on in HPIPM's SPEED / BALANCE / ROBUST): the fallback to the centring-only step never fires on these QPs - the
      oracle's outputs do not change by a bit;
    * itref_corr_max = 2 (BALANCE): iterative refinement of the corrector's KKT solve fires on ONE instance, the largest outlier - and
      moves the oracle's square-root form onto the device's point (same iteration count as the device, <= 1e-6 instead of 5e-4).  The
      device (classical Riccati, no refinement), the oracle's classical form and the refined square-root form agree there; the plain
      square-root form is the odd one out: that outlier is the oracle's solve accuracy, not the device's.  Every other instance is
      untouched (their linear-system residuals are below the exit tolerances, as HPIPM tests before refining)."""
    f = _load()
    xs, us, sts, its = _oracle(oracle, f)
    xc, uc, stc, itc = _oracle(oracle, f, cond_pred_corr=1)
    assert np.array_equal(xs, xc) and np.array_equal(us, uc) and np.array_equal(its, itc)
    xr, ur, str_, itr = _oracle(oracle, f, itref_corr_max=2)
    assert (str_ == 0).all()
    dev_plain = _err(f, f["x_dev"], f["u_dev"], xs, us)
    dev_ref = _err(f, f["x_dev"], f["u_dev"], xr, ur)
    moved = _err(f, xr, ur, xs, us)
    print("device vs oracle, refined", dev_ref, "\noracle refined vs plain ", moved)
    worst = int(np.argmax(dev_plain))
    assert f["kind"][worst] == "outlier" and moved[worst] > 1e-4
    assert itr[worst] == f["it_dev"][worst] != its[worst]
    assert dev_ref[worst] <= 1e-6 and dev_ref[worst] <= 1e-3 * dev_plain[worst]
    others = np.arange(len(moved)) != worst
    assert (moved[others] == 0.0).all() and np.array_equal(itr[others], its[others])
    # both together: the same as refinement alone
    xb, ub, stb, itb = _oracle(oracle, f, cond_pred_corr=1, itref_corr_max=2)
    assert np.array_equal(xb, xr) and np.array_equal(ub, ur)


def test_lane_emulator_reproduces_the_device_on_the_outliers(emu):
    """Nothing hardware in the gap: the kernels' C++ executed on the CPU (IEEE division and square root instead of the
    v_rcp / v_rsq + Newton forms, libm instead of the device's sincos, no FMA contraction) gives the device's outputs."""
    f = _load()
    r = _emu(emu, f)
    e = _err(f, r["x"], r["u"], f["x_dev"], f["u_dev"])
    eo = _err(f, f["x_dev"], f["u_dev"], f["x_orc"], f["u_orc"])
    print("emulator vs device", e, r["qp_iter"], f["it_dev"])
    assert (r["qp_status"] == 0).all()
    assert np.array_equal(r["qp_iter"], f["it_dev"])
    # the largest outlier (a different factorisation form on the oracle's side): the emulator sits 1000x closer to the device
    worst = int(np.argmax(eo))
    assert e[worst] <= 1e-2 * eo[worst], (e, eo)
    # every other instance of the set is rounding-sensitive in itself: the SAME source differs from the device by 1e-6 .. 5e-5
    # there (iteration counts equal) - the size of the device-vs-oracle differences on them - and by 1e-13 on ordinary instances:
    # what amplifies is the QP (control weight R = 0, degenerate hard-row vertices), not an implementation
    sens = f["kind"] != "ordinary"
    assert e[sens].max() <= 1e-4 and e[~sens].max() <= 1e-11, e
    assert np.median(e[sens]) >= 1e-7


@pytest.mark.gpu
def test_device_reproduces_its_stored_outputs_and_its_emulator(emu):
    from mpc_collisionavoidance_amd import BatchOcpSolver
    f = _load()
    n = f["x_in"].shape[0]
    s = BatchOcpSolver(_ocp(f), n)
    wl = dict(x_init=f["x_in"], u_init=f["u_in"], K=f["K"], **{k: np.ascontiguousarray(f[k]) for k in ("x0", "yref", "yref_e", "p", "lh")})
    scenario.load_into(s, wl)
    s.set_option("wide", 0)   # (the stored outputs are the throughput mapping's: a handle of twelve would take the latency mapping)
    st = s.solve()
    xg, ug, qi = s.get_all("x"), s.get_all("u"), s.get_int("qp_iter")
    s.close()
    assert (st == 0).all() and np.array_equal(qi, f["it_dev"])
    # the stored outputs came from a 2048-instance handle (static_obstacles on, sorted queue): scheduling and storage options do
    # not change a bit, and neither does the batch an instance sits in
    assert np.array_equal(xg, f["x_dev"]) and np.array_equal(ug, f["u_dev"])
    r = _emu(emu, f)
    e = _err(f, r["x"], r["u"], xg, ug)
    sens = f["kind"] != "ordinary"
    assert np.array_equal(r["qp_iter"], qi) and e[sens].max() <= 1e-4 and e[~sens].max() <= 1e-11, e
