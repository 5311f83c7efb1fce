"""The tail of the device-vs-oracle differences under the DEFAULT QP solver profile (BALANCE: HPIPM's mode + acados' overwrites as recalled -
include/usvmpc.h USVMPC_HPIPM_*; the oracle with the mode's two rounds of iterative refinement), as a committed fixture
(tests/golden/parity_tail_balance.npz, made on an MI355X by tools/outlier_fixture.py: BASELINE configs[2] closed loop, 2048 instances x 10
ticks, every solve compared from identical inputs).

Under the profile of rounds 4 / 5 ("R04": tests/golden/parity_outliers_pf_ca.npz, tests/test_parity_outliers.py) that run had two or three
instances above north_star's 1e-5 (up to 5.3e-4), the largest traced to the oracle's unrefined square-root Riccati solve.  Under the default
profile it has ONE of 20 450 compared solves (3.4e-5, same iteration count on both sides; the six next are 3.5e-6 ... 6.6e-6): the fixture holds
the run's statistics, that instance ("outlier"), the largest differences below it ("near") and a few ordinary instances.
CPU suite: the oracle replays its stored outputs, the kernel bodies on the lane emulator take the device's iteration counts.
GPU suite: the device reproduces its stored outputs bit for bit."""
import os

import numpy as np
import pytest

from mpc_collisionavoidance_amd import _capi, scenario, usv_models
from tests.test_emu_kernels import emu_rti

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIX = os.path.join(ROOT, "tests", "golden", "parity_tail_balance.npz")


def _load():
    f = np.load(FIX)
    d = {k: f[k] for k in f.files}
    d["N"], d["K"], d["steps"], d["dt"] = int(f["N"]), int(f["K"]), int(f["sim_steps"]), float(f["dt"])
    return d


def _err(f, xa, ua, xb, ub):
    n = xa.shape[0]
    sx = np.maximum(1e-2, np.abs(f["x_orc"]).max(axis=(0, 1)))
    su = np.maximum(1e-2, np.abs(f["u_orc"]).max(axis=(0, 1)))
    return np.maximum((np.abs(xa - xb) / sx).reshape(n, -1).max(axis=1), (np.abs(ua - ub) / su).reshape(n, -1).max(axis=1))


def _ocp(f):
    ocp = usv_models.make_ocp("usv_model_pf_ca", f["N"] * f["dt"], f["N"], f["K"])
    ocp.solver_options.sim_method_num_steps = f["steps"]
    ocp.solver_options.hpipm_mode = str(f["profile"])
    return ocp


def test_the_tail_of_the_run_is_what_the_documents_say():
    f = _load()
    assert str(f["profile"]) == "BALANCE"
    assert int(f["n_compared"]) >= 20000                                   # 2048 instances x 10 ticks, less the unconverged
    # north_star asks <= 1e-5 relative trajectory error: met by all but ONE solve of the run (R04: three, up to 5.3e-4)
    assert int(f["n_above"]) == 1 and float(f["err_max"]) <= 5e-5
    assert float(f["err_p99"]) <= 1e-6 and float(f["err_p50"]) <= 1e-10
    kinds = list(f["kind"])
    assert kinds.count("outlier") == int(f["n_above"]) and kinds.count("near") >= 4 and kinds.count("ordinary") >= 2
    e = _err(f, f["x_dev"], f["u_dev"], f["x_orc"], f["u_orc"])
    out = f["kind"] == "outlier"
    assert (e[out] > 1e-5).all() and (e[~out] <= 1e-5).all() and (e[f["kind"] == "ordinary"] <= 1e-10).all()
    assert np.array_equal(f["it_dev"], f["it_orc"])                         # the same iteration count on every instance of the set


def test_oracle_replays_its_stored_outputs(oracle):
    f = _load()
    spec = oracle.spec(2, f["N"], f["N"] * f["dt"], f["K"], sim_steps=f["steps"], hpipm_mode=str(f["profile"]))
    x, u = f["x_in"].copy(), f["u_in"].copy()
    st, it = oracle.rti_batch(spec, x, u, *[np.ascontiguousarray(f[k]) for k in ("x0", "yref", "yref_e", "p", "lh")], threads=0)
    # (the stored oracle outputs were computed on the GPU box's host: same sources, possibly other compiler flags)
    assert (st == 0).all() and np.array_equal(it, f["it_orc"])
    assert _err(f, x, u, f["x_orc"], f["u_orc"]).max() <= 1e-9


def test_lane_emulator_follows_the_device(emu):
    f = _load()
    n = f["x_in"].shape[0]
    desc = _capi.desc_from_ocp(_ocp(f), batch=n)
    wl = {k: np.ascontiguousarray(f[k]) for k in ("x0", "yref", "yref_e", "p", "lh")}
    r = emu_rti(emu, desc, wl, f["x_in"], f["u_in"])
    assert (r["qp_status"] == 0).all()
    assert np.abs(r["qp_iter"] - f["it_dev"]).max() <= 1
    same = r["qp_iter"] == f["it_dev"]
    e = _err(f, r["x"], r["u"], f["x_dev"], f["u_dev"])
    # the same source in two builds (IEEE division and libm on the CPU, v_rcp + Newton and fused multiply-adds on the device): rounding-level
    # differences, amplified on the sensitive instances by this model's control weight R = 0 (as between device and oracle: tests/test_parity_outliers.py)
    assert same.mean() >= 0.8 and e[same].max() <= 1e-4, (e, r["qp_iter"], f["it_dev"])
    assert e[(f["kind"] == "ordinary") & same].max() <= 1e-10


@pytest.mark.gpu
def test_device_reproduces_its_stored_outputs():
    from mpc_collisionavoidance_amd import BatchOcpSolver
    f = _load()
    n = f["x_in"].shape[0]
    s = BatchOcpSolver(_ocp(f), n)
    wl = dict(x_init=f["x_in"], u_init=f["u_in"], K=f["K"], **{k: np.ascontiguousarray(f[k]) for k in ("x0", "yref", "yref_e", "p", "lh")})
    scenario.load_into(s, wl)
    for wide in (0, 1):   # (the stored outputs are the throughput mapping's; the latency mapping returns the same bits)
        s.set_option("wide", wide)
        s.set_all("x", f["x_in"]); s.set_all("u", f["u_in"])
        st = s.solve()
        assert (st == 0).all() and np.array_equal(s.get_int("qp_iter"), f["it_dev"])
        assert np.array_equal(s.get_all("x"), f["x_dev"]) and np.array_equal(s.get_all("u"), f["u_dev"]), wide
    s.close()


def test_the_sensitive_instances_have_no_solution_to_compare_at_1e_5(oracle):
    """What the one instance above 1e-5 and its neighbours are: QPs on which the oracle asked for 1e-11 does not converge at all (step length at
    its floor at a degenerate vertex of the hard rows, control weight R = 0: status 4), and on which the oracle's OWN two Riccati forms - same
    iteration counts - differ by up to 1e-2.  Device and oracle take the same number of iterations on every one of them; each returns a point
    inside the tolerance ball HPIPM's exit test defines, 3e-6 ... 3e-5 apart."""
    f = _load()
    args = [np.ascontiguousarray(f[k]) for k in ("x0", "yref", "yref_e", "p", "lh")]
    sens = f["kind"] != "ordinary"

    def run(**o):
        spec = oracle.spec(2, f["N"], f["N"] * f["dt"], f["K"], sim_steps=f["steps"], hpipm_mode=str(f["profile"]), **o)
        x, u = f["x_in"].copy(), f["u_in"].copy()
        st, it = oracle.rti_batch(spec, x, u, *args, threads=0)
        return x, u, st, it

    xt, ut, stt, itt = run(tol_stat=1e-11, tol_eq=1e-11, tol_ineq=1e-11, tol_comp=1e-11, qp_iter_max=200)
    assert (stt[sens] == 4).all()
    xs, us, sts, its = run()
    xc, uc, stc, itc = run(riccati=oracle.RICCATI_CLASSIC)
    assert (stc == 0).all() and np.array_equal(itc, its)
    forms = _err(f, xc, uc, xs, us)
    dev = _err(f, f["x_dev"], f["u_dev"], xs, us)
    assert forms[sens].max() >= 1e-3 and forms[sens].max() >= 100 * dev[sens].max()   # the oracle's two forms differ far more than device and oracle do
    assert forms[~sens].max() <= 1e-10
