"""Host-side logic that needs no GPU: the acados_template look-alike, the model registry, the
description validation, and that the C-ABI library loads and exports every symbol of
include/usvmpc.h.  No compute entry point is called here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from mpc_collisionavoidance_amd import _capi, scenario, usv_models
from mpc_collisionavoidance_amd.acados_template import AcadosModel, AcadosOcp, SymVec

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_symbols_are_exported():
    """Every function declared in include/usvmpc.h must be an export of libusvmpc.so."""
    hdr = open(os.path.join(ROOT, "include", "usvmpc.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)  # declarations only, not the interface comments
    declared = set(re.findall(r"\b(usvmpc_[a-z_0-9]+)\s*\(", hdr))
    assert {"usvmpc_create", "usvmpc_solve", "usvmpc_set", "usvmpc_get", "usvmpc_destroy"} <= declared
    path = _capi.lib_path()
    assert os.path.exists(path), "libusvmpc.so missing - run __graft_entry__.build()"
    lib = C.CDLL(path)
    for sym in sorted(declared):
        assert hasattr(lib, sym), sym
    assert declared == set(_capi.EXPORTS)


def test_no_cpu_fallback_create_fails_without_device():
    """Without a HIP device usvmpc_create must fail loudly (E_NODEVICE), never fall back."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    ocp = usv_models.make_ocp("usv_model", 1.0, 20)
    d = _capi.desc_from_ocp(ocp, batch=2)
    h = C.c_void_p()
    rc = _capi.lib().usvmpc_create(C.byref(d), C.byref(h))
    assert rc == -6 and not h.value
    with pytest.raises(RuntimeError):
        usv_models.acados_settings(1.0, 20, name="usv_model")


def test_desc_struct_matches_header_layout():
    nx, nu = C.c_int(), C.c_int()
    lib = _capi.lib()
    for mid, (ex, eu) in _capi.MODEL_DIMS.items():
        assert lib.usvmpc_model_dims(mid, C.byref(nx), C.byref(nu)) == 0 and (nx.value, nu.value) == (ex, eu)
    assert lib.usvmpc_model_dims(7, C.byref(nx), C.byref(nu)) == -1
    d = _capi.Desc()
    lib.usvmpc_default_options(C.byref(d))  # writes the LAST fields of the struct: checks its size/offsets
    # (the QP solver profile BALANCE = HPIPM's mode + acados' overwrites: include/usvmpc.h)
    assert (d.qp_iter_max, d.mu0, d.thr0, d.tol_stat, d.tol_eq, d.alpha_min) == (50, 1.0, 0.1, 1e-6, 1e-8, 1e-8)
    assert (d.hpipm_mode, d.cond_pred_corr, d.cpc_factor) == (_capi.HPIPM_MODES["BALANCE"], 1, 2.0)   # the struct's last fields
    e = _capi.default_options(_capi.Desc())
    fields = ("qp_iter_max", "mu0", "thr0", "tol_stat", "tol_eq", "tol_ineq", "tol_comp", "alpha_min", "hpipm_mode", "cond_pred_corr", "cpc_factor",
              "sim_num_steps", "nlp_max_iter", "nlp_tol_stat")
    assert all(getattr(e, f) == getattr(d, f) for f in fields)
    # every profile: the library's table and the Python mirror agree; R04 is the behaviour up to round 5
    for mode in _capi.HPIPM_MODES.values():
        a, b = _capi.Desc(), _capi.Desc()
        assert lib.usvmpc_hpipm_profile(C.byref(a), mode) == 0
        _capi.hpipm_profile(b, mode)
        assert all(getattr(a, f) == getattr(b, f) for f in fields[:11] if f != "thr0")
    assert lib.usvmpc_hpipm_profile(C.byref(d), 9) == -1
    assert (a.mu0, a.alpha_min, a.cond_pred_corr) == (10.0, 1e-12, 0)   # (R04 is the last mode)
    with pytest.raises(Exception):
        _capi.hpipm_profile(_capi.Desc(), "FAST")


@pytest.mark.parametrize("name,nx,nu,K", [("usv_model", 5, 2, 0), ("usv_model_guidance_ca1", 8, 1, 8), ("usv_model_pf_ca", 14, 2, 4)])
def test_reference_ocp_definitions(name, nx, nu, K):
    """Numbers of the reference's acados_settings.py survive the trip into the C description."""
    ocp = usv_models.make_ocp(name, 1.0, 20)
    assert ocp.model.x.size()[0] == nx and ocp.model.u.size()[0] == nu
    d = _capi.desc_from_ocp(ocp, batch=3)
    assert (d.model, d.N, d.K, d.batch) == (_capi.MODEL_IDS[name], 20, K, 3)
    ny = nx + nu
    W = np.array(d.W[:ny * ny]).reshape(ny, ny)
    assert np.array_equal(W, ocp.cost.W)
    if name == "usv_model_pf_ca":
        Vu = np.array(d.Vu[:ny * nu]).reshape(ny, nu)
        assert Vu[8, 0] == 1.0 and Vu[9, 1] == 1.0 and Vu.sum() == 2.0  # the reference's quirk is kept
        assert W[14, 14] == 0.0 and W[15, 15] == 0.0                      # R = 0
        assert list(d.idxbx[:5]) == [3, 4, 5, 12, 13] and d.soft == 0 and d.uh[0] == 1000000
    if name == "usv_model_guidance_ca1":
        assert d.soft == 1 and d.lsh[0] == -0.2 and d.zl[7] == 1.0 and d.Zl[0] == 0.0 and d.lbu[0] == -0.5
        assert np.array_equal(ocp.parameter_values, 100 * np.ones(16))
    if name == "usv_model":
        assert list(d.ubx[:5]) == [1.5, 1.5, 1.0, 35.0, 35.0] and d.nbu == 2


def test_validation_errors_like_acados():
    ocp = usv_models.make_ocp("usv_model_guidance_ca1", 2.0, 40, 10)
    ocp.cost.W = np.eye(4)
    with pytest.raises(Exception, match="mismatching dimension"):
        _capi.desc_from_ocp(ocp)
    ocp = usv_models.make_ocp("usv_model_pf_ca", 0.4, 40, 10)
    ocp.solver_options.nlp_solver_type = "DDP"
    with pytest.raises(Exception, match="SQP_RTI"):
        _capi.desc_from_ocp(ocp)
    ocp = usv_models.make_ocp("usv_model_pf_ca", 0.4, 40, 10)
    ocp.constraints.lh = np.zeros(3)
    with pytest.raises(Exception, match="lh/uh"):
        _capi.desc_from_ocp(ocp)
    ocp = AcadosOcp()
    m = AcadosModel()
    m.name, m.x, m.u, m.p = "Spatialbycicle_model", SymVec(6), SymVec(2), SymVec(0)
    ocp.model = m
    with pytest.raises(Exception, match="not in the registry"):
        _capi.desc_from_ocp(ocp)


def test_options_pass_through():
    ocp = usv_models.make_ocp("usv_model", 1.0, 20)
    ocp.solver_options.qp_solver_iter_max = 17
    ocp.solver_options.qp_solver_tol_stat = 1e-4
    d = _capi.desc_from_ocp(ocp)
    assert d.qp_iter_max == 17 and d.tol_stat == 1e-4 and d.tol_eq == 1e-8


def test_scenario_shapes_and_determinism():
    for name, K in (("usv_model", 0), ("usv_model_guidance_ca1", 10), ("usv_model_pf_ca", 10)):
        a = scenario.make_batch(name, 12, K, 7, seed=3)
        b = scenario.make_batch(name, 12, K, 7, seed=3)
        nx, nu = a["nx"], a["nu"]
        assert a["x0"].shape == (7, nx) and a["yref"].shape == (7, 12, nx + nu) and a["p"].shape == (7, 13, 2 * K)
        assert a["lh"].shape == (7, 12, K) and a["x_init"].shape == (7, 13, nx) and a["u_init"].shape == (7, 12, nu)
        assert all(np.array_equal(a[k], b[k]) for k in ("x0", "p", "lh", "x_init"))
        assert np.array_equal(a["x_init"][:, 0], a["x0"])
        if K:
            ipx, ipy = (5, 6) if nx == 8 else (10, 11)
            d = np.hypot(a["x0"][:, None, ipx] - a["p"][:, 0, 0::2], a["x0"][:, None, ipy] - a["p"][:, 0, 1::2])
            assert (d > a["lh"][:, 0]).all()  # every vessel starts outside every keep-out circle
    m = scenario.make_batch("usv_model_pf_ca", 8, 4, 3, moving=True)
    assert not np.array_equal(m["p"][:, 0], m["p"][:, 8])


def test_reference_of_the_wrong_length_is_refused():
    """acados raises a dimension error for a yref / yref_e that does not match ny / ny_e; so does the constructor
    (before any device is touched)."""
    from mpc_collisionavoidance_amd import BatchOcpSolver
    ocp = usv_models.make_ocp("usv_model_pf_ca", 1.0, 10, 4)
    ocp.cost.yref = np.zeros(5)
    with pytest.raises(Exception, match="inconsistent dimension"):
        BatchOcpSolver(ocp, 2)
    ocp = usv_models.make_ocp("usv_model_pf_ca", 1.0, 10, 4)
    ocp.cost.yref_e = np.zeros(3)
    with pytest.raises(Exception, match="inconsistent dimension"):
        BatchOcpSolver(ocp, 2)


def test_bounds_beyond_1e100_are_refused():
    """The IPM multiplies the two slacks (and the two multipliers) of a row: a bound that stands for "none" has to stay inside
    the double range next to them (host_spec.hpp).  Refused at create, before any device is touched."""
    ocp = usv_models.make_ocp("usv_model_pf_ca", 0.4, 40, 10)
    ocp.constraints.uh = 1e300 * np.ones(10)
    d = _capi.desc_from_ocp(ocp, batch=2)
    h = C.c_void_p()
    assert _capi.lib().usvmpc_create(C.byref(d), C.byref(h)) == -1 and not h.value
    ocp = usv_models.make_ocp("usv_model", 1.0, 20)
    ocp.constraints.ubu = np.array([np.inf, 1.0])
    d = _capi.desc_from_ocp(ocp, batch=2)
    assert _capi.lib().usvmpc_create(C.byref(d), C.byref(h)) == -1 and not h.value
