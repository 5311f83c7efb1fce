"""The symbolic route: casadi_lite (the slice of `from casadi import *` the reference's model files use),
codegen (forward-mode device code + structure traits) and the generated-library path.

CPU tests check the shim, the generator against sympy differentiation, the generated kernels on the lane
emulator against the hand-written models and the oracle, and a model that is NOT in the registry against the
oracle's generated-model hook.  When the reference tree is present (this container only, never the GPU box) the
reference's own 12 `usv_model.py` files are run unchanged through the shim.  GPU tests compile and run the
generated libraries on the device.
"""
import ctypes as C
import importlib.util
import os
import sys

import numpy as np
import pytest
import sympy as sp

from mpc_collisionavoidance_amd import _capi, casadi_lite as ca, codegen, genbuild, scenario, usv_models
from mpc_collisionavoidance_amd.acados_template import AcadosModel, AcadosOcp
from tests import util
from tests.test_emu_kernels import emu_rti

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU = os.path.join(ROOT, "tests", "emu")
REF = "/root/reference/catkin_ws/src/nmpc_ca/scripts"
WORK = os.path.join(ROOT, "mpc_collisionavoidance_amd", "csrc", "gen")


# ------------------------------------------------------------------ shim behaviour
def test_shim_api_surface():
    x = ca.MX.sym("x")
    y = ca.MX.sym("x")                       # same name, distinct symbol (usv_pf_ca/usv_model.py:123-131)
    assert x is not y and x.size() == (1, 1)
    v = ca.vertcat(x, y, 3.0)
    assert v.size() == (3, 1) and v.size()[0] == 3 and ca.vertcat([]).size() == (0, 1)
    assert (x + 0) is x and (1 * x) is x and (x * 0).is_constant() and (x - 0) is x
    assert (x * y) is (x * y)                # structural interning
    e = np.float64(2.0) * x + np.pi          # numpy scalars defer to the symbolic operators
    assert ca.evaluate([e], {x: 1.5})[0] == 2.0 * 1.5 + np.pi
    assert ca.evaluate([ca.if_else(x > 1.25, 64.55, -25)], {x: 1.3})[0] == 64.55
    assert ca.evaluate([ca.if_else(x > 1.25, 64.55, -25)], {x: 1.2})[0] == -25
    assert ca.evaluate([ca.atan2(y, x + 0.001), ca.fabs(-x), ca.sqrt(x * x)], {x: 2.0, y: 1.0}) == \
        [np.arctan2(1.0, 2.001), 2.0, 2.0]
    d = ca.vertcat(x, y) - ca.vertcat(y, x)   # f_impl = xdot - f_expl
    assert d.size() == (2, 1)
    with pytest.raises(TypeError):
        bool(x > 1)
    assert ca.np.math.atan2(1.0, 1.0) == np.arctan2(1.0, 1.0) and ca.pi == np.pi   # names the star import provides
    assert hasattr(ca.types, "SimpleNamespace") and ca.np.array([1.0]).shape == (1,)


def _to_sympy(nodes, symmap):
    memo = {}
    fn = {"sin": sp.sin, "cos": sp.cos, "tan": sp.tan, "sqrt": sp.sqrt, "exp": sp.exp, "log": sp.log, "tanh": sp.tanh}
    for n in ca.topo(nodes):
        a = [memo[c.key] for c in n.args]
        k = n.kind
        if k == "sym": r = symmap[n]
        elif k == "const": r = sp.Float(n.value)
        elif k == "add": r = a[0] + a[1]
        elif k == "sub": r = a[0] - a[1]
        elif k == "mul": r = a[0] * a[1]
        elif k == "div": r = a[0] / a[1]
        elif k == "neg": r = -a[0]
        elif k == "pow": r = a[0] ** a[1]
        elif k in fn: r = fn[k](a[0])
        elif k == "fabs": r = sp.Abs(a[0])
        elif k == "atan2": r = sp.atan2(a[0], a[1])
        elif k == "gt": r = sp.Gt(a[0], a[1])
        elif k == "lt": r = sp.Lt(a[0], a[1])
        elif k == "if_else": r = sp.Piecewise((a[1], a[0]), (a[2], True))
        else: raise NotImplementedError(k)
        memo[n.key] = r
    return [memo[n.key] for n in nodes]


def _check_against_sympy(info, tmp_path, tag):
    """Generated tangent code (compiled as plain C) vs sympy differentiation of the same graph."""
    import subprocess
    src, so = tmp_path / (tag + ".c"), tmp_path / (tag + ".so")
    src.write_text(codegen.emit_oracle_c(info))
    subprocess.check_call(["gcc", "-O1", "-shared", "-fPIC", "-o", str(so), str(src), "-lm"])
    L = C.CDLL(str(so))
    dp = C.POINTER(C.c_double)
    nx, nu = info.nx, info.nu
    xs, us = sp.symbols("x0:%d" % nx, real=True), sp.symbols("u0:%d" % nu, real=True)
    fs = sp.Matrix(_to_sympy(info.f, {**dict(zip(info.x, xs)), **dict(zip(info.u, us))}))
    F = sp.lambdify((xs, us), fs, "numpy")
    J = sp.lambdify((xs, us), fs.jacobian(sp.Matrix(list(us) + list(xs))), "numpy")
    rng = np.random.default_rng(1)
    for t in range(6):
        x, u = rng.normal(size=nx), rng.normal(size=nu)
        if t % 2 == 0 and nx >= 5:
            x[3 if nx > 8 else 0] = 1.4  # the u > 1.25 branch of the 3-DOF block where it exists
        Jg, f, js = np.zeros((nx, nx + nu)), np.zeros(nx), np.zeros(nx)
        for c in range(nx + nu):
            s, su = np.zeros(nx), np.zeros(max(nu, 1))
            if c < nu: su[c] = 1
            else: s[c - nu] = 1
            L.usv_gen_fjvp(x.ctypes.data_as(dp), u.ctypes.data_as(dp), s.ctypes.data_as(dp), su.ctypes.data_as(dp),
                           f.ctypes.data_as(dp), js.ctypes.data_as(dp))
            Jg[:, c] = js
        Jr = np.asarray(J(x, u), dtype=float)
        assert np.allclose(f, np.asarray(F(x, u), dtype=float).ravel(), rtol=1e-12, atol=1e-12)
        assert np.abs(Jg - Jr).max() <= 1e-9 * max(1.0, np.abs(Jr).max())


@pytest.mark.parametrize("name,K", [("usv_model", 0), ("usv_model_guidance_ca1", 10), ("usv_model_pf_ca", 10)])
def test_generated_traits_and_derivatives_of_the_registry_models(tmp_path, name, K):
    ocp = usv_models.make_ocp(name, 1.0, 20, K or None, symbolic=True)
    info = codegen.analyse(ocp.model)
    want = {"usv_model": (5, 2, 0, 0, 0, 0, 0),
            "usv_model_guidance_ca1": (8, 1, 10, 5, 6, 0b11, (1 << 3) | (1 << 6) | (1 << 7)),
            "usv_model_pf_ca": (14, 2, 10, 10, 11, 0b1110000000, 0b11011100011000)}[name]   # = csrc/models.hpp
    assert (info.nx, info.nu, info.K, info.ipx, info.ipy, info.out_unit, info.in_unit) == want
    from tests.test_oracle_models import SENS   # = csrc/models.hpp SENS / DIAG_ONE
    mid = {"usv_model": 0, "usv_model_guidance_ca1": 1, "usv_model_pf_ca": 2}[name]
    assert (info.sens, info.diag_one) == (SENS[mid][0], SENS[mid][1])
    _check_against_sympy(info, tmp_path, name)


VARIANTS = ["usv_acados", "usv_guidance", "usv_guidance2", "usv_guidance3", "usv_guidance4", "usv_guidance5",
            "usv_guidance_ca", "usv_guidance_ca1", "usv_low_level", "usv_pf", "usv_pf_ca", "usv_position_control"]


def _load_reference_model(variant):
    ca.install()
    spec = importlib.util.spec_from_file_location("refmodel_" + variant, os.path.join(REF, variant, "usv_model.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    model, constraint = mod.usv_model()
    m = AcadosModel()
    m.x, m.u, m.p, m.f_expl_expr, m.name = model.x, model.U, model.p, model.f_expl_expr, model.name
    ce = getattr(constraint, "expr", None)
    m.con_h_expr = ce if isinstance(ce, ca.MXVec) else None
    return model, m


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not present (it never travels to the GPU box)")
@pytest.mark.parametrize("variant", VARIANTS)
def test_reference_model_files_run_unchanged_through_the_shim(tmp_path, variant):
    """Every USV model file of the reference executes with `from casadi import *` resolved to casadi_lite, and
    the code generated from it matches sympy differentiation of its own expression graph."""
    model, m = _load_reference_model(variant)
    info = codegen.analyse(m)
    assert info.nx == model.x.size()[0] and info.nu == model.U.size()[0] and info.nx + info.nu <= 16
    _check_against_sympy(info, tmp_path, variant)
    if variant in ("usv_acados", "usv_guidance_ca1", "usv_pf_ca"):   # and equals the oracle's hand restatement
        from oracle import binding as ob
        mid = {"usv_acados": 0, "usv_guidance_ca1": 1, "usv_pf_ca": 2}[variant]
        rng = np.random.default_rng(0)
        for _ in range(5):
            x, u = rng.normal(size=info.nx), rng.normal(size=info.nu)
            vals = {**dict(zip(info.x, x)), **dict(zip(info.u, u))}
            assert np.allclose(ca.evaluate(info.f, vals), ob.model_f(mid, x, u), rtol=1e-13, atol=1e-13)


def test_unsupported_constructs_are_refused():
    x, u, p = ca.MX.sym("x", 3), ca.MX.sym("u", 1), ca.MX.sym("p", 2)
    m = AcadosModel()
    m.name, m.x, m.u, m.p = "t", x, u, p
    m.f_expl_expr = ca.vertcat(x[1], u[0] * p[0], 0)          # dynamics depending on p
    with pytest.raises(Exception, match="parameter"):
        codegen.analyse(m)
    m.f_expl_expr = ca.vertcat(x[1], u[0], 0)
    m.con_h_expr = ca.vertcat(x[0] * x[1] - p[0] - p[1])       # not a circular-obstacle row
    with pytest.raises(NotImplementedError):
        codegen.analyse(m)
    big = AcadosModel()
    big.name, big.x, big.u, big.p = "big", ca.MX.sym("x", 15), ca.MX.sym("u", 2), ca.vertcat([])
    big.f_expl_expr = ca.vertcat(*[big.x[i] for i in range(15)])
    with pytest.raises(Exception, match="exceed"):
        codegen.analyse(big)


# ------------------------------------------------------------------ generated kernels on the lane emulator
def _emu_lib(path):
    lib = C.CDLL(path)
    lib.usv_emu_solve.argtypes = [C.POINTER(_capi.Desc)] + [_capi._dp] * 10 + [_capi._ip] * 3 + [_capi._dp] * 4
    return lib


@pytest.mark.parametrize("name,K", [("usv_model_guidance_ca1", 4), ("usv_model_pf_ca", 3)])
def test_generated_kernels_equal_hand_written_ones_on_the_emulator(oracle, emu, name, K):
    N, B = 6, 3
    ocp_s = usv_models.make_ocp(name, N * scenario.DT[name], N, K, symbolic=True)
    wl = scenario.make_batch(name, N, K, B, seed=5)
    info = codegen.analyse(ocp_s.model)
    soft = name == "usv_model_guidance_ca1"
    gen = _emu_lib(genbuild.build_emu_lib(info, (K + 15) // 16, soft, EMU))
    desc_g = _capi.desc_from_ocp(ocp_s, batch=B, generated=True)
    rg = emu_rti(gen, desc_g, wl, wl["x_init"], wl["u_init"])
    spec = util.oracle_spec(oracle, name, N, scenario.DT[name], K)
    xo, uo, sto, ito = util.oracle_rti(oracle, spec, wl, wl["x_init"], wl["u_init"])
    assert np.array_equal(rg["status"], sto) and np.abs(rg["qp_iter"] - ito).max() <= 1
    assert util.rel_err(rg["x"], xo) < 1e-8 and util.rel_err(rg["u"], uo) < 1e-8
    rs = emu_rti(emu, _capi.desc_from_ocp(usv_models.make_ocp(name, N * scenario.DT[name], N, K), batch=B),
                 wl, wl["x_init"], wl["u_init"])
    assert util.rel_err(rg["x"], rs["x"]) < 1e-11 and util.rel_err(rg["u"], rs["u"]) < 1e-11


def _kinematic_ocp(N, Tf):
    """A model that is NOT in the registry: planar kinematic vessel with first-order speed / yaw-rate lags,
    x = (px, py, psi, v, om), u = (v_cmd, om_cmd); LS cost on position error, speed and controls; input boxes."""
    px, py, psi, v, om = (ca.MX.sym(n) for n in ("px", "py", "psi", "v", "om"))
    vc, oc = ca.MX.sym("v_cmd"), ca.MX.sym("om_cmd")
    ocp = AcadosOcp()
    m = AcadosModel()
    m.name = "kinematic_vessel"
    m.x, m.u, m.p = ca.vertcat(px, py, psi, v, om), ca.vertcat(vc, oc), ca.vertcat([])
    m.f_expl_expr = ca.vertcat(v * ca.cos(psi), v * ca.sin(psi), om, (vc - v) / 0.8, (oc - om) / 0.4 - 0.3 * om * ca.fabs(om))
    ocp.model = m
    nx, nu = 5, 2
    ny = nx + nu
    ocp.dims.N = N
    ocp.cost.W = np.diag([1.0, 1.0, 0.1, 0.5, 0.05, 0.02, 0.02])
    ocp.cost.W_e = np.diag([5.0, 5.0, 0.5, 0.5, 0.05])
    ocp.cost.Vx = np.vstack([np.eye(nx), np.zeros((nu, nx))])
    Vu = np.zeros((ny, nu)); Vu[5, 0] = 1.0; Vu[6, 1] = 1.0
    ocp.cost.Vu, ocp.cost.Vx_e = Vu, np.eye(nx)
    ocp.cost.yref, ocp.cost.yref_e = np.zeros(ny), np.zeros(nx)
    ocp.constraints.lbu, ocp.constraints.ubu, ocp.constraints.idxbu = np.array([-0.2, -0.8]), np.array([1.5, 0.8]), np.array([0, 1])
    ocp.constraints.lbx, ocp.constraints.ubx, ocp.constraints.idxbx = np.array([-0.5]), np.array([1.2]), np.array([3])
    ocp.constraints.x0 = np.zeros(nx)
    ocp.solver_options.tf = Tf
    return ocp


def _kinematic_workload(N, B, seed=0):
    rng = np.random.default_rng(seed)
    x0 = np.column_stack([rng.uniform(-2, 2, B), rng.uniform(-2, 2, B), rng.uniform(-1, 1, B), rng.uniform(0.2, 1.0, B), rng.uniform(-0.2, 0.2, B)])
    yr = np.zeros(7); yr[3] = 0.8
    return dict(x0=x0, yref=np.tile(yr, (B, N, 1)), yref_e=np.tile(yr[:5], (B, 1)), p=np.zeros((B, N + 1, 0)), lh=np.zeros((B, N, 0)),
                x_init=np.tile(x0[:, None, :], (1, N + 1, 1)), u_init=np.zeros((B, N, 2)), K=0)


def test_model_outside_the_registry_against_the_oracle_hook(oracle):
    N, B = 8, 3
    ocp = _kinematic_ocp(N, 0.8)
    wl = _kinematic_workload(N, B)
    info = codegen.analyse(ocp.model)
    assert (info.nx, info.nu, info.K, info.in_unit) == (5, 2, 0, (1 << 2) | (1 << 3))   # px, py feed nothing
    os.makedirs(WORK, exist_ok=True)
    oracle.register_generated(codegen.emit_oracle_c(info), WORK)
    spec = oracle.spec_from_ocp(ocp, oracle.MGEN)
    gen = _emu_lib(genbuild.build_emu_lib(info, 0, False, EMU))
    desc = _capi.desc_from_ocp(ocp, batch=B, generated=True)
    xe, ue, xo, uo = wl["x_init"], wl["u_init"], wl["x_init"].copy(), wl["u_init"].copy()
    for it in range(3):
        r = emu_rti(gen, desc, wl, xe, ue)
        xe, ue = r["x"], r["u"]
        xo, uo, sto, ito = util.oracle_rti(oracle, spec, wl, xo, uo)
        assert np.array_equal(r["status"], sto) and (sto == 0).all()
        assert util.rel_err(xe, xo) < 1e-8 and util.rel_err(ue, uo) < 1e-8
    assert np.abs(ue[:, :, 0]).max() <= 1.5 + 1e-9 and np.abs(ue[:, :, 1]).max() <= 0.8 + 1e-9


# ------------------------------------------------------------------ on the device
@pytest.mark.gpu
def test_generated_library_on_the_device(oracle):
    from mpc_collisionavoidance_amd import BatchOcpSolver
    name, N, K, B = "usv_model_pf_ca", 20, 5, 64
    wl = scenario.make_batch(name, N, K, B, seed=3)
    res = []
    for symbolic in (True, False):
        s = BatchOcpSolver(usv_models.make_ocp(name, N * scenario.DT[name], N, K, symbolic=symbolic), B)
        assert s.generated == symbolic
        scenario.load_into(s, wl)
        for it in range(2):
            st = s.solve()
        res.append((s.get_all("x"), s.get_all("u"), st, s.get_int("qp_iter")))
        s.close()
    assert np.array_equal(res[0][2], res[1][2]) and np.abs(res[0][3] - res[1][3]).max() <= 1
    assert util.rel_err(res[0][0], res[1][0]) < 1e-9 and util.rel_err(res[0][1], res[1][1]) < 1e-9


@pytest.mark.gpu
def test_model_outside_the_registry_on_the_device(oracle):
    from mpc_collisionavoidance_amd import BatchOcpSolver
    N, B = 12, 40
    ocp = _kinematic_ocp(N, 1.2)
    wl = _kinematic_workload(N, B, seed=4)
    info = codegen.analyse(ocp.model)
    os.makedirs(WORK, exist_ok=True)
    oracle.register_generated(codegen.emit_oracle_c(info), WORK)
    spec = oracle.spec_from_ocp(ocp, oracle.MGEN)
    s = BatchOcpSolver(ocp, B)
    assert s.generated
    s.set("x0", 0, wl["x0"]); s.set_all("x", wl["x_init"]); s.set_all("u", wl["u_init"])
    s.set_all("yref", wl["yref"]); s.set("yref", N, wl["yref_e"])
    xo, uo = wl["x_init"].copy(), wl["u_init"].copy()
    for it in range(4):
        st = s.solve()
        xo, uo, sto, ito = util.oracle_rti(oracle, spec, wl, xo, uo)
        assert np.array_equal(st, sto) and (st == 0).all()
        assert util.rel_err(s.get_all("x"), xo) < 1e-7 and util.rel_err(s.get_all("u"), uo) < 1e-7
    s.close()
