"""The three in-scope OCP definitions, written the way the reference's scripts write them.

Each `usv_model_*()` returns the `(model, constraint)` pair of the reference's `usv_model.py`
and each `acados_settings_*()` fills an AcadosOcp like the reference's `acados_settings.py`
(paths relative to /root/reference/catkin_ws/src/nmpc_ca/scripts/):

    usv_acados/usv_model.py:40-199          usv_acados/acados_settings.py:40-160
    usv_guidance_ca1/usv_model.py:40-199    usv_guidance_ca1/acados_settings.py:40-209
    usv_pf_ca/usv_model.py:40-230           usv_pf_ca/acados_settings.py:40-190

CasADi is not needed: the dynamics are hand-written device functions selected by `model.name`
(csrc/models.hpp), and the symbolic vectors are length-only placeholders.  Weights, selectors,
bounds and soft-constraint data are plain numbers and are passed through unchanged, so a caller
may edit them on the AcadosOcp before constructing the solver, exactly as with acados.  The
obstacle count is a parameter here (the reference hard-codes 8 resp. 4).
"""
import types

import numpy as np
import scipy.linalg

from . import casadi_lite as ca
from .acados_template import AcadosModel, AcadosOcp, AcadosOcpSolver, BatchOcpSolver, SymVec


# ------------------------------------------------------------------------------------------ symbolic forms
# The same three models written with the CasADi-style layer (casadi_lite), formula by formula as the
# reference's usv_model.py files state them.  With `symbolic=True` the builders below attach these
# expression graphs, and solver_options.model_source = "symbolic" makes the solver compile its device model
# from them (codegen.py) instead of using the hand-written csrc/models.hpp - the route every other model file
# of the reference takes.
def _dof3_sym(c, u, v, r, Tport, Tstbd):
    """usv_acados/usv_model.py:61-77,110-122"""
    X_u_dot, Y_v_dot, Y_r_dot, N_v_dot, N_r_dot = -2.25, -23.13, -1.31, -16.41, -2.79
    Yvv, Yvr, Nrv, Nrr = -99.99, -5.49, -8.8, -3.49
    m, Iz, B = 30, 4.1, 0.41
    Xu = ca.if_else(u > 1.25, 64.55, -25)
    Xuu = ca.if_else(u > 1.25, -70.92, 0)
    Yv = 0.5 * (-40 * 1000 * ca.fabs(v)) * (1.1 + 0.0045 * (1.01 / 0.09) - 0.1 * (0.27 / 0.09) + 0.016 * ((0.27 / 0.09) * (0.27 / 0.09)))
    Nr = (-0.52) * ca.sqrt(u * u + v * v)
    Tu = Tport + c * Tstbd
    Tr = (Tport - c * Tstbd) * B / 2
    return (
        (Tu - (-m + 2 * Y_v_dot) * v - (Y_r_dot + N_v_dot) * r * r - (-Xu * u - Xuu * ca.fabs(u) * u)) / (m - X_u_dot),
        (-(m - X_u_dot) * u * r - (-Yv - Yvv * ca.fabs(v) - Yvr * ca.fabs(r)) * v) / (m - Y_v_dot),
        (Tr - (-2 * Y_v_dot * u * v - (Y_r_dot + N_v_dot) * r * u + X_u_dot * u * r) - (-Nr * r - Nrv * ca.fabs(v) * r - Nrr * ca.fabs(r) * r)) / (Iz - N_r_dot),
    )


def _distances(px, py, p, K):
    return ca.vertcat(*[ca.sqrt((px - p[2 * i]) * (px - p[2 * i]) + (py - p[2 * i + 1]) * (py - p[2 * i + 1])) for i in range(K)])


def _attach_symbolic(model, constraint, name, K):
    if name == "usv_model":
        u, v, r, Tport, Tstbd = (ca.MX.sym(n) for n in ("u", "v", "r", "Tport", "Tstbd"))
        U0, U1 = ca.MX.sym("UTportdot"), ca.MX.sym("UTstbddot")
        fu, fv, fr = _dof3_sym(0.78, u, v, r, Tport, Tstbd)
        model.x, model.U, model.p = ca.vertcat(u, v, r, Tport, Tstbd), ca.vertcat(U0, U1), ca.vertcat([])
        model.f_expl_expr = ca.vertcat(fu, fv, fr, U0, U1)
        constraint.expr = None
    elif name == "usv_model_guidance_ca1":
        u, v, ye, chie, psied, xned, yned, psi = (ca.MX.sym(n) for n in ("u", "v", "ye", "chie", "psied", "xned", "yned", "psi"))
        U = ca.MX.sym("Upsieddot")
        p = [ca.MX.sym("o%d" % i) for i in range(2 * K)]
        T1 = 1.0
        beta = ca.atan2(v, u + 0.001)
        psie = chie - beta
        model.x, model.U, model.p = ca.vertcat(u, v, ye, chie, psied, xned, yned, psi), ca.vertcat(U), ca.vertcat(*p)
        model.f_expl_expr = ca.vertcat(0, 0, u * ca.sin(psie) + v * ca.cos(psie), (psied - psie) / T1, U,
                                       u * ca.cos(psi) - v * ca.sin(psi), u * ca.sin(psi) + v * ca.cos(psi), (psied - psie) / T1)
        constraint.expr = _distances(xned, yned, p, K)
    else:
        names = ("psi", "sinpsi", "cospsi", "u", "v", "r", "ye", "x1", "y1", "ak", "nedx", "nedy", "Tport", "Tstbd")
        psi, sinpsi, cospsi, u, v, r, ye, x1, y1, ak, nedx, nedy, Tport, Tstbd = (ca.MX.sym(n) for n in names)
        U0, U1 = ca.MX.sym("UTportdot"), ca.MX.sym("UTstbddot")
        p = [ca.MX.sym("o%d" % i) for i in range(2 * K)]
        c = 1.0
        fu, fv, fr = _dof3_sym(c, u, v, r, Tport, Tstbd)
        chi = psi + ca.atan2(v, u + .001)
        model.x = ca.vertcat(psi, sinpsi, cospsi, u, v, r, ye, x1, y1, ak, nedx, nedy, Tport, Tstbd)
        model.U, model.p = ca.vertcat(U0, U1), ca.vertcat(*p)
        model.f_expl_expr = ca.vertcat(
            r, ca.cos(chi) * r, -ca.sin(chi) * r, fu, fv, fr,
            -(u * ca.cos(psi) - v * ca.sin(psi)) * ca.sin(ak) + (u * ca.sin(psi) + v * ca.cos(psi)) * ca.cos(ak),
            0, 0, 0, u * ca.cos(psi) - v * ca.sin(psi), u * ca.sin(psi) + v * ca.cos(psi), U0, U1 / c)
        constraint.expr = _distances(nedx, nedy, p, K)
    model.xdot = ca.MX.sym("xdot", len(model.x)) if len(model.x) > 1 else ca.vertcat(ca.MX.sym("xdot"))
    model.z = ca.vertcat([])
    model.f_impl_expr = model.xdot - model.f_expl_expr


# ------------------------------------------------------------------------------------------ M0
def usv_model():
    """`usv_model`: 3-DOF speed controller, no obstacles (usv_acados/usv_model.py)."""
    model = types.SimpleNamespace()
    constraint = types.SimpleNamespace()
    model.name = "usv_model"
    model.x = SymVec(5, ["u", "v", "r", "Tport", "Tstbd"])
    model.xdot = SymVec(5)
    model.U = SymVec(2, ["UTportdot", "UTstbddot"])
    model.z = SymVec(0)
    model.p = SymVec(0)
    model.f_expl_expr = SymVec(5)
    model.f_impl_expr = model.xdot - model.f_expl_expr
    model.u_min, model.u_max = -1.5, 1.5
    model.Tport_min = model.Tstbd_min = -30
    model.Tport_max = model.Tstbd_max = 35
    model.r_min, model.r_max = -1.0, 1.0
    model.Tstbddot_min = model.Tportdot_min = -30
    model.Tstbddot_max = model.Tportdot_max = 30
    model.x0 = np.array([0.001, 0, 0, 0, 0])
    constraint.expr = None
    return model, constraint


def _common(model, constraint, N, Tf):
    ocp = AcadosOcp()
    m = AcadosModel()
    m.f_impl_expr, m.f_expl_expr = model.f_impl_expr, model.f_expl_expr
    m.x, m.xdot, m.u, m.z, m.p, m.name = model.x, model.xdot, model.U, model.z, model.p, model.name
    m.con_h_expr = constraint.expr
    ocp.model = m
    ocp.dims.N = N
    ocp.cost.cost_type = "LINEAR_LS"
    ocp.cost.cost_type_e = "LINEAR_LS"
    ocp.solver_options.tf = Tf
    ocp.solver_options.qp_solver = "PARTIAL_CONDENSING_HPIPM"
    ocp.solver_options.nlp_solver_type = "SQP_RTI"
    ocp.solver_options.hessian_approx = "GAUSS_NEWTON"
    ocp.solver_options.integrator_type = "ERK"
    return ocp


def ocp_usv(Tf, N, symbolic=False):
    model, constraint = usv_model()
    if symbolic:
        _attach_symbolic(model, constraint, "usv_model", 0)
    ocp = _common(model, constraint, N, Tf)
    nx, nu = 5, 2
    ny = nx + nu
    Q = np.diag([1e3, 1e-3, 1e3, 1e-1, 1e-1])
    R = np.eye(nu)
    R[0, 0] = 1e-2
    R[1, 1] = 1e-2
    Qe = np.diag([5e3, 5e-3, 5e3, 5e-1, 5e-1])
    ocp.cost.W = scipy.linalg.block_diag(Q, R)
    ocp.cost.W_e = Qe
    Vx = np.zeros((ny, nx))
    Vx[:nx, :nx] = np.eye(nx)
    ocp.cost.Vx = Vx
    Vu = np.zeros((ny, nu))
    Vu[5, 0] = 1.0
    Vu[6, 1] = 1.0
    ocp.cost.Vu = Vu
    ocp.cost.Vx_e = np.eye(nx)
    ocp.cost.yref = np.zeros(ny)
    ocp.cost.yref_e = np.zeros(nx)
    ocp.constraints.lbx = np.array([model.u_min, model.u_min, model.r_min, model.Tport_min, model.Tstbd_min])
    ocp.constraints.ubx = np.array([model.u_max, model.u_max, model.r_max, model.Tport_max, model.Tstbd_max])
    ocp.constraints.idxbx = np.array([0, 1, 2, 3, 4])
    ocp.constraints.lbu = np.array([model.Tportdot_min, model.Tstbddot_min])
    ocp.constraints.ubu = np.array([model.Tportdot_max, model.Tstbddot_max])
    ocp.constraints.idxbu = np.array([0, 1])
    ocp.constraints.x0 = model.x0
    return constraint, model, ocp


# ------------------------------------------------------------------------------------------ M1
def usv_model_guidance_ca1(n_obstacles=8):
    """`usv_model_guidance_ca1`: kinematic guidance with soft circular obstacles."""
    K = int(n_obstacles)
    model = types.SimpleNamespace()
    constraint = types.SimpleNamespace()
    model.name = "usv_model_guidance_ca1"
    model.x = SymVec(8, ["u", "v", "ye", "chie", "psied", "xned", "yned", "psi"])
    model.xdot = SymVec(8)
    model.U = SymVec(1, ["Upsieddot"])
    model.z = SymVec(0)
    model.p = SymVec(2 * K)
    model.f_expl_expr = SymVec(8)
    model.f_impl_expr = model.xdot - model.f_expl_expr
    model.psied_min, model.psied_max = -np.pi, np.pi
    model.psieddot_min, model.psieddot_max = -0.5, 0.5
    model.Upsieddot_min, model.Upsieddot_max = -0.5, 0.5
    constraint.distance_min = 1.5
    model.x0 = np.zeros(8)
    constraint.expr = SymVec(K)
    model.params = types.SimpleNamespace(T1=1.0)
    return model, constraint


def ocp_guidance_ca1(Tf, N, n_obstacles=8, symbolic=False):
    K = int(n_obstacles)
    model, constraint = usv_model_guidance_ca1(K)
    if symbolic:
        _attach_symbolic(model, constraint, "usv_model_guidance_ca1", K)
    ocp = _common(model, constraint, N, Tf)
    nx, nu = 8, 1
    ny = nx + nu
    Q = np.diag([0, 0, 0.05, 0.01, 0, 0, 0, 0])
    R = np.eye(nu)
    R[0, 0] = 0.2
    Qe = np.diag([0, 0, 0.1, 0.05, 0, 0, 0, 0])
    ocp.cost.W = scipy.linalg.block_diag(Q, R)
    ocp.cost.W_e = Qe
    Vx = np.zeros((ny, nx))
    Vx[:nx, :nx] = np.eye(nx)
    ocp.cost.Vx = Vx
    Vu = np.zeros((ny, nu))
    Vu[8, 0] = 1.0
    ocp.cost.Vu = Vu
    ocp.cost.Vx_e = np.eye(nx)
    ocp.cost.zl = 1 * np.ones((K,))
    ocp.cost.Zl = 0 * np.ones((K,))
    ocp.cost.zu = 1 * np.ones((K,))
    ocp.cost.Zu = 0 * np.ones((K,))
    ocp.cost.yref = np.zeros(ny)
    ocp.cost.yref_e = np.zeros(nx)
    ocp.constraints.lbu = np.array([model.Upsieddot_min])
    ocp.constraints.ubu = np.array([model.Upsieddot_max])
    ocp.constraints.idxbu = np.array([0])
    ocp.constraints.lh = constraint.distance_min * np.ones(K)
    ocp.constraints.uh = 1000000 * np.ones(K)
    ocp.constraints.lsh = -0.2 * np.ones(K)
    ocp.constraints.ush = np.zeros(K)
    ocp.constraints.idxsh = np.arange(K)
    ocp.constraints.x0 = model.x0
    ocp.parameter_values = 100 * np.ones(2 * K)
    return constraint, model, ocp


# ------------------------------------------------------------------------------------------ M2
def usv_model_pf_ca(n_obstacles=4):
    """`usv_model_pf_ca`: 3-DOF path following with hard circular obstacles."""
    K = int(n_obstacles)
    model = types.SimpleNamespace()
    constraint = types.SimpleNamespace()
    model.name = "usv_model_pf_ca"
    model.x = SymVec(14, ["psi", "sinpsi", "cospsi", "u", "v", "r", "ye", "x1", "y1", "ak", "nedx", "nedy",
                          "Tport", "Tstbd"])
    model.xdot = SymVec(14)
    model.U = SymVec(2, ["UTportdot", "UTstbddot"])
    model.z = SymVec(0)
    model.p = SymVec(2 * K)
    model.f_expl_expr = SymVec(14)
    model.f_impl_expr = model.xdot - model.f_expl_expr
    model.u_min, model.u_max = -2.0, 2.0
    model.Tport_min = model.Tstbd_min = -30
    model.Tport_max = model.Tstbd_max = 36.5
    model.r_min, model.r_max = -10.0, 10.0
    model.Tstbddot_min = model.Tportdot_min = -30
    model.Tstbddot_max = model.Tportdot_max = 30
    constraint.distance_min = 0.0
    starting_angle = 0.00
    x1, y1, x2, y2 = 1.0, -1.0, 1.0, 3.8
    ak = np.arctan2(y2 - y1, x2 - x1)
    model.x0 = np.array([starting_angle, np.sin(starting_angle), np.cos(starting_angle), 0.001, 0.00, 0.00, 0.0,
                         x1, y1, ak, 0, 0, 0.00, 0.00])
    constraint.expr = SymVec(K)
    return model, constraint


def ocp_pf_ca(Tf, N, n_obstacles=4, symbolic=False):
    K = int(n_obstacles)
    model, constraint = usv_model_pf_ca(K)
    if symbolic:
        _attach_symbolic(model, constraint, "usv_model_pf_ca", K)
    ocp = _common(model, constraint, N, Tf)
    nx, nu = 14, 2
    ny = nx + nu
    Q = np.diag([0, 0.3, 0.3, 80.0, 0, 0, 0.8, 0, 0, 0, 0, 0, 0.0001, 0.0001])
    R = np.eye(nu)
    R[0, 0] = 0.0
    R[1, 1] = 0.0
    Qe = np.diag([0, 0.5, 0.5, 100.0, 0, 0, 1.0, 0, 0, 0, 0, 0, 0.0005, 0.0005])
    ocp.cost.W = scipy.linalg.block_diag(Q, R)
    ocp.cost.W_e = Qe
    Vx = np.zeros((ny, nx))
    Vx[:nx, :nx] = np.eye(nx)
    ocp.cost.Vx = Vx
    Vu = np.zeros((ny, nu))
    Vu[8, 0] = 1.0  # as the reference has it (usv_pf_ca/acados_settings.py:115-117)
    Vu[9, 1] = 1.0
    ocp.cost.Vu = Vu
    ocp.cost.Vx_e = np.eye(nx)
    ocp.cost.yref = np.zeros(ny)
    ocp.cost.yref_e = np.zeros(nx)
    ocp.constraints.lbx = np.array([model.u_min, model.u_min, model.r_min, model.Tport_min, model.Tstbd_min])
    ocp.constraints.ubx = np.array([model.u_max, model.u_max, model.r_max, model.Tport_max, model.Tstbd_max])
    ocp.constraints.idxbx = np.array([3, 4, 5, 12, 13])
    ocp.constraints.lbu = np.array([model.Tportdot_min, model.Tstbddot_min])
    ocp.constraints.ubu = np.array([model.Tportdot_max, model.Tstbddot_max])
    ocp.constraints.idxbu = np.array([0, 1])
    ocp.constraints.lh = constraint.distance_min * np.ones(K)
    ocp.constraints.uh = 1000000 * np.ones(K)
    ocp.constraints.x0 = model.x0
    ocp.parameter_values = np.zeros(2 * K)
    return constraint, model, ocp


OCP_BUILDERS = {"usv_model": ocp_usv, "usv_model_guidance_ca1": ocp_guidance_ca1, "usv_model_pf_ca": ocp_pf_ca}


def make_ocp(name, Tf, N, n_obstacles=None, symbolic=False):
    """symbolic=True: attach the expression graphs and route the solver through the code generator."""
    if name == "usv_model":
        ocp = ocp_usv(Tf, N, symbolic=symbolic)[2]
    elif n_obstacles is None:
        ocp = OCP_BUILDERS[name](Tf, N, symbolic=symbolic)[2]
    else:
        ocp = OCP_BUILDERS[name](Tf, N, n_obstacles, symbolic=symbolic)[2]
    if symbolic:
        ocp.solver_options.model_source = "symbolic"
    return ocp


def acados_settings(Tf, N, name="usv_model_guidance_ca1", n_obstacles=None, device=0):
    """`acados_settings(Tf, N)` of the reference: returns (constraint, model, acados_solver)."""
    if name == "usv_model":
        constraint, model, ocp = ocp_usv(Tf, N)
    elif n_obstacles is None:
        constraint, model, ocp = OCP_BUILDERS[name](Tf, N)
    else:
        constraint, model, ocp = OCP_BUILDERS[name](Tf, N, n_obstacles)
    return constraint, model, AcadosOcpSolver(ocp, json_file="acados_ocp.json", device=device)


def batch_settings(Tf, N, batch, name="usv_model_pf_ca", n_obstacles=None, device=0):
    ocp = make_ocp(name, Tf, N, n_obstacles)
    return ocp, BatchOcpSolver(ocp, batch, device=device)
