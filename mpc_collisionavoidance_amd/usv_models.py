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

from .acados_template import AcadosModel, AcadosOcp, AcadosOcpSolver, BatchOcpSolver, SymVec


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


def ocp_usv(Tf, N):
    model, constraint = usv_model()
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


def ocp_guidance_ca1(Tf, N, n_obstacles=8):
    K = int(n_obstacles)
    model, constraint = usv_model_guidance_ca1(K)
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


def ocp_pf_ca(Tf, N, n_obstacles=4):
    K = int(n_obstacles)
    model, constraint = usv_model_pf_ca(K)
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


def make_ocp(name, Tf, N, n_obstacles=None):
    if name == "usv_model":
        return ocp_usv(Tf, N)[2]
    if n_obstacles is None:
        return OCP_BUILDERS[name](Tf, N)[2]
    return OCP_BUILDERS[name](Tf, N, n_obstacles)[2]


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
