"""Generates tests/golden/ref_model_<variant>.npz: known answers of the REFERENCE's own model files.

Runs only where /root/reference exists (the build container; the reference never travels to the GPU box).  Each of
the three in-scope model files - catkin_ws/src/nmpc_ca/scripts/{usv_acados,usv_guidance_ca1,usv_pf_ca}/usv_model.py -
is executed unchanged (its `from casadi import *` resolved to mpc_collisionavoidance_amd.casadi_lite), and its
expression graphs are evaluated at seeded points:
    f      the explicit right-hand side model.f_expl_expr, by direct evaluation of the reference's graph;
    J      d f / d [u; x], by SYMPY differentiation of that graph (independent of this repo's forward-mode generator);
    h, dh  the obstacle rows constraint.expr (con_h_expr) and their derivative with respect to x, likewise.
The vectors are data (inputs and expected outputs); no reference source text is stored.  A `-m gpu` test
(tests/test_gpu_ref_vectors.py) compares the device transcription (csrc/models.hpp, obs_dist in csrc/qp_ipm.hpp) with
them through usvmpc_debug_model_eval / usvmpc_debug_obstacle_eval, and a CPU test does the same for the oracle.
This pins the model transcription to the reference; it does NOT pin the solver arithmetic
(acados / HPIPM are absent), so parity stays "unpinned" in the sense of SURVEY.md 8c.

usage: python tests/golden/make_ref_model_vectors.py
"""
import importlib.util
import os
import sys

import numpy as np
import sympy as sp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from mpc_collisionavoidance_amd import casadi_lite as ca  # noqa: E402
from tests.test_symbolic import _to_sympy  # noqa: E402

REF = "/root/reference/catkin_ws/src/nmpc_ca/scripts"
VARIANTS = {"usv_acados": 0, "usv_guidance_ca1": 1, "usv_pf_ca": 2}   # -> usvmpc model id
NPTS = 48


def load(variant):
    ca.install()
    spec = importlib.util.spec_from_file_location("refmodel_" + variant, os.path.join(REF, variant, "usv_model.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.usv_model()


def points(variant, nx, nu, rng):
    """Physically plausible seeded points (plus both branches of the 3-DOF block's if_else and both signs of |.|)."""
    x = rng.normal(size=(NPTS, nx))
    u = rng.normal(size=(NPTS, nu)) * (10.0 if nu == 2 else 0.3)
    if variant == "usv_acados":
        x[:, 0] = rng.uniform(-0.2, 1.5, NPTS); x[:, 1] = rng.uniform(-0.3, 0.3, NPTS); x[:, 2] = rng.uniform(-0.8, 0.8, NPTS)
        x[:, 3:5] = rng.uniform(-30, 35, (NPTS, 2))
        x[::5, 0] = rng.uniform(1.26, 1.5, x[::5, 0].shape)
    elif variant == "usv_guidance_ca1":
        x[:, 0] = rng.uniform(0.2, 1.4, NPTS); x[:, 1] = rng.uniform(-0.2, 0.2, NPTS)
        x[:, 5:7] = rng.uniform(-5, 20, (NPTS, 2))
    else:
        x[:, 0] = rng.uniform(-3.2, 3.2, NPTS); x[:, 3] = rng.uniform(-0.2, 1.5, NPTS); x[:, 4] = rng.uniform(-0.3, 0.3, NPTS)
        x[:, 5] = rng.uniform(-0.8, 0.8, NPTS); x[:, 9] = rng.uniform(-3.2, 3.2, NPTS)
        x[:, 10:12] = rng.uniform(-5, 20, (NPTS, 2)); x[:, 12:14] = rng.uniform(-30, 36, (NPTS, 2))
        x[::5, 3] = rng.uniform(1.26, 1.5, x[::5, 3].shape)
    return x, u


def main():
    for variant, mid in VARIANTS.items():
        model, constraint = load(variant)
        X, U, P = ca.scalars(model.x), ca.scalars(model.U), ca.scalars(model.p)
        F = ca.scalars(model.f_expl_expr)
        nx, nu, npar = len(X), len(U), len(P)
        H = ca.scalars(constraint.expr) if isinstance(getattr(constraint, "expr", None), ca.MXVec) else []
        K = len(H)
        xs, us, ps = sp.symbols("x0:%d" % nx, real=True), sp.symbols("u0:%d" % nu, real=True), sp.symbols("p0:%d" % max(npar, 1), real=True)
        symmap = {**dict(zip(X, xs)), **dict(zip(U, us)), **dict(zip(P, ps))}
        fs = sp.Matrix(_to_sympy(F, symmap))
        Jf = sp.lambdify((xs, us), fs.jacobian(sp.Matrix(list(us) + list(xs))), "numpy")
        rng = np.random.default_rng(20260929 + mid)
        x, u = points(variant, nx, nu, rng)
        f = np.zeros((NPTS, nx)); J = np.zeros((NPTS, nx, nu + nx))
        for i in range(NPTS):
            vals = {**dict(zip(X, x[i])), **dict(zip(U, u[i]))}
            f[i] = ca.evaluate(F, vals)
            J[i] = np.asarray(Jf(x[i], u[i]), dtype=float)
        out = dict(model_id=mid, x=x, u=u, f=f, J=J, K=K)
        if K:
            hs = sp.Matrix(_to_sympy(H, symmap))
            Hf = sp.lambdify((xs, ps), hs, "numpy")
            Hj = sp.lambdify((xs, ps), hs.jacobian(sp.Matrix(list(xs))), "numpy")
            p = rng.uniform(-5, 20, (NPTS, npar))
            p[::7] = 100.0                                   # the reference's parked slots (parameter_values = 100)
            h = np.zeros((NPTS, K)); dh = np.zeros((NPTS, K, nx))
            for i in range(NPTS):
                h[i] = np.asarray(Hf(x[i], p[i]), dtype=float).ravel()
                dh[i] = np.asarray(Hj(x[i], p[i]), dtype=float)
            out.update(p=p, h=h, dh=dh)
        path = os.path.join(HERE, "ref_model_%s.npz" % variant)
        np.savez_compressed(path, **out)
        print(variant, "nx", nx, "nu", nu, "K", K, "->", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
