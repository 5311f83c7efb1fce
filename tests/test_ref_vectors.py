"""Known answers derived from the REFERENCE's own model files (tests/golden/ref_model_*.npz, generated in the build
container by tests/golden/make_ref_model_vectors.py: direct evaluation of the reference's CasADi graphs for f and h,
sympy differentiation of those graphs for the Jacobians).

CPU: the oracle's hand restatement of the three models reproduces them.  GPU (`-m gpu`): so does the device
transcription, called exactly as the kernels call it (M::fjvp one tangent column at a time, obs_dist for the obstacle
rows) through usvmpc_debug_model_eval / usvmpc_debug_obstacle_eval.  This pins SURVEY.md 8a rows a2 (right-hand side +
Jacobians feeding the ERK4 / VDE) and a4 (h and its gradient) to the reference on hardware.  The solver arithmetic itself
(acados, HPIPM) is absent from the reference tree and stays unpinned.
"""
import ctypes as C
import os

import numpy as np
import pytest

from mpc_collisionavoidance_amd import _capi

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
VARIANTS = ["usv_acados", "usv_guidance_ca1", "usv_pf_ca"]
POS = {1: (5, 6), 2: (10, 11)}   # (xned, yned) / (nedx, nedy) state indices of the obstacle rows


def _load(variant):
    return np.load(os.path.join(GOLD, "ref_model_%s.npz" % variant))


def _close(a, b, rtol):
    return np.abs(a - b).max() <= rtol * max(1.0, np.abs(b).max())


@pytest.mark.parametrize("variant", VARIANTS)
def test_oracle_models_reproduce_the_reference_vectors(oracle, variant):
    g = _load(variant)
    mid = int(g["model_id"])
    nx, nu = g["x"].shape[1], g["u"].shape[1]
    for i in range(g["x"].shape[0]):
        assert _close(oracle.model_f(mid, g["x"][i], g["u"][i]), g["f"][i], 1e-13)
        Jx, Ju = oracle.model_jac(mid, g["x"][i], g["u"][i])
        assert _close(np.hstack([Ju, Jx]), g["J"][i], 1e-12), (variant, i)
        if int(g["K"]):
            h, Cxy = oracle.model_h(mid, g["x"][i], g["p"][i])
            assert _close(h, g["h"][i], 1e-13)
            ipx, ipy = POS[mid]
            dh = np.zeros((int(g["K"]), nx))
            dh[:, ipx], dh[:, ipy] = Cxy[:, 0], Cxy[:, 1]
            assert _close(dh, g["dh"][i], 1e-12)          # in particular: zero derivative w.r.t. every other state


@pytest.mark.gpu
@pytest.mark.parametrize("variant", VARIANTS)
def test_device_models_reproduce_the_reference_vectors(variant):
    g = _load(variant)
    mid = int(g["model_id"])
    L = _capi.lib()
    x, u = np.ascontiguousarray(g["x"]), np.ascontiguousarray(g["u"])
    n, nx, nu = x.shape[0], x.shape[1], u.shape[1]
    f, J = np.zeros((n, nx)), np.zeros((n, nx, nu + nx))
    dp = _capi._dp
    rc = L.usvmpc_debug_model_eval(mid, 0, n, x.ctypes.data_as(dp), u.ctypes.data_as(dp), f.ctypes.data_as(dp), J.ctypes.data_as(dp))
    assert rc == 0
    assert _close(f, g["f"], 1e-12), np.abs(f - g["f"]).max()
    assert _close(J, g["J"], 1e-11), np.abs(J - g["J"]).max()
    K = int(g["K"])
    if K:
        ipx, ipy = POS[mid]
        pos = np.ascontiguousarray(x[:, [ipx, ipy]])
        p = np.ascontiguousarray(g["p"])
        h, grad = np.zeros((n, K)), np.zeros((n, K, 2))
        rc = L.usvmpc_debug_obstacle_eval(0, n, K, pos.ctypes.data_as(dp), p.ctypes.data_as(dp), h.ctypes.data_as(dp), grad.ctypes.data_as(dp))
        assert rc == 0
        assert _close(h, g["h"], 1e-13), np.abs(h - g["h"]).max()
        dh = g["dh"]
        assert _close(grad[:, :, 0], dh[:, :, ipx], 1e-12) and _close(grad[:, :, 1], dh[:, :, ipy], 1e-12)
        other = np.delete(dh, [ipx, ipy], axis=2)
        assert np.abs(other).max() == 0.0     # the reference's rows depend on the position only: what the kernel assumes
