"""Shared helpers for the parity tests: run the same workload through the CPU oracle and compare."""
import numpy as np

from mpc_collisionavoidance_amd import _capi, scenario, usv_models

MODEL_ID = _capi.MODEL_IDS


def oracle_spec(ob, name, N, dt, K, **opts):
    return ob.spec(MODEL_ID[name], N, N * dt, K if name != "usv_model" else 0, **opts)


def oracle_rti(ob, spec, wl, x, u, x0=None):
    """One batched RTI iteration of the oracle on copies of (x, u)."""
    x, u = x.copy(), u.copy()
    st, it = ob.rti_batch(spec, x, u, wl["x0"] if x0 is None else x0, wl["yref"], wl["yref_e"], wl["p"], wl["lh"])
    return x, u, st, it


def rel_err(a, b, floor=1e-2):
    """Per-component relative error used by every parity test: for each component c of the last axis,
    max |a_c - b_c| over everything else divided by that component's own magnitude max |b_c| (floored at `floor`
    so that identically-zero components - references, parked obstacle slots - do not divide by zero); the
    result is the worst component.  A thrust of 30 N therefore no longer hides an error on a sway velocity of
    0.05 m/s, as one batch-wide scale would."""
    a, b = np.asarray(a, dtype=float), np.asarray(b, dtype=float)
    if a.size == 0:
        return 0.0
    if a.ndim == 0:
        return float(abs(a - b) / max(floor, abs(b)))
    ax = tuple(range(a.ndim - 1))
    scale = np.maximum(floor, np.abs(b).max(axis=ax))
    return float((np.abs(a - b).max(axis=ax) / scale).max())


def rel_err_per_instance(a, b, floor=1e-2):
    """The same norm as rel_err (every component scaled by its own magnitude over the whole batch), but returned per
    instance (first axis): [B] array of each instance's worst component."""
    a, b = np.asarray(a, dtype=float), np.asarray(b, dtype=float)
    if a.shape[0] == 0:
        return np.zeros(0)
    ax = tuple(range(a.ndim - 1))
    scale = np.maximum(floor, np.abs(b).max(axis=ax))
    return (np.abs(a - b) / scale).reshape(a.shape[0], -1).max(axis=1)


def make(name, N, K, B, dt=None, seed=1234, **kw):
    dt = scenario.DT[name] if dt is None else dt
    ocp = usv_models.make_ocp(name, N * dt, N, None if name == "usv_model" else K)
    wl = scenario.make_batch(name, N, K if name != "usv_model" else 0, B, dt=dt, seed=seed, **kw)
    return ocp, wl
