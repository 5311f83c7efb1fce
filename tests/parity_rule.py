"""The documented parity rule for device-vs-oracle comparisons from identical inputs, in ONE place
(TEST INFRASTRUCTURE; also imported by bench.py's un-timed parity leg):

    every instance within north_star's 1e-5 relative trajectory error of the oracle -
    unless it carries the independent KKT certificate of tests/kkt.py (the device's point satisfies the KKT conditions of
    ITS QP at the exit tolerances acados / HPIPM ask for: stat <= 1e-6, eq / ineq / comp <= 1e-8, lam, t >= 0);
    a certified instance must still stay within CAP = 5e-3, and at most MAX_FRAC of a tick's instances may need the certificate.

An uncertified instance above 1e-5 fails the rule, whatever its size.  usv_model is held to 1e-7 on every instance by its callers and
never gets here.  The soft-row model (usv_model_guidance_ca1) is held by its callers to 1e-7 on 99 % and 1e-5 on all of the instances
that took the oracle's iteration count; the handful (<= 1 % of a tick, asserted by the caller) that stop an iteration apart from the
oracle come here with the tighter cap SOFT_DIT_CAP (tests/test_gpu_closed_loop.py).
"""
import numpy as np

from tests import kkt

NORTH_STAR = 1e-5
CAP = 5e-3
SOFT_DIT_CAP = 1e-3   # soft-row model, instances one iteration apart from the oracle (measured: 1.4e-4)
MAX_FRAC = 0.004
KKT_TOL = (1.02e-6, 1.02e-8, 1.02e-8, 1.02e-8)  # (the checker's QP data is the oracle's linearisation: 2 % slack on the tolerances)


def _pad_pi(pi):
    return np.concatenate([np.zeros_like(pi[:, :1]), pi], axis=1)


def _pad_s(s):
    return np.concatenate([s, np.zeros_like(s[:, :1])], axis=1)


def _step(xn, un, xb, ub):
    B, N, nu = ub.shape
    dz = np.zeros((B, N + 1, nu + xb.shape[2]))
    dz[:, :N, :nu] = un - ub
    dz[:, :, nu:] = xn - xb
    return dz


def certify(ob, spec, solver, idx, xin, uin, x0, data, soft=False):
    """KKT certificate of the device's last solve for the instances `idx`: [len(idx)] bool and the residual dict.
    xin / uin / x0 / data = (yref, yref_e, p, lh): the inputs of that solve for the whole batch."""
    idx = np.asarray(idx, dtype=int)
    if idx.size == 0:
        return np.zeros(0, dtype=bool), {}
    xg, ug = solver.get_all("x")[idx], solver.get_all("u")[idx]
    qp = kkt.linearize_batch(ob, spec, xin[idx], uin[idx], x0[idx], *[d[idx] for d in data])
    res = kkt.kkt_batch(qp, _step(xg, ug, xin[idx], uin[idx]), _pad_pi(solver.get_all("pi")[idx]),
                        solver.get_all("lam")[idx], solver.get_all("t")[idx],
                        _pad_s(solver.get_all("sl")[idx]) if soft else None, _pad_s(solver.get_all("su")[idx]) if soft else None)
    return kkt.certified(res, *KKT_TOL), res


def check(ob, spec, solver, ok, e, xin, uin, x0, data, soft=False, max_frac=MAX_FRAC, cap=CAP):
    """Apply the rule to one tick.  ok: [B] bool, instances converged on both sides; e: [ok.sum()] per-instance error.
    Returns dict(above, certified, uncertified, worst, worst_uncertified, violations): violations is a list of strings, empty when
    the rule holds (the caller asserts on it; bench.py turns it into its exit code)."""
    B = ok.shape[0]
    where = np.where(ok)[0]
    above = where[e > NORTH_STAR]
    out = dict(above=int(above.size), certified=0, uncertified=0, worst=float(e.max()) if e.size else 0.0, worst_uncertified=0.0,
               violations=[])
    if above.size:
        cert, res = certify(ob, spec, solver, above, xin, uin, x0, data, soft=soft)
        ea = e[e > NORTH_STAR]
        out["certified"] = int(cert.sum())
        out["uncertified"] = int((~cert).sum())
        if (~cert).any():
            out["worst_uncertified"] = float(ea[~cert].max())
            out["violations"].append("%d instance(s) above 1e-5 without the KKT certificate: %s (errors %s)"
                                     % ((~cert).sum(), above[~cert].tolist(), ea[~cert].tolist()))
        if ea.max() > cap:
            out["violations"].append("certified instance(s) beyond %g: %s" % (cap, above[ea > cap].tolist()))
        if above.size > max(1, int(max_frac * B)):
            out["violations"].append("%d instances above 1e-5 (allowed: %d)" % (above.size, max(1, int(max_frac * B))))
    return out
