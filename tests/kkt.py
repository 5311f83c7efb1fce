"""Independent acceptance check of a QP solution (TEST INFRASTRUCTURE): the KKT conditions of the stage QP of one
SQP-RTI iteration, evaluated in numpy for a whole batch at once.

It does not follow anybody's iteration path: for a convex QP a point (dz, pi, lam, t, slacks) that satisfies these
conditions IS a solution, whatever algorithm produced it.  The QP data (A, B, b, H, g, C, d) come from the oracle's
linearisation layer (oracle/usv_oracle.c: usv_linearize), which tests/test_ref_vectors.py pins against vectors derived
from the reference's own model files; the candidate solution comes from whoever is being checked - the device through the
C ABI (usvmpc_get "x" / "u" / "pi" / "lam" / "t" / "sl" / "su"), the lane emulator, or the oracle itself.

The conditions are those of the slack form an HPIPM-style IPM solves, with its four residual families:
    stat : H dz + g + [B A]' pi_{k+1} - [0; pi_k] - C'(lam_l - lam_u) = 0   (free variables only: x_0 is eliminated)
           and, for soft rows, Z s + z - lam - lam_s = 0
    eq   : dx_0 = x0 - xbar_0,  dx_{k+1} = A dx + B du + b
    ineq : C dz + s_l - d_l - t_l = 0,  d_u - C dz + s_u - t_u = 0,  s - ls - t_s = 0,  with t >= 0
    comp : lam * t (every pair), with lam >= 0
Row order of `lam` / `t` (include/usvmpc.h): [bu.., bx.., h..] lower | the same upper | slack rows lower | upper.
"""
import ctypes as C

import numpy as np


def linearize_batch(ob, spec, x, u, x0, yref, yref_e, p, lh):
    """The oracle's QP data of every instance at the iterate (x, u): dict of arrays with a leading batch axis."""
    L = ob.lib()
    B = x.shape[0]
    q = L.usv_qp_alloc(C.byref(spec))
    qc = q.contents
    N, nx, nu, nz, K, nbu, nbx = qc.N, qc.nx, qc.nu, qc.nz, qc.K, qc.nbu, qc.nbx

    def view(ptr, shape):
        n = int(np.prod(shape))
        return np.ctypeslib.as_array(ptr, shape=(max(n, 1),))[:n].reshape(shape)

    views = dict(A=view(qc.A, (N, nx, nx)), B=view(qc.B, (N, nx, nu)), b=view(qc.b, (N, nx)), H=view(qc.H, (N + 1, nz, nz)),
                 g=view(qc.g, (N + 1, nz)), dx0=view(qc.dx0, (nx,)), lbu=view(qc.lbu, (N, nbu)), ubu=view(qc.ubu, (N, nbu)),
                 lbx=view(qc.lbx, (N + 1, nbx)), ubx=view(qc.ubx, (N + 1, nbx)), Cxy=view(qc.Cxy, (N + 1, K, 2)),
                 lg=view(qc.lg, (N + 1, K)), ug=view(qc.ug, (N + 1, K)))
    out = {k: np.empty((B,) + v.shape) for k, v in views.items()}
    arrs = [np.ascontiguousarray(a, dtype=np.float64) for a in (x, u, x0, yref, yref_e, p, lh)]
    dp = ob._dp
    for b in range(B):
        L.usv_linearize(C.byref(spec), *[a[b].ctypes.data_as(dp) for a in arrs], q)
        for k, v in views.items():
            out[k][b] = v
    out.update(N=N, nx=nx, nu=nu, nz=nz, K=K, nbu=nbu, nbx=nbx, soft=int(qc.soft), idxbu=list(qc.idxbu[:nbu]),
               idxbx=list(qc.idxbx[:nbx]), ipx=qc.ipx, ipy=qc.ipy, sbx=list(qc.sbx[:nbx]),
               zl=view(qc.zl, (K + nbx,)).copy(), zu=view(qc.zu, (K + nbx,)).copy(), Zl=view(qc.Zl, (K + nbx,)).copy(),
               Zu=view(qc.Zu, (K + nbx,)).copy(), lsl=view(qc.lsl, (K + nbx,)).copy(), lsu=view(qc.lsu, (K + nbx,)).copy())
    L.usv_qp_free(q)
    return out


def kkt_batch(qp, dz, pi, lam, t, sl=None, su=None):
    """Per-instance KKT residuals of the candidate (dz [B,N+1,nz], pi [B,N+1,nx] (entry 0 unused), lam / t [B,N+1,nlam],
    sl / su [B,N+1,K] for soft h rows) on the QPs `qp` (linearize_batch).  Soft state bounds are not covered here.
    Returns dict of [B] arrays: stat, eq, ineq, comp, neg (most negative lam / t; 0 if none), comp_noslack."""
    N, nx, nu, nz, K, nbu, nbx = (qp[k] for k in ("N", "nx", "nu", "nz", "K", "nbu", "nbx"))
    assert not any(qp["sbx"]), "soft state bounds: use tests/test_oracle_qp.kkt_residuals"
    B = dz.shape[0]
    soft = bool(qp["soft"]) and K > 0
    nrow = nbu + nbx + K
    ns = K if soft else 0
    assert lam.shape[2] == 2 * (nrow + ns), (lam.shape, nrow, ns)
    stat, eq, ineq, comp, neg, comp2 = (np.zeros(B) for _ in range(6))
    if sl is None:
        sl, su = np.zeros((B, N + 1, K)), np.zeros((B, N + 1, K))

    def up(acc, v):
        if v.size:
            np.maximum(acc, np.abs(v).reshape(B, -1).max(axis=1), out=acc)

    up(eq, dz[:, 0, nu:] - qp["dx0"])
    iu = np.asarray(qp["idxbu"], dtype=int)
    ix = nu + np.asarray(qp["idxbx"], dtype=int)
    jx, jy = nu + qp["ipx"], nu + qp["ipy"]
    for k in range(N + 1):
        z = dz[:, k]
        r = np.einsum("bij,bj->bi", qp["H"][:, k], z) + qp["g"][:, k]
        if k < N:
            BA = np.concatenate([qp["B"][:, k], qp["A"][:, k]], axis=2)
            r += np.einsum("bji,bj->bi", BA, pi[:, k + 1])
            up(eq, np.einsum("bij,bj->bi", BA, z) + qp["b"][:, k] - dz[:, k + 1, nu:])
        if k >= 1:
            r[:, nu:] -= pi[:, k]
        L, T = lam[:, k], t[:, k]
        ll, lu, tl, tu = L[:, :nrow], L[:, nrow:2 * nrow], T[:, :nrow], T[:, nrow:2 * nrow]
        # (row, value, lower, upper, lam_l, lam_u, t_l, t_u, s_l, s_u) of the rows this stage has
        zero = np.zeros((B, 0))
        blocks = []
        if k < N and nbu:
            blocks.append((slice(0, nbu), z[:, iu], qp["lbu"][:, k], qp["ubu"][:, k], None))
            r[:, iu] -= ll[:, :nbu] - lu[:, :nbu]
        if 1 <= k < N:
            if nbx:
                blocks.append((slice(nbu, nbu + nbx), z[:, ix], qp["lbx"][:, k], qp["ubx"][:, k], None))
                r[:, ix] -= ll[:, nbu:nbu + nbx] - lu[:, nbu:nbu + nbx]
            if K:
                cx, cy = qp["Cxy"][:, k, :, 0], qp["Cxy"][:, k, :, 1]
                v = cx * z[:, jx:jx + 1] + cy * z[:, jy:jy + 1]
                blocks.append((slice(nbu + nbx, nrow), v, qp["lg"][:, k], qp["ug"][:, k], (sl[:, k], su[:, k]) if soft else None))
                dl = ll[:, nbu + nbx:] - lu[:, nbu + nbx:]
                r[:, jx] -= (dl * cx).sum(axis=1)
                r[:, jy] -= (dl * cy).sum(axis=1)
        used = np.zeros(nrow, dtype=bool)
        for rows, v, lo, hi, s in blocks:
            used[rows] = True
            s_l, s_u = (s if s is not None else (0.0, 0.0))
            up(ineq, v + s_l - lo - tl[:, rows])
            up(ineq, hi - v + s_u - tu[:, rows])
            up(comp, ll[:, rows] * tl[:, rows])
            up(comp, lu[:, rows] * tu[:, rows])
            up(comp2, ll[:, rows] * (v + s_l - lo))
            up(comp2, lu[:, rows] * (hi - v + s_u))
            if s is not None:   # soft h rows: slack stationarity, slack bounds, their complementarity
                o = 2 * nrow
                lsl_, lsu_ = L[:, o:o + K], L[:, o + ns:o + ns + K]
                tsl_, tsu_ = T[:, o:o + K], T[:, o + ns:o + ns + K]
                kk = rows.start - (nbu + nbx)
                assert kk == 0
                up(stat, qp["Zl"][:K] * s_l + qp["zl"][:K] - ll[:, rows] - lsl_)
                up(stat, qp["Zu"][:K] * s_u + qp["zu"][:K] - lu[:, rows] - lsu_)
                up(ineq, s_l - qp["lsl"][:K] - tsl_)
                up(ineq, s_u - qp["lsu"][:K] - tsu_)
                up(comp, lsl_ * tsl_)
                up(comp, lsu_ * tsu_)
                up(comp2, lsl_ * (s_l - qp["lsl"][:K]))
                up(comp2, lsu_ * (s_u - qp["lsu"][:K]))
                for a in (lsl_, lsu_, tsl_, tsu_):
                    np.maximum(neg, np.maximum(0.0, -a).max(axis=1), out=neg)
        for a in (ll[:, used], lu[:, used], tl[:, used], tu[:, used]):
            if a.size:
                np.maximum(neg, np.maximum(0.0, -a).max(axis=1), out=neg)
        # rows the stage does not have must carry no multiplier
        if (~used).any():
            assert not np.any(L[:, :2 * nrow].reshape(B, 2, nrow)[:, :, ~used]), "multiplier on a row the stage does not have"
        sel = np.arange(0, nu) if k == 0 else (np.arange(nu, nz) if k == N else np.arange(nz))
        up(stat, r[:, sel])
    return dict(stat=stat, eq=eq, ineq=ineq, comp=comp, neg=neg, comp_noslack=comp2)


def certified(res, tol_stat=1e-6, tol_eq=1e-8, tol_ineq=1e-8, tol_comp=1e-8):
    """[B] bool: the candidate passes the IPM's own exit test as evaluated here (multipliers and slacks non-negative)."""
    return ((res["stat"] <= tol_stat) & (res["eq"] <= tol_eq) & (res["ineq"] <= tol_ineq) & (res["comp"] <= tol_comp)
            & (res["neg"] == 0.0))
