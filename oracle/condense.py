"""Partial condensing of the OCP QP, restated in numpy, with its own interior-point solve.

TEST INFRASTRUCTURE ONLY (see oracle/usv_oracle.h): imported by tests/ and tools/, never by the product package.

What acados' qp_solver = PARTIAL_CONDENSING_HPIPM does when qp_solver_cond_N = N2 < N (BASELINE.json configs[4]:
N = 80 -> N2 = 10; /root/reference/catkin_ws/src/nmpc_ca/scripts/usv_pf_ca/acados_settings.py:172 selects the solver,
the reference leaves cond_N at its default N, i.e. blocks of one stage):

  part_cond   HPIPM d_part_cond_qp restated (blocks of N / N2 stages, the first N mod N2 of them one longer) - consecutive
              stages k0 .. k0+M-1 become ONE stage whose state is x_k0 and
              whose input is the stack (u_k0, .., u_k0+M-1); the intermediate states are eliminated through the
              dynamics, x_{k0+j} = Phi_j x_k0 + Gam_j u_hat + c_j, which turns the block's cost into a dense
              (nx + M nu)^2 Hessian and EVERY inequality row of an intermediate stage (input bound, state bound,
              obstacle row) into a dense general row in (u_hat, x_k0);
  solve       the oracle's Mehrotra predictor-corrector IPM (oracle/usv_oracle.c usv_qp_solve: same cold start, step
              rule and exit test) on that condensed QP, Riccati recursion over the N2 dense stages;
  expand      back to the N stages (HPIPM d_part_cond_qp_expand_sol, primal part).

The condensed and the uncondensed QP have the same solution; their IPM iterates differ because the cold start
z = 0 means "intermediate states on the linearised dynamics" for the condensed QP and "intermediate states unchanged"
for the uncondensed one.  Hard rows only (usv_model_pf_ca, the model of configs[4]).
"""
import numpy as np


def part_cond(qp, N2):
    """qp: dict of oracle.binding.linearize_and_solve(..., solve=False).  Returns the condensed QP: a list of N2 + 1
    stage dicts (H, g, A, B, b, C, dl, du) plus what expand() needs."""
    N, nx, nu, nz, K = qp["N"], qp["nx"], qp["nu"], qp["nz"], qp["K"]
    if not 1 <= N2 <= N:
        raise ValueError("the number of blocks must lie in 1..N")
    if qp["soft"] or any(qp["sbx"]):
        raise NotImplementedError("hard rows only")
    # HPIPM d_part_cond_qp_compute_block_size: N1 = N / N2 stages per block, the first N - N2 N1 blocks one more
    N1, R1 = N // N2, N - N2 * (N // N2)
    Ms = [N1 + (1 if i < R1 else 0) for i in range(N2)]
    k0s = [i * N1 + min(i, R1) for i in range(N2)]
    ipx, ipy = nu + qp["ipx"], nu + qp["ipy"]
    stages, maps = [], []
    for i in range(N2):
        k0, M = k0s[i], Ms[i]
        nuh = M * nu
        nzh = nuh + nx
        Phi, Gam, c = np.eye(nx), np.zeros((nx, nuh)), np.zeros(nx)
        Hh, gh = np.zeros((nzh, nzh)), np.zeros(nzh)
        Cs, dls, dus, Ts = [], [], [], []
        for j in range(M):
            k = k0 + j
            # z_k = [u_k; x_k] = T w + d,  w = [u_hat; x_k0]
            T = np.zeros((nz, nzh))
            T[np.arange(nu), j * nu + np.arange(nu)] = 1.0
            T[nu:, :nuh] = Gam
            T[nu:, nuh:] = Phi
            d = np.concatenate([np.zeros(nu), c])
            Ts.append((T, d))
            H, g = qp["H"][k], qp["g"][k]
            Hh += T.T @ H @ T
            gh += T.T @ (g + H @ d)
            rows = []   # (c vector over z_k, dl, du) in the oracle's row order
            for r, jb in enumerate(qp["idxbu"]):
                e = np.zeros(nz); e[jb] = 1.0
                rows.append((e, qp["lbu"][k, r], qp["ubu"][k, r]))
            if k >= 1:
                for r, jb in enumerate(qp["idxbx"]):
                    e = np.zeros(nz); e[nu + jb] = 1.0
                    rows.append((e, qp["lbx"][k, r], qp["ubx"][k, r]))
                for r in range(K):
                    e = np.zeros(nz); e[ipx], e[ipy] = qp["Cxy"][k, r]
                    rows.append((e, qp["lg"][k, r], qp["ug"][k, r]))
            for e, lo, hi in rows:
                Cs.append(e @ T)
                dls.append(lo - e @ d)
                dus.append(hi - e @ d)
            A, B, b = qp["A"][k], qp["B"][k], qp["b"][k]
            Gam = A @ Gam
            Gam[:, j * nu:(j + 1) * nu] += B
            c = A @ c + b
            Phi = A @ Phi
        stages.append(dict(H=Hh, g=gh, A=Phi, B=Gam, b=c, C=np.array(Cs).reshape(-1, nzh), dl=np.array(dls), du=np.array(dus),
                           nu=nuh))
        maps.append(Ts)
    HN = qp["H"][N][nu:, nu:]
    stages.append(dict(H=HN, g=qp["g"][N][nu:], A=None, B=None, b=None, C=np.zeros((0, nx)), dl=np.zeros(0), du=np.zeros(0), nu=0))
    return dict(stages=stages, maps=maps, nx=nx, nu=nu, N=N, N2=N2, Ms=Ms, k0s=k0s, dx0=qp["dx0"].copy())


def solve(cq, opts):
    """Mehrotra IPM of oracle/usv_oracle.c (usv_qp_solve) on the dense-stage QP.  opts: dict(qp_iter_max, mu0, thr0,
    tol_stat, tol_eq, tol_ineq, tol_comp, alpha_min, cond_pred_corr, cpc_factor).  Returns dict(w [per stage], status, iter, res,
    cpc_fallbacks)."""
    st, nx = cq["stages"], cq["nx"]
    S = len(st)
    w = [np.zeros(s["nu"] + nx) for s in st]
    pi = [np.zeros(nx) for _ in st]
    ll, lu, tl, tu = [], [], [], []
    for s in st:
        m = s["dl"].size
        a, b = np.maximum(0.0 - s["dl"], opts["thr0"]), np.maximum(s["du"] - 0.0, opts["thr0"])
        tl.append(a); tu.append(b); ll.append(opts["mu0"] / a); lu.append(opts["mu0"] / b)
    nc = 2 * sum(s["dl"].size for s in st)

    def residuals():
        rgs, rbs, rdl, rdu = [], [], [], []
        rg = rb = rd = rm = 0.0
        mu = 0.0
        e0 = cq["dx0"] - w[0][st[0]["nu"]:]
        rb = max(rb, np.abs(e0).max())
        for k, s in enumerate(st):
            nuk = s["nu"]
            r = s["g"] + s["H"] @ w[k]
            if k < S - 1:
                BA = np.hstack([s["B"], s["A"]])
                rbk = s["b"] + BA @ w[k] - w[k + 1][st[k + 1]["nu"]:]
                rbs.append(rbk)
                rb = max(rb, np.abs(rbk).max())
                r = r + BA.T @ pi[k + 1]
            if k >= 1:
                r[nuk:] -= pi[k]
            v = s["C"] @ w[k]
            r = r - s["C"].T @ (ll[k] - lu[k])
            a, b = v - s["dl"] - tl[k], s["du"] - v - tu[k]
            rdl.append(a); rdu.append(b)
            if a.size:
                rd = max(rd, np.abs(a).max(), np.abs(b).max())
                rm = max(rm, (ll[k] * tl[k]).max(), (lu[k] * tu[k]).max())
                mu += (ll[k] * tl[k]).sum() + (lu[k] * tu[k]).sum()
            rgs.append(r)
            lo = nuk if k == 0 else 0       # x_0 is not a variable
            if k == 0:
                rg = max(rg, np.abs(r[:nuk]).max() if nuk else 0.0)
            else:
                rg = max(rg, np.abs(r).max())
        return rgs, rbs, rdl, rdu, e0, np.array([rg, rb, rd, rm]), (mu / nc if nc else 0.0)

    def kkt_step(rgs, rbs, rdl, rdu, e0, ml, mu_, factor, cache):
        """Riccati solve of the reduced system; returns dw, dpi, dll, dlu, dtl, dtu."""
        gt = []
        for k, s in enumerate(st):
            Gl, Gu = ll[k] / tl[k], lu[k] / tu[k]
            gl = ml[k] / tl[k] + Gl * rdl[k]
            gu = mu_[k] / tu[k] + Gu * rdu[k]
            gt.append(rgs[k] + s["C"].T @ (gl - gu))
            if factor:
                cache["Ht"][k] = s["H"] + s["C"].T @ ((Gl + Gu)[:, None] * s["C"])
        if factor:
            P = cache["Ht"][S - 1].copy()
            cache["P"][S - 1] = P
            for k in range(S - 2, -1, -1):
                s = st[k]
                nuk = s["nu"]
                BA = np.hstack([s["B"], s["A"]])
                G = cache["Ht"][k] + BA.T @ cache["P"][k + 1] @ BA
                Luu = np.linalg.cholesky(G[:nuk, :nuk])
                Lxu = np.linalg.solve(Luu, G[:nuk, nuk:]).T       # G_xu Luu^-T
                cache["Luu"][k], cache["Lxu"][k] = Luu, Lxu
                cache["P"][k] = G[nuk:, nuk:] - Lxu @ Lxu.T
                cache["Pb"][k] = cache["P"][k + 1] @ rbs[k]
        p = [None] * S
        lus = [None] * S
        p[S - 1] = gt[S - 1].copy()
        for k in range(S - 2, -1, -1):
            s = st[k]
            nuk = s["nu"]
            BA = np.hstack([s["B"], s["A"]])
            rq = gt[k] + BA.T @ (cache["Pb"][k] + p[k + 1])
            lus[k] = np.linalg.solve(cache["Luu"][k], rq[:nuk])
            p[k] = rq[nuk:] - cache["Lxu"][k] @ lus[k]
        dw = [None] * S
        dpi = [np.zeros(nx) for _ in st]
        dx = e0.copy()
        for k in range(S - 1):
            s = st[k]
            BA = np.hstack([s["B"], s["A"]])
            t = lus[k] + cache["Lxu"][k].T @ dx
            du = -np.linalg.solve(cache["Luu"][k].T, t)
            dw[k] = np.concatenate([du, dx])
            dx = rbs[k] + BA @ dw[k]
            dpi[k + 1] = p[k + 1] + cache["P"][k + 1] @ dx
        dw[S - 1] = dx
        dll, dlu, dtl, dtu = [], [], [], []
        for k, s in enumerate(st):
            wv = s["C"] @ dw[k]
            a = wv + rdl[k]
            b = -wv + rdu[k]
            dtl.append(a); dtu.append(b)
            dll.append(-(ml[k] + ll[k] * a) / tl[k])
            dlu.append(-(mu_[k] + lu[k] * b) / tu[k])
        return dw, dpi, dll, dlu, dtl, dtu

    def step_length(dll, dlu, dtl, dtu):
        a = 1.0
        for k in range(S):
            for v, dv in ((ll[k], dll[k]), (lu[k], dlu[k]), (tl[k], dtl[k]), (tu[k], dtu[k])):
                neg = dv < 0.0
                if neg.any():
                    a = min(a, float((-v[neg] / dv[neg]).min()))
        return a

    status, it, fallbacks = 1, 0, 0
    cache = dict(Ht=[None] * S, P=[None] * S, Luu=[None] * S, Lxu=[None] * S, Pb=[None] * S)
    rgs, rbs, rdl, rdu, e0, res, mu = residuals()
    for it in range(opts["qp_iter_max"]):
        if not np.all(np.isfinite(res)):
            status = 3
            break
        if res[0] <= opts["tol_stat"] and res[1] <= opts["tol_eq"] and res[2] <= opts["tol_ineq"] and res[3] <= opts["tol_comp"]:
            status = 0
            break
        ml = [ll[k] * tl[k] for k in range(S)]
        mu_ = [lu[k] * tu[k] for k in range(S)]
        try:
            dw, dpi, dll, dlu, dtl, dtu = kkt_step(rgs, rbs, rdl, rdu, e0, ml, mu_, True, cache)
        except np.linalg.LinAlgError:
            status = 3
            break
        a_aff = step_length(dll, dlu, dtl, dtu)
        if nc:
            mu_aff = sum(((ll[k] + a_aff * dll[k]) * (tl[k] + a_aff * dtl[k])).sum() +
                         ((lu[k] + a_aff * dlu[k]) * (tu[k] + a_aff * dtu[k])).sum() for k in range(S)) / nc
            sigma = (mu_aff / mu) ** 3
            ml = [ll[k] * tl[k] + dll[k] * dtl[k] - sigma * mu for k in range(S)]
            mu_ = [lu[k] * tu[k] + dlu[k] * dtu[k] - sigma * mu for k in range(S)]
            dw, dpi, dll, dlu, dtl, dtu = kkt_step(rgs, rbs, rdl, rdu, e0, ml, mu_, False, cache)
            if opts.get("cond_pred_corr", 0):
                # HPIPM's conditional predictor-corrector as oracle/usv_oracle.c has it: a corrected step that leaves the duality measure
                # above cpc_factor x the predictor's is replaced by the centring-only one
                a_pc = step_length(dll, dlu, dtl, dtu)
                mu_pc = sum(((ll[k] + a_pc * dll[k]) * (tl[k] + a_pc * dtl[k])).sum() +
                            ((lu[k] + a_pc * dlu[k]) * (tu[k] + a_pc * dtu[k])).sum() for k in range(S)) / nc
                if mu_pc > opts.get("cpc_factor", 2.0) * mu_aff:
                    ml = [ll[k] * tl[k] - sigma * mu for k in range(S)]
                    mu_ = [lu[k] * tu[k] - sigma * mu for k in range(S)]
                    dw, dpi, dll, dlu, dtl, dtu = kkt_step(rgs, rbs, rdl, rdu, e0, ml, mu_, False, cache)
                    fallbacks += 1
        a = step_length(dll, dlu, dtl, dtu)
        if a < opts["alpha_min"]:
            status = 2
            break
        a = a * ((1.0 - a) * 0.99 + a * 0.9999999)
        for k in range(S):
            w[k] = w[k] + a * dw[k]
            if k >= 1:
                pi[k] = pi[k] + a * dpi[k]
            ll[k] = ll[k] + a * dll[k]; lu[k] = lu[k] + a * dlu[k]
            tl[k] = tl[k] + a * dtl[k]; tu[k] = tu[k] + a * dtu[k]
        rgs, rbs, rdl, rdu, e0, res, mu = residuals()
    else:
        it = opts["qp_iter_max"]
        if res[0] <= opts["tol_stat"] and res[1] <= opts["tol_eq"] and res[2] <= opts["tol_ineq"] and res[3] <= opts["tol_comp"]:
            status = 0
    return dict(w=w, status=status, iter=it, res=res, cpc_fallbacks=fallbacks)


def expand(cq, sol):
    """dz [N+1, nz] of the original stages from the condensed solution."""
    nx, nu, N = cq["nx"], cq["nu"], cq["N"]
    dz = np.zeros((N + 1, nu + nx))
    for i, Ts in enumerate(cq["maps"]):
        for j, (T, d) in enumerate(Ts):
            dz[cq["k0s"][i] + j] = T @ sol["w"][i] + d
    dz[N, nu:] = sol["w"][-1]
    return dz


def evaluate(cq, dz):
    """A step dz [N+1, nz] of the ORIGINAL stages judged as a point of the CONDENSED QP: (objective, largest violation of
    its inequality rows, largest dynamics / initial-state residual).  The block variables are read off dz (u_hat = the
    block's inputs, x_hat = the state at the block's first stage)."""
    nx, nu, N = cq["nx"], cq["nu"], cq["N"]
    st = cq["stages"]
    w = []
    for i in range(cq["N2"]):
        k0, M = cq["k0s"][i], cq["Ms"][i]
        w.append(np.concatenate([dz[k0:k0 + M, :nu].reshape(-1), dz[k0, nu:]]))
    w.append(dz[N, nu:].copy())
    obj, viol = 0.0, 0.0
    eq = float(np.abs(cq["dx0"] - w[0][st[0]["nu"]:]).max())
    for k, s in enumerate(st):
        obj += 0.5 * w[k] @ s["H"] @ w[k] + s["g"] @ w[k]
        if s["dl"].size:
            v = s["C"] @ w[k]
            viol = max(viol, float(np.maximum(s["dl"] - v, 0.0).max()), float(np.maximum(v - s["du"], 0.0).max()))
        if k < len(st) - 1:
            eq = max(eq, float(np.abs(s["b"] + np.hstack([s["B"], s["A"]]) @ w[k] - w[k + 1][st[k + 1]["nu"]:]).max()))
    return obj, viol, eq


# (the oracle's default profile - usv_oracle.c usv_opts_profile, BALANCE - less the iterative refinement, which the dense numpy solves do not need)
DEFAULT_OPTS = dict(qp_iter_max=50, mu0=1.0, thr0=0.1, tol_stat=1e-6, tol_eq=1e-8, tol_ineq=1e-8, tol_comp=1e-8, alpha_min=1e-8,
                    cond_pred_corr=1, cpc_factor=2.0)
R04_OPTS = dict(qp_iter_max=50, mu0=10.0, thr0=0.1, tol_stat=1e-6, tol_eq=1e-8, tol_ineq=1e-8, tol_comp=1e-8, alpha_min=1e-12, cond_pred_corr=0)


def rti_condensed(binding, spec, x, u, x0, yref, yref_e, p, lh, N2, **opts):
    """One SQP-RTI iteration of one instance whose QP is partially condensed to N2 stages before it is solved.
    Returns dict(x, u, status, qp_iter, qp_status) like binding.rti."""
    o = dict(DEFAULT_OPTS)
    o.update(opts)
    qp, _ = binding.linearize_and_solve(spec, x, u, x0, yref, yref_e, p, lh, solve=False)
    cq = part_cond(qp, N2)
    sol = solve(cq, o)
    dz = expand(cq, sol)
    nu = qp["nu"]
    ok = sol["status"] in (0, 1)
    xn, un = np.array(x, dtype=float).copy(), np.array(u, dtype=float).copy()
    if ok:
        xn += dz[:, nu:]
        un += dz[:-1, :nu]
    return dict(x=xn, u=un, status=0 if ok else 4, qp_iter=sol["iter"], qp_status=sol["status"], res=sol["res"], cq=cq, dz=dz)
