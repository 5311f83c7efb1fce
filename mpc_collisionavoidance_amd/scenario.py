"""Synthetic batch workloads: LiDAR obstacle-scenario sweeps about the
reference's path-following scenario.

Inputs are derived the way the reference's callers derive them (paths relative to
/root/reference/catkin_ws/src/nmpc_ca/): cross-track error and course error as
src/nmpc_guidance_ca1.cpp:460-461,495-511; the pf_ca state and yref as
scripts/usv_pf_ca/main.py:95-133; obstacle radius = R + boat radius 0.5
(src/nmpc_guidance_ca1.cpp:139,319), lh = R (soft model, +0.2 through lsh) or R + 0.2 (hard model,
scripts/usv_pf_ca/main.py:126); unused slots at (1000,1000) with r = 0
(src/nmpc_guidance_ca1.cpp:365-376).  Pure numpy; no solver code here.
"""
import numpy as np

PATH = (4.0, -5.0, 4.0, 25.0)  # (x1, y1) -> (x2, y2), scripts/usv_guidance_ca1/main.py:99-102


def wrap(a):
    return (a + np.pi) % (2.0 * np.pi) - np.pi


# Shooting interval per model.  usv_model / usv_model_guidance_ca1: 0.05 s as the reference
# (scripts/usv_acados/main.py:52-53, scripts/usv_guidance_ca1/main.py:54-55).  usv_model_pf_ca:
# 0.01 s as the reference (scripts/usv_pf_ca/main.py:54-55) - its sway damping Yv = -19890|v| is
# stiff and the explicit RK4 map is unstable at 0.05 s once |v| > 0.07.
DT = {"usv_model": 0.05, "usv_model_guidance_ca1": 0.05, "usv_model_pf_ca": 0.01}


def make_batch(name, N, K, B, dt=None, seed=1234, moving=False, n_active=None):
    """Returns dict(x0 [B,nx], yref [B,N,ny], yref_e [B,nx], p [B,N+1,2K], lh [B,N,K],
    x_init [B,N+1,nx], u_init [B,N,nu])."""
    if dt is None:
        dt = DT[name]
    rng = np.random.default_rng(seed)
    x1, y1, x2, y2 = PATH
    ak = np.arctan2(y2 - y1, x2 - x1)
    nedx = rng.uniform(2.0, 6.0, B)
    nedy = rng.uniform(-5.0, 15.0, B)
    psi = ak + rng.uniform(-0.6, 0.6, B)
    u = rng.uniform(0.3, 1.2, B)
    v = rng.uniform(-0.1, 0.1, B) if name == "usv_model_guidance_ca1" else rng.uniform(-0.03, 0.03, B)
    r = rng.uniform(-0.1, 0.1, B)
    Tp = rng.uniform(0.0, 15.0, B)
    Ts = rng.uniform(0.0, 15.0, B)
    ye = -(nedx - x1) * np.sin(ak) + (nedy - y1) * np.cos(ak)
    beta = np.arctan2(v, u + 0.001)
    course = psi + beta
    if name == "usv_model":
        nx, nu = 5, 2
        x0 = np.stack([u, v, r, Tp, Ts], axis=1)
        yr = np.zeros(nx + nu)
        yr[0] = 1.3  # scripts/usv_acados/main.py:73
        yref = np.tile(yr, (B, N, 1))
        yref_e = np.tile(yr[:nx], (B, 1))
    elif name == "usv_model_guidance_ca1":
        nx, nu = 8, 1
        chie = wrap(psi + beta - ak)
        psied = wrap(psi - ak)
        x0 = np.stack([u, v, ye, chie, psied, nedx, nedy, psi], axis=1)
        yref = np.zeros((B, N, nx + nu))
        yref_e = np.zeros((B, nx))
    elif name == "usv_model_pf_ca":
        nx, nu = 14, 2
        x0 = np.stack([psi, np.sin(psi), np.cos(psi), u, v, r, ye, np.full(B, x1), np.full(B, y1),
                       np.full(B, ak), nedx, nedy, Tp, Ts], axis=1)
        yr = np.zeros(nx + nu)
        yr[1], yr[2], yr[3] = np.sin(ak), np.cos(ak), 0.7
        yref = np.tile(yr, (B, N, 1))
        yref_e = np.tile(yr[:nx], (B, 1))
    else:
        raise ValueError(name)
    K = int(K)
    p = np.zeros((B, N + 1, 2 * K))
    lh = np.zeros((B, N, K))
    if K:
        na = K if n_active is None else int(n_active)
        R = rng.uniform(0.3, 1.5, (B, K)) + 0.5
        margin = 0.0 if name == "usv_model_guidance_ca1" else 0.2
        lhv = R + margin
        if name == "usv_model_guidance_ca1":
            # polar about the vehicle: bearing within +-60 deg of the course, range R+0.7 .. 6 m
            rad = (R + 0.7) + rng.uniform(0.0, 1.0, (B, K)) * np.maximum(6.0 - (R + 0.7), 0.0)
            bearing = course[:, None] + rng.uniform(-np.pi / 3, np.pi / 3, (B, K))
            ox = nedx[:, None] + rad * np.cos(bearing)
            oy = nedy[:, None] + rad * np.sin(bearing)
        else:
            # hard rows, short look-ahead L = u*N*dt: obstacles beside the straight-line prediction,
            # keep-out circle 0.01 .. 0.51 m clear of it (quadratic skew towards grazing), abeam of a point 0 .. 1.5 L ahead
            L = (u * N * dt)[:, None]
            s_al = rng.uniform(0.0, 1.5, (B, K)) * L
            clear = 0.01 + 0.5 * rng.uniform(0.0, 1.0, (B, K)) ** 2
            side = np.where(rng.uniform(0.0, 1.0, (B, K)) < 0.5, -1.0, 1.0)
            cx_, cy_ = np.cos(course)[:, None], np.sin(course)[:, None]
            off = side * (lhv + clear)
            ox = nedx[:, None] + s_al * cx_ - off * cy_
            oy = nedy[:, None] + s_al * cy_ + off * cx_
        if na < K:
            ox[:, na:], oy[:, na:], lhv[:, na:] = 1000.0, 1000.0, 0.0
        if not moving:
            vel = np.zeros((B, K, 2))
        elif name == "usv_model_guidance_ca1":
            vel = rng.uniform(-0.3, 0.3, (B, K, 2))
        else:
            # hard rows: obstacles slide parallel to the course, which keeps their lateral clearance
            # (a drift towards the path would make the hard-constrained QPs infeasible)
            vpar = rng.uniform(-0.3, 0.3, (B, K))
            vel = np.stack([vpar * np.cos(course)[:, None], vpar * np.sin(course)[:, None]], axis=2)
        t = (np.arange(N + 1) * dt)[None, :, None]
        p[:, :, 0::2] = ox[:, None, :] + t * vel[:, None, :, 0]
        p[:, :, 1::2] = oy[:, None, :] + t * vel[:, None, :, 1]
        lh[:] = lhv[:, None, :]
    # initial guess: kinematic straight-line rollout (constant body velocities, u = 0).  acados'
    # own cold start x_k = x0 linearises every stage's obstacle rows at the current position,
    # which makes the hard rows of usv_model_pf_ca mutually inconsistent with moving at all.
    x_init = np.tile(x0[:, None, :], (1, N + 1, 1))
    tk = (np.arange(N + 1) * dt)[None, :]
    vx = (u * np.cos(psi) - v * np.sin(psi))[:, None]
    vy = (u * np.sin(psi) + v * np.cos(psi))[:, None]
    yed = -vx * np.sin(ak) + vy * np.cos(ak)
    if name == "usv_model_guidance_ca1":
        x_init[:, :, 2] += tk * yed
        x_init[:, :, 5] += tk * vx
        x_init[:, :, 6] += tk * vy
    elif name == "usv_model_pf_ca":
        x_init[:, :, 6] += tk * yed
        x_init[:, :, 10] += tk * vx
        x_init[:, :, 11] += tk * vy
    u_init = np.zeros((B, N, nu))
    return dict(x0=x0, yref=yref, yref_e=yref_e, p=p, lh=lh, x_init=x_init, u_init=u_init,
                nx=nx, nu=nu, K=K, N=N, dt=dt)


def load_into(solver, wl):
    """Push a workload into a BatchOcpSolver (iterate initialised to x_k = x0, u = 0)."""
    solver.set("x0", 0, wl["x0"])
    solver.set_all("x", wl["x_init"])
    solver.set_all("u", wl["u_init"])
    solver.set_all("yref", wl["yref"])
    solver.set("yref", solver.N, wl["yref_e"])
    if wl["K"]:
        solver.set_all("p", wl["p"])
        solver.set_all("lh", wl["lh"])
