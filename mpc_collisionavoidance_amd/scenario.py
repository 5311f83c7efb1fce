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

# Benchmark settings: shooting interval 0.05 s for EVERY model, i.e. Tf = 2 s at N = 40.
# usv_model_pf_ca cannot take one RK4 step of 0.05 s (see above), so its integrator runs 5 steps per interval
# (acados sim_method_num_steps = 5): the integration step stays the reference's 0.01 s while the look-ahead is
# the 2 s the obstacle rows need to become active.  NOISE_MASK: states the closed-loop disturbance is added to -
# every state for the soft-row model, (u, r) for usv_model_pf_ca as the reference's own "Add noise" hooks have it
# (scripts/usv_pf_ca/main.py:181-183: x0[3], x0[5]; position noise on a vehicle that grazes a HARD keep-out
# circle makes the next QP infeasible by construction).
BENCH_DT = 0.05
BENCH_SIM_STEPS = {"usv_model": 1, "usv_model_guidance_ca1": 1, "usv_model_pf_ca": 5}
NOISE_MASK = {"usv_model": (1 << 5) - 1, "usv_model_guidance_ca1": (1 << 8) - 1, "usv_model_pf_ca": (1 << 3) | (1 << 5)}

ALL_STATES_MASK = (1 << 14) - 1   # SURVEY.md 8(d) as worded: "x0 <- previous x1 + N(0, 1e-3)" on every state

_CY = 0.5 * (-40.0 * 1000.0) * (1.1 + 0.0045 * (1.01 / 0.09) - 0.1 * (0.27 / 0.09) + 0.016 * ((0.27 / 0.09) ** 2))


def _pf_ca_rhs(y, ak, Tp, Ts):
    """usv_model_pf_ca right-hand side with zero input (thrusts constant), vectorised over the batch
    (scripts/usv_pf_ca/usv_model.py:137-160).  y = rows (psi, sinpsi, cospsi, u, v, r, ye, nedx, nedy) of shape [9, B]:
    the states that move; only used to roll the initial guess forward."""
    m, Iz, Bw = 30.0, 4.1, 0.41
    Xud, Yvd, Yrd, Nvd, Nrd = -2.25, -23.13, -1.31, -16.41, -2.79
    Yvv, Yvr, Nrv, Nrr = -99.99, -5.49, -8.8, -3.49
    psi, u, v, r = y[0], y[3], y[4], y[5]
    fast = u > 1.25
    Xu, Xuu = np.where(fast, 64.55, -25.0), np.where(fast, -70.92, 0.0)
    au, av, ar = np.abs(u), np.abs(v), np.abs(r)
    Nr = -0.52 * np.sqrt(u * u + v * v)
    a = Yrd + Nvd
    f = np.empty_like(y)
    chi = psi + np.arctan2(v, u + 0.001)
    sp, cp = np.sin(psi), np.cos(psi)
    vx, vy = u * cp - v * sp, u * sp + v * cp
    f[0] = r
    f[1] = np.cos(chi) * r
    f[2] = -np.sin(chi) * r
    f[3] = ((Tp + Ts) - (-m + 2.0 * Yvd) * v - a * r * r - (-Xu * u - Xuu * au * u)) / (m - Xud)
    f[4] = (-(m - Xud) * u * r - (-_CY * av - Yvv * av - Yvr * ar) * v) / (m - Yvd)
    f[5] = ((Tp - Ts) * (Bw / 2.0) - (-2.0 * Yvd * u * v - a * r * u + Xud * u * r) -
            (-Nr * r - Nrv * av * r - Nrr * ar * r)) / (Iz - Nrd)
    f[6] = -vx * np.sin(ak) + vy * np.cos(ak)
    f[7] = vx
    f[8] = vy
    return f


def _pf_ca_rollout(x0, N, dt, steps):
    """Zero-input simulation with the solver's own integrator (RK4, `steps` steps per interval): [B, N+1, nx]."""
    h = dt / steps
    mov = [0, 1, 2, 3, 4, 5, 6, 10, 11]
    y = np.ascontiguousarray(x0[:, mov].T)
    ak, Tp, Ts = x0[:, 9].copy(), x0[:, 12].copy(), x0[:, 13].copy()
    out = np.repeat(x0[:, None, :], N + 1, axis=1)
    for k in range(N):
        for _ in range(steps):
            k1 = _pf_ca_rhs(y, ak, Tp, Ts)
            k2 = _pf_ca_rhs(y + 0.5 * h * k1, ak, Tp, Ts)
            k3 = _pf_ca_rhs(y + 0.5 * h * k2, ak, Tp, Ts)
            k4 = _pf_ca_rhs(y + h * k3, ak, Tp, Ts)
            y = y + (h / 6.0) * (k1 + 2.0 * k2 + 2.0 * k3 + k4)
        out[:, k + 1, mov] = y.T
    return out


def make_batch(name, N, K, B, dt=None, seed=1234, moving=False, n_active=None, generator="beside", sim_steps=1,
               max_range=6.0, clip_time=1.2):
    """Returns dict(x0 [B,nx], yref [B,N,ny], yref_e [B,nx], p [B,N+1,2K], lh [B,N,K],
    x_init [B,N+1,nx], u_init [B,N,nu]).

    generator (obstacle placement and initial guess of usv_model_pf_ca; the other models have one):
      "survey" - SURVEY.md 8(d): sway v ~ U(-0.1, 0.1); obstacles in polar form about the vehicle (bearing within
                 +-60 deg of the course, range R+0.7 .. 6 m), with ONE stated clip: an obstacle whose keep-out circle
                 the course ray would enter closer than 0.4 m + 1.2 s * u is moved outwards along its bearing until it
                 does not (hard rows: nearer than that the first QP has no feasible point - the vehicle can neither
                 stop nor turn in time); initial guess = zero-input simulation with the solver's integrator.
      "survey_verbatim" - SURVEY.md 8(d) to the letter: the same draws as "survey" (same seed -> same vehicles, same obstacles before
                 the clip), NO clip, and acados' own initial guess x_k = x0, u = 0.  For usv_model_pf_ca's hard rows that means a
                 share of first QPs without a feasible point (the line reports them: status_nonzero_frac, qp_not_converged_frac).
      "beside" - round-1 workload for the reference's dt = 0.01 s (0.4 s look-ahead at N = 40): v ~ U(-0.03, 0.03),
                 obstacles beside the straight-line roll-out; kept for the parity tests at the reference's step size."""
    if dt is None:
        dt = DT[name]
    if generator not in ("survey", "survey_verbatim", "beside"):
        raise ValueError(generator)
    verbatim = generator == "survey_verbatim"
    survey = generator in ("survey", "survey_verbatim") and name == "usv_model_pf_ca"
    rng = np.random.default_rng(seed)
    x1, y1, x2, y2 = PATH
    ak = np.arctan2(y2 - y1, x2 - x1)
    nedx = rng.uniform(2.0, 6.0, B)
    nedy = rng.uniform(-5.0, 15.0, B)
    psi = ak + rng.uniform(-0.6, 0.6, B)
    u = rng.uniform(0.3, 1.2, B)
    v = rng.uniform(-0.1, 0.1, B) if (name == "usv_model_guidance_ca1" or survey) else rng.uniform(-0.03, 0.03, B)
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
        if name == "usv_model_guidance_ca1" or survey:
            # polar about the vehicle: bearing within +-60 deg of the course, range R+0.7 .. 6 m
            rad = (R + 0.7) + rng.uniform(0.0, 1.0, (B, K)) * np.maximum(max_range - (R + 0.7), 0.0)
            rel = rng.uniform(-np.pi / 3, np.pi / 3, (B, K))
            if survey and not verbatim:
                # hard rows: the course ray enters the keep-out circle (radius lh) at s = lon - sqrt(lh^2 - lat^2);
                # obstacles with s < s_min(u) move out along their bearing to the range where s = s_min
                smin = (0.4 + clip_time * u)[:, None]
                lat, lon = rad * np.sin(rel), rad * np.cos(rel)
                hit = np.abs(lat) < lhv
                s_enter = np.where(hit, lon - np.sqrt(np.maximum(lhv ** 2 - lat ** 2, 0.0)), np.inf)
                cb = np.cos(rel)
                # s_enter grows with the range; it equals s_min at the larger root of the quadratic below PROVIDED that
                # root still has the centre beyond s_min (lon >= s_min) - otherwise the chord the ray cuts vanishes
                # first, and the obstacle moves out to where the ray is tangent to its circle
                disc = (smin * cb) ** 2 - (smin ** 2 - lhv ** 2)
                root = smin * cb + np.sqrt(np.maximum(disc, 0.0))
                tangent = lhv / np.maximum(np.abs(np.sin(rel)), 1e-12)
                rad_out = np.where((disc >= 0.0) & (root * cb >= smin), root, tangent)
                rad = np.where(hit & (s_enter < smin), np.maximum(rad, rad_out) + 1e-9, rad)
            bearing = course[:, None] + rel
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
        elif name == "usv_model_guidance_ca1" or survey:
            vel = rng.uniform(-0.3, 0.3, (B, K, 2))
            if survey:
                # hard rows: an obstacle that closes in on the vehicle cannot be answered by a slack; the component of its
                # velocity towards the vehicle's position is removed (it may still cross the path ahead)
                er = np.stack([ox - nedx[:, None], oy - nedy[:, None]], axis=2)
                er /= np.linalg.norm(er, axis=2, keepdims=True)
                vr = (vel * er).sum(axis=2, keepdims=True)
                vel = vel - np.minimum(vr, 0.0) * er
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
    if verbatim:
        pass  # x_k = x0: what AcadosOcpSolver starts from when the caller sets nothing (scripts/usv_pf_ca/main.py sets no iterate)
    elif name == "usv_model_guidance_ca1":
        x_init[:, :, 2] += tk * yed
        x_init[:, :, 5] += tk * vx
        x_init[:, :, 6] += tk * vy
    elif survey:
        x_init = _pf_ca_rollout(x0, N, dt, sim_steps)
    elif name == "usv_model_pf_ca":
        x_init[:, :, 6] += tk * yed
        x_init[:, :, 10] += tk * vx
        x_init[:, :, 11] += tk * vy
    u_init = np.zeros((B, N, nu))
    return dict(x0=x0, yref=yref, yref_e=yref_e, p=p, lh=lh, x_init=x_init, u_init=u_init,
                nx=nx, nu=nu, K=K, N=N, dt=dt, sim_steps=sim_steps, generator=generator)


def make_bench_batch(name, N, K, B, seed=1234, moving=False, verbatim=False, n_active=None):
    """The benchmark workload of SURVEY.md 8(d) for `name`: dt = 0.05 s, the "survey" generator.  SURVEY's obstacle field
    (range up to 6 m, course ray clear for 0.4 m + 1.2 s * u) is sized for the 2 s look-ahead of N = 40; other horizons scale
    both with it.  Shorter (BASELINE configs[1], Tf = 1 s): range up to 3 Tf metres, clear for 0.6 Tf - with the 2 s field a 1 s
    look-ahead never reaches an obstacle (2 % of the instances with an active row after 12 closed-loop ticks; scaled: 71 %, no
    failed solve).  Longer (configs[4], Tf = 4 s): range up to 3 Tf, clear for 1.1 Tf - at N = 80 the same circles in the same
    6 m sector wall the vehicle in and a third of the hard-row QPs have no feasible point.
    verbatim: SURVEY.md 8(d) without the builder's departures (make_batch, "survey_verbatim"): no obstacle clip, x_k = x0 as the initial
    guess; the disturbance then goes on every state (ALL_STATES_MASK); the 5 RK4 steps of usv_model_pf_ca stay - one RK4 step of 0.05 s
    is outside the model's stability region (DT above), the solves would be NaN."""
    Tf = N * BENCH_DT
    long_h = Tf > 2.0 + 1e-9
    return make_batch(name, N, K, B, dt=BENCH_DT, seed=seed, moving=moving, generator="survey_verbatim" if verbatim else "survey",
                      n_active=n_active,   # (fewer obstacles than slots: the rest parked at (1000, 1000), r = 0 - nmpc_guidance_ca1.cpp:365-376)
                      sim_steps=BENCH_SIM_STEPS[name], max_range=3.0 * Tf,
                      clip_time=1.1 * Tf if long_h else 0.6 * Tf)


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
