"""ctypes binding of the CPU oracle (oracle/usv_oracle.c).

TEST INFRASTRUCTURE ONLY — imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg, never by the product package (mpc_collisionavoidance_amd).  Parity unpinned: see
oracle/usv_oracle.h.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libusv_oracle.so")

NXM, NUM, NZM, KM, NYM = 14, 2, 16, 32, 16
M0, M1, M2, MGEN = 0, 1, 2, 3
RICCATI_SQRT, RICCATI_CLASSIC = 0, 1

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)


class Opts(C.Structure):
    _fields_ = [("qp_iter_max", C.c_int), ("mu0", C.c_double), ("thr0", C.c_double),
                ("tol_stat", C.c_double), ("tol_eq", C.c_double), ("tol_ineq", C.c_double),
                ("tol_comp", C.c_double), ("alpha_min", C.c_double), ("riccati", C.c_int),
                ("cond_pred_corr", C.c_int), ("cpc_factor", C.c_double), ("itref_corr_max", C.c_int)]


class Spec(C.Structure):
    _fields_ = [("model", C.c_int), ("N", C.c_int), ("dt", C.c_double), ("K", C.c_int),
                ("nx", C.c_int), ("nu", C.c_int), ("ny", C.c_int), ("ny_e", C.c_int),
                ("W", C.c_double * (NYM * NYM)), ("W_e", C.c_double * (NXM * NXM)),
                ("Vx", C.c_double * (NYM * NXM)), ("Vu", C.c_double * (NYM * NUM)),
                ("Vx_e", C.c_double * (NXM * NXM)),
                ("nbu", C.c_int), ("idxbu", C.c_int * NUM), ("lbu", C.c_double * NUM), ("ubu", C.c_double * NUM),
                ("nbx", C.c_int), ("idxbx", C.c_int * NXM), ("lbx", C.c_double * NXM), ("ubx", C.c_double * NXM),
                ("uh", C.c_double * KM), ("soft", C.c_int),
                ("lsh", C.c_double * KM), ("ush", C.c_double * KM),
                ("zl", C.c_double * KM), ("zu", C.c_double * KM), ("Zl", C.c_double * KM), ("Zu", C.c_double * KM),
                ("opts", Opts),
                ("sbx", C.c_int * NXM), ("lsbx", C.c_double * NXM), ("usbx", C.c_double * NXM),
                ("zl_bx", C.c_double * NXM), ("zu_bx", C.c_double * NXM), ("Zl_bx", C.c_double * NXM), ("Zu_bx", C.c_double * NXM),
                ("sim_steps", C.c_int), ("nlp_max_iter", C.c_int), ("nlp_tol", C.c_double * 4)]


class Qp(C.Structure):
    _fields_ = [("N", C.c_int), ("nx", C.c_int), ("nu", C.c_int), ("nz", C.c_int), ("K", C.c_int),
                ("nbu", C.c_int), ("nbx", C.c_int), ("soft", C.c_int),
                ("idxbu", C.c_int * NUM), ("idxbx", C.c_int * NXM), ("ipx", C.c_int), ("ipy", C.c_int),
                ("A", _dp), ("B", _dp), ("b", _dp), ("H", _dp), ("g", _dp), ("dx0", _dp),
                ("lbu", _dp), ("ubu", _dp), ("lbx", _dp), ("ubx", _dp), ("Cxy", _dp), ("lg", _dp), ("ug", _dp),
                ("zl", _dp), ("zu", _dp), ("Zl", _dp), ("Zu", _dp), ("lsl", _dp), ("lsu", _dp),
                ("sbx", C.c_int * NXM), ("scratch", C.c_void_p)]


class QpSol(C.Structure):
    _fields_ = [("dz", _dp), ("pi", _dp), ("lam_bu", _dp), ("t_bu", _dp), ("lam_bx", _dp), ("t_bx", _dp),
                ("lam_g", _dp), ("t_g", _dp), ("sl", _dp), ("su", _dp), ("lam_s", _dp), ("t_s", _dp),
                ("sl_bx", _dp), ("su_bx", _dp), ("lam_sbx", _dp), ("t_sbx", _dp),
                ("iter", C.c_int), ("status", C.c_int), ("res", C.c_double * 4), ("cpc_fallbacks", C.c_int)]


def build(force=False):
    """Compile oracle/libusv_oracle.so with gcc (idempotent)."""
    srcs = [os.path.join(_HERE, f) for f in ("usv_oracle.c", "usv_oracle.h", "usv_guidance_oracle.c",
                                             "usv_guidance_oracle.h", "Makefile")]
    if not force and os.path.exists(_LIB) and os.path.getmtime(_LIB) >= max(os.path.getmtime(f) for f in srcs):
        return _LIB
    subprocess.check_call(["make", "-C", _HERE, "-B", "libusv_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB)
        L.usv_spec_defaults.argtypes = [C.POINTER(Spec), C.c_int, C.c_int, C.c_double, C.c_int]
        L.usv_opts_profile.argtypes = [C.POINTER(Opts), C.c_int]
        L.usv_model_dims.argtypes = [C.c_int, _ip, _ip]
        L.usv_model_f.argtypes = [C.c_int, _dp, _dp, _dp]
        L.usv_model_f.restype = None
        L.usv_model_jac.argtypes = [C.c_int, _dp, _dp, _dp, _dp]
        L.usv_model_jac.restype = None
        L.usv_model_h.argtypes = [C.c_int, C.c_int, _dp, _dp, _dp, _dp]
        L.usv_model_h.restype = None
        L.usv_rk4_sens.argtypes = [C.c_int, C.c_double, _dp, _dp, _dp, _dp, _dp]
        L.usv_rk4_sens.restype = None
        L.usv_qp_alloc.argtypes = [C.POINTER(Spec)]
        L.usv_qp_alloc.restype = C.POINTER(Qp)
        L.usv_qp_free.argtypes = [C.POINTER(Qp)]
        L.usv_qp_free.restype = None
        L.usv_qp_sol_alloc.argtypes = [C.POINTER(Qp)]
        L.usv_qp_sol_alloc.restype = C.POINTER(QpSol)
        L.usv_qp_sol_free.argtypes = [C.POINTER(QpSol)]
        L.usv_qp_sol_free.restype = None
        L.usv_linearize.argtypes = [C.POINTER(Spec)] + [_dp] * 7 + [C.POINTER(Qp)]
        L.usv_linearize.restype = None
        L.usv_qp_solve.argtypes = [C.POINTER(Qp), C.POINTER(Opts), C.POINTER(QpSol)]
        L.usv_rti.argtypes = [C.POINTER(Spec)] + [_dp] * 11
        L.usv_rti_batch.argtypes = [C.POINTER(Spec), C.c_int] + [_dp] * 7 + [_ip, _ip]
        L.usv_erk_sens.argtypes = [C.c_int, C.c_double, C.c_int, _dp, _dp, _dp, _dp, _dp]
        L.usv_erk_sens.restype = None
        L.usv_sqp_batch.argtypes = [C.POINTER(Spec), C.c_int] + [_dp] * 7 + [_ip, _ip, _dp]
        L.usv_rti_batch_mt.argtypes = [C.POINTER(Spec), C.c_int] + [_dp] * 7 + [_ip, _ip, C.c_int]
        L.usv_oracle_register_generated.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
        L.usv_oracle_register_generated.restype = None
        _fp = C.POINTER(C.c_float)
        L.usv_guidance_reset_ref.argtypes = [_dp, C.c_double, _ip, _fp]
        L.usv_guidance_reset_ref.restype = None
        L.usv_guidance_obstacles_ref.argtypes = [C.c_int, C.c_double, C.c_double, C.c_double, _dp, C.c_int, _dp, _dp, _ip]
        L.usv_guidance_obstacles_ref.restype = None
        L.usv_guidance_prepare_ref.argtypes = [C.c_int, _dp, _dp, _dp, C.c_int, _dp, C.c_int, _ip, _fp] + [_dp] * 5
        L.usv_guidance_publish_ref.argtypes = [C.c_double, C.c_double, C.c_double, _fp, _dp, _dp, _dp]
        L.usv_guidance_publish_ref.restype = None
        _lib = L
    return _lib


def _d(a):
    return a.ctypes.data_as(_dp)


def _arr(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def dims(model):
    nx, nu = C.c_int(), C.c_int()
    lib().usv_model_dims(model, C.byref(nx), C.byref(nu))
    return nx.value, nu.value


HPIPM_MODES = {"BALANCE": 0, "SPEED": 1, "ROBUST": 2, "R04": 3}   # usv_oracle.h USV_HPIPM_*


def spec(model, N, Tf, K=0, **opts):
    """hpipm_mode = "BALANCE" (default) | "SPEED" | "ROBUST" | "R04": the QP solver profile (usv_opts_profile), applied before the other
    options, which override single fields of it."""
    s = Spec()
    if lib().usv_spec_defaults(C.byref(s), model, N, float(Tf), K) != 0:
        raise ValueError("bad spec")
    if "hpipm_mode" in opts:
        m = opts.pop("hpipm_mode")
        if lib().usv_opts_profile(C.byref(s.opts), HPIPM_MODES.get(str(m).upper(), m) if not isinstance(m, int) else m) != 0:
            raise ValueError("hpipm_mode must be one of %s" % ", ".join(HPIPM_MODES))
    for k, v in opts.items():
        if k in ("sim_steps", "nlp_max_iter"):
            setattr(s, k, int(v))
        elif k == "soft_bx":   # {position in the bx list: (lsbx, usbx, zl, zu, Zl, Zu)}
            for i, d in v.items():
                s.sbx[i] = 1
                s.lsbx[i], s.usbx[i], s.zl_bx[i], s.zu_bx[i], s.Zl_bx[i], s.Zu_bx[i] = (float(t) for t in d)
        elif k == "nlp_tol":
            for i in range(4):
                s.nlp_tol[i] = float(v[i] if np.ndim(v) else v)
        else:
            setattr(s.opts, k, v)
    return s


def model_f(model, x, u):
    nx, nu = dims(model)
    x, u = _arr(x), _arr(u)
    f = np.zeros(nx)
    lib().usv_model_f(model, _d(x), _d(u), _d(f))
    return f


def model_jac(model, x, u):
    nx, nu = dims(model)
    x, u = _arr(x), _arr(u)
    Jx, Ju = np.zeros((nx, nx)), np.zeros((nx, nu))
    lib().usv_model_jac(model, _d(x), _d(u), _d(Jx), _d(Ju))
    return Jx, Ju


def model_h(model, x, p):
    x, p = _arr(x), _arr(p)
    K = p.size // 2
    h, Cxy = np.zeros(K), np.zeros((K, 2))
    lib().usv_model_h(model, K, _d(x), _d(p), _d(h), _d(Cxy))
    return h, Cxy


def rk4_sens(model, dt, x, u):
    nx, nu = dims(model)
    x, u = _arr(x), _arr(u)
    xn, A, B = np.zeros(nx), np.zeros((nx, nx)), np.zeros((nx, nu))
    lib().usv_rk4_sens(model, float(dt), _d(x), _d(u), _d(xn), _d(A), _d(B))
    return xn, A, B


def _np_from(ptr, shape):
    n = int(np.prod(shape))
    if n == 0:
        return np.zeros(shape)
    return np.ctypeslib.as_array(ptr, shape=(n,)).reshape(shape).copy()


def linearize_and_solve(s, x, u, x0, yref, yref_e, p, lh, solve=True):
    """Returns (qp dict, sol dict) of one RTI iteration's QP, for independent checking."""
    L = lib()
    x, u, x0, yref, yref_e, p, lh = map(_arr, (x, u, x0, yref, yref_e, p, lh))
    q = L.usv_qp_alloc(C.byref(s))
    L.usv_linearize(C.byref(s), _d(x), _d(u), _d(x0), _d(yref), _d(yref_e), _d(p), _d(lh), q)
    qc = q.contents
    N, nx, nu, nz, K = qc.N, qc.nx, qc.nu, qc.nz, qc.K
    nbu, nbx = qc.nbu, qc.nbx
    qp = dict(N=N, nx=nx, nu=nu, nz=nz, K=K, nbu=nbu, nbx=nbx, soft=qc.soft,
              idxbu=list(qc.idxbu[:nbu]), idxbx=list(qc.idxbx[:nbx]), ipx=qc.ipx, ipy=qc.ipy,
              A=_np_from(qc.A, (N, nx, nx)), B=_np_from(qc.B, (N, nx, nu)), b=_np_from(qc.b, (N, nx)),
              H=_np_from(qc.H, (N + 1, nz, nz)), g=_np_from(qc.g, (N + 1, nz)), dx0=_np_from(qc.dx0, (nx,)),
              lbu=_np_from(qc.lbu, (N, nbu)), ubu=_np_from(qc.ubu, (N, nbu)),
              lbx=_np_from(qc.lbx, (N + 1, nbx)), ubx=_np_from(qc.ubx, (N + 1, nbx)),
              Cxy=_np_from(qc.Cxy, (N + 1, K, 2)), lg=_np_from(qc.lg, (N + 1, K)), ug=_np_from(qc.ug, (N + 1, K)),
              zl=_np_from(qc.zl, (K + nbx,)), zu=_np_from(qc.zu, (K + nbx,)), Zl=_np_from(qc.Zl, (K + nbx,)), Zu=_np_from(qc.Zu, (K + nbx,)),
              lsl=_np_from(qc.lsl, (K + nbx,)), lsu=_np_from(qc.lsu, (K + nbx,)), sbx=list(qc.sbx[:nbx]))
    sol = None
    if solve:
        sp = L.usv_qp_sol_alloc(q)
        L.usv_qp_solve(q, C.byref(s.opts), sp)
        sc = sp.contents
        sol = dict(dz=_np_from(sc.dz, (N + 1, nz)), pi=_np_from(sc.pi, (N + 1, nx)),
                   lam_bu=_np_from(sc.lam_bu, (N, 2, nbu)), t_bu=_np_from(sc.t_bu, (N, 2, nbu)),
                   lam_bx=_np_from(sc.lam_bx, (N + 1, 2, nbx)), t_bx=_np_from(sc.t_bx, (N + 1, 2, nbx)),
                   lam_g=_np_from(sc.lam_g, (N + 1, 2, K)), t_g=_np_from(sc.t_g, (N + 1, 2, K)),
                   sl=_np_from(sc.sl, (N + 1, K)), su=_np_from(sc.su, (N + 1, K)),
                   lam_s=_np_from(sc.lam_s, (N + 1, 2, K)), t_s=_np_from(sc.t_s, (N + 1, 2, K)),
                   sl_bx=_np_from(sc.sl_bx, (N + 1, nbx)), su_bx=_np_from(sc.su_bx, (N + 1, nbx)),
                   lam_sbx=_np_from(sc.lam_sbx, (N + 1, 2, nbx)), t_sbx=_np_from(sc.t_sbx, (N + 1, 2, nbx)),
                   iter=sc.iter, status=sc.status, res=np.array(sc.res[:]), cpc_fallbacks=sc.cpc_fallbacks)
        L.usv_qp_sol_free(sp)
    L.usv_qp_free(q)
    return qp, sol


def rti(s, x, u, x0, yref, yref_e, p, lh):
    """One SQP-RTI iteration. x,u are updated copies. Returns dict(x,u,sl,su,pi,status,info)."""
    N, nx, nu, K = s.N, s.nx, s.nu, s.K
    x, u = _arr(x).copy(), _arr(u).copy()
    x0, yref, yref_e, p, lh = map(_arr, (x0, yref, yref_e, p, lh))
    sl, su = np.zeros((N, max(K, 1))), np.zeros((N, max(K, 1)))
    pi, info = np.zeros((N, nx)), np.zeros(8)
    st = lib().usv_rti(C.byref(s), _d(x), _d(u), _d(x0), _d(yref), _d(yref_e), _d(p), _d(lh),
                       _d(sl), _d(su), _d(pi), _d(info))
    return dict(x=x.reshape(N + 1, nx), u=u.reshape(N, nu), sl=sl[:, :K], su=su[:, :K], pi=pi,
                status=st, qp_iter=int(info[0]), qp_status=int(info[1]), res=info[2:6].copy())


_native = None


def native_lib():
    """The same sources compiled for THIS machine's CPU with full optimisation (-O3 -march=native, FMA contraction
    allowed): used only to TIME the CPU baseline in bench.py, so that the baseline is not handicapped by the
    bit-stable flags the checker is built with.  Built on first use (never shipped: the build host's CPU is not the
    GPU box's).  Returns None when it cannot be built; results then come from the checker library."""
    global _native
    if _native is None:
        out = os.path.join(_HERE, "_native", "libusv_oracle_native.so")
        try:
            os.makedirs(os.path.dirname(out), exist_ok=True)
            subprocess.check_call(["gcc", "-O3", "-march=native", "-fPIC", "-std=c99", "-fopenmp", "-shared", "-o", out,
                                   os.path.join(_HERE, "usv_oracle.c"), os.path.join(_HERE, "usv_guidance_oracle.c"), "-lm"],
                                  stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            L = C.CDLL(out)
            L.usv_rti_batch.argtypes = [C.POINTER(Spec), C.c_int] + [_dp] * 7 + [_ip, _ip]
            L.usv_rti_batch_mt.argtypes = [C.POINTER(Spec), C.c_int] + [_dp] * 7 + [_ip, _ip, C.c_int]
            _native = L
        except Exception:
            _native = False
    return _native or None


def rti_batch(s, x, u, x0, yref, yref_e, p, lh, threads=1, native=False):
    """In-place batched RTI over leading batch axis (row-major per instance); threads > 1 (or 0 = all cores)
    distributes the instances over OpenMP threads."""
    B = x.shape[0]
    for a in (x, u):
        assert a.dtype == np.float64 and a.flags.c_contiguous
    x0, yref, yref_e, p, lh = map(_arr, (x0, yref, yref_e, p, lh))
    status = np.zeros(B, dtype=np.int32)
    it = np.zeros(B, dtype=np.int32)
    L = (native_lib() if native else None) or lib()
    if threads == 1:
        L.usv_rti_batch(C.byref(s), B, _d(x), _d(u), _d(x0), _d(yref), _d(yref_e), _d(p), _d(lh),
                        status.ctypes.data_as(_ip), it.ctypes.data_as(_ip))
    else:
        L.usv_rti_batch_mt(C.byref(s), B, _d(x), _d(u), _d(x0), _d(yref), _d(yref_e), _d(p), _d(lh),
                               status.ctypes.data_as(_ip), it.ctypes.data_as(_ip), int(threads))
    return status, it


def erk_sens(model, dt, steps, x, u):
    nx, nu = dims(model)
    x, u = _arr(x), _arr(u if nu else np.zeros(1))
    xn, A, B = np.zeros(nx), np.zeros((nx, nx)), np.zeros((nx, max(nu, 1)))
    lib().usv_erk_sens(model, float(dt), int(steps), _d(x), _d(u), _d(xn), _d(A), _d(B))
    return xn, A, B[:, :nu]


def sqp_batch(s, x, u, x0, yref, yref_e, p, lh):
    """In-place batched full SQP; returns (status, sqp_iter, nlp_res[B,4])."""
    B = x.shape[0]
    for a in (x, u):
        assert a.dtype == np.float64 and a.flags.c_contiguous
    x0, yref, yref_e, p, lh = map(_arr, (x0, yref, yref_e, p, lh))
    status, it, res = np.zeros(B, dtype=np.int32), np.zeros(B, dtype=np.int32), np.zeros((B, 4))
    lib().usv_sqp_batch(C.byref(s), B, _d(x), _d(u), _d(x0), _d(yref), _d(yref_e), _d(p), _d(lh),
                        status.ctypes.data_as(_ip), it.ctypes.data_as(_ip), _d(res))
    return status, it, res


# ---- reference ROS-node arithmetic either side of the solve (usv_guidance_oracle.c)
def guidance_reset(waypoints, psi):
    w = _arr(waypoints)
    k, pp = C.c_int(), C.c_float()
    lib().usv_guidance_reset_ref(_d(w), float(psi), C.byref(k), C.byref(pp))
    return k.value, pp.value


def guidance_obstacles(K, psi, nedx, nedy, obs):
    obs = _arr(obs).reshape(-1, 3)
    p, r, ch = np.zeros(2 * K), np.zeros(K), np.zeros(K, dtype=np.int32)
    lib().usv_guidance_obstacles_ref(K, float(psi), float(nedx), float(nedy), _d(obs), obs.shape[0], _d(p), _d(r),
                                     ch.ctypes.data_as(_ip))
    return p, r, ch


def guidance_prepare(K, vel_uv, pose, waypoints, obs, k, past_psied):
    """Returns dict(active, k, past_psied, x0, p_obs, r_obs, ak, ye) for one instance."""
    vel_uv, pose, w = _arr(vel_uv), _arr(pose), _arr(waypoints)
    obs = _arr(obs).reshape(-1, 3)
    kk, pp = C.c_int(int(k)), C.c_float(float(past_psied))
    x0, p, r, ak, ye = np.zeros(8), np.zeros(2 * K), np.zeros(K), np.zeros(1), np.zeros(1)
    act = lib().usv_guidance_prepare_ref(K, _d(vel_uv), _d(pose), _d(w), w.size // 2, _d(obs), obs.shape[0],
                                         C.byref(kk), C.byref(pp), _d(x0), _d(p), _d(r), _d(ak), _d(ye))
    return dict(active=act, k=kk.value, past_psied=pp.value, x0=x0, p_obs=p, r_obs=r, ak=ak[0], ye=ye[0])


def obstacle_sim(pose, world, max_radius=100.0, lmax=64):
    """obstacle_sim_node.simulate(): (obstacles [n,3] body frame, n)."""
    pose, world = _arr(pose), _arr(np.asarray(world, dtype=float).reshape(-1, 3))
    out = np.zeros((lmax, 3))
    L = lib()
    L.usv_obstacle_sim_ref.argtypes = [_dp, _dp, C.c_int, C.c_double, C.c_int, _dp]
    n = L.usv_obstacle_sim_ref(_d(pose), _d(world), world.shape[0], float(max_radius), lmax, _d(out))
    return out[:n].copy(), n


def guidance_publish(x1_psied, u0, ak, past_psied):
    pp = C.c_float(float(past_psied))
    h, r, s = C.c_double(), C.c_double(), C.c_double()
    lib().usv_guidance_publish_ref(float(x1_psied), float(u0), float(ak), C.byref(pp), C.byref(h), C.byref(r), C.byref(s))
    return dict(heading=h.value, r=r.value, speed=s.value, past_psied=pp.value)


# ---- generated models (codegen.emit_oracle_c) and generic OCP -> Spec translation
_gen_keep = []


def register_generated(c_source, workdir):
    """Compile the generated plain-C model, register it as model id MGEN. Returns (nx, nu)."""
    import hashlib
    tag = hashlib.sha256(c_source.encode()).hexdigest()[:12]
    src = os.path.join(workdir, "gen_%s.c" % tag)
    so = os.path.join(workdir, "gen_%s.so" % tag)
    if not os.path.exists(so):
        with open(src, "w") as f:
            f.write(c_source)
        subprocess.check_call(["gcc", "-O2", "-fPIC", "-shared", "-ffp-contract=off", "-o", so, src, "-lm"])
    G = C.CDLL(so)
    _gen_keep.append(G)
    fn = C.cast(G.usv_gen_fjvp, C.c_void_p)
    nx, nu = G.usv_gen_nx(), G.usv_gen_nu()
    lib().usv_oracle_register_generated(fn, nx, nu, G.usv_gen_ipx(), G.usv_gen_ipy())
    return nx, nu


def spec_from_ocp(ocp, model_id, **opts):
    """Spec for any model id from an AcadosOcp look-alike (weights, selectors, bounds, soft data)."""
    N, Tf = int(ocp.dims.N), float(ocp.solver_options.tf)
    K = 0 if ocp.model.con_h_expr is None else ocp.model.con_h_expr.size()[0]
    s = spec(model_id, N, Tf, K, **opts)
    nx, nu = s.nx, s.nu
    ny = nx + nu
    for name, arr, n in (("W", ocp.cost.W, ny * ny), ("W_e", ocp.cost.W_e, nx * nx), ("Vx", ocp.cost.Vx, ny * nx),
                         ("Vu", ocp.cost.Vu, ny * nu), ("Vx_e", ocp.cost.Vx_e, nx * nx)):
        a = np.ascontiguousarray(arr, dtype=np.float64).reshape(-1)
        assert a.size == n, name
        dst = getattr(s, name)
        for i in range(dst._length_):
            dst[i] = a[i] if i < n else 0.0
    con, cost = ocp.constraints, ocp.cost
    idxbu = np.asarray(con.idxbu, dtype=int).reshape(-1)
    s.nbu = idxbu.size
    for i in range(s.nbu):
        s.idxbu[i], s.lbu[i], s.ubu[i] = int(idxbu[i]), float(con.lbu[i]), float(con.ubu[i])
    idxbx = np.asarray(con.idxbx, dtype=int).reshape(-1)
    s.nbx = idxbx.size
    for i in range(s.nbx):
        s.idxbx[i], s.lbx[i], s.ubx[i] = int(idxbx[i]), float(con.lbx[i]), float(con.ubx[i])
    s.soft = 1 if np.asarray(con.idxsh).size else 0
    idxsbx = np.asarray(getattr(con, "idxsbx", []), dtype=int).reshape(-1)
    nsbx = idxsbx.size
    for j, i in enumerate(idxsbx):   # acados orders the slack penalties [sbx.., sh..]
        s.sbx[i] = 1
        s.lsbx[i], s.usbx[i] = float(con.lsbx[j]), float(con.usbx[j])
        s.zl_bx[i], s.zu_bx[i], s.Zl_bx[i], s.Zu_bx[i] = float(cost.zl[j]), float(cost.zu[j]), float(cost.Zl[j]), float(cost.Zu[j])
    for i in range(K):
        s.uh[i] = float(con.uh[i])
        if s.soft:
            s.lsh[i], s.ush[i] = float(con.lsh[i]), float(con.ush[i])
            s.zl[i], s.zu[i], s.Zl[i], s.Zu[i] = (float(cost.zl[nsbx + i]), float(cost.zu[nsbx + i]),
                                                  float(cost.Zl[nsbx + i]), float(cost.Zu[nsbx + i]))
    return s
