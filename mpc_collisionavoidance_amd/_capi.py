"""ctypes binding of libusvmpc.so (include/usvmpc.h).

The product path has no CPU fallback: if the HIP library is missing or no device is present the
import / create call raises.
"""
import ctypes as C
import os
import sys

import numpy as np

NXM, NUM, NYM, KM = 14, 2, 16, 32
MODEL_IDS = {"usv_model": 0, "usv_model_guidance_ca1": 1, "usv_model_pf_ca": 2}
MODEL_DIMS = {0: (5, 2), 1: (8, 1), 2: (14, 2)}

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)


class Desc(C.Structure):
    """usvmpc_desc"""
    _fields_ = [("model", C.c_int), ("N", C.c_int), ("Tf", C.c_double), ("K", C.c_int),
                ("batch", C.c_int), ("device", C.c_int),
                ("W", C.c_double * (NYM * NYM)), ("W_e", C.c_double * (NXM * NXM)),
                ("Vx", C.c_double * (NYM * NXM)), ("Vu", C.c_double * (NYM * NUM)),
                ("Vx_e", C.c_double * (NXM * NXM)),
                ("nbu", C.c_int), ("idxbu", C.c_int * NUM), ("lbu", C.c_double * NUM), ("ubu", C.c_double * NUM),
                ("nbx", C.c_int), ("idxbx", C.c_int * NXM), ("lbx", C.c_double * NXM), ("ubx", C.c_double * NXM),
                ("uh", C.c_double * KM), ("soft", C.c_int),
                ("lsh", C.c_double * KM), ("ush", C.c_double * KM),
                ("zl", C.c_double * KM), ("zu", C.c_double * KM), ("Zl", C.c_double * KM), ("Zu", C.c_double * KM),
                ("qp_iter_max", C.c_int), ("mu0", C.c_double), ("thr0", C.c_double),
                ("tol_stat", C.c_double), ("tol_eq", C.c_double), ("tol_ineq", C.c_double),
                ("tol_comp", C.c_double), ("alpha_min", C.c_double),
                ("sbx", C.c_int * NXM), ("lsbx", C.c_double * NXM), ("usbx", C.c_double * NXM),
                ("zl_bx", C.c_double * NXM), ("zu_bx", C.c_double * NXM), ("Zl_bx", C.c_double * NXM), ("Zu_bx", C.c_double * NXM),
                ("sim_num_steps", C.c_int), ("nlp_max_iter", C.c_int),
                ("nlp_tol_stat", C.c_double), ("nlp_tol_eq", C.c_double), ("nlp_tol_ineq", C.c_double),
                ("nlp_tol_comp", C.c_double),
                ("hpipm_mode", C.c_int), ("cond_pred_corr", C.c_int), ("cpc_factor", C.c_double)]


# QP solver profiles (include/usvmpc.h, USVMPC_HPIPM_*): acados' `hpipm_mode` names, and this library's behaviour up to round 5
HPIPM_MODES = {"BALANCE": 0, "SPEED": 1, "ROBUST": 2, "R04": 3}


def lib_path():
    # USVMPC_LIB selects an alternate build of the same library (kernel A/B experiments)
    return os.environ.get("USVMPC_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "libusvmpc.so")


# torch bundles its own HIP runtime under the same SONAME.  If torch is imported BEFORE this library is
# loaded both share torch's copy and zero-copy tensors / RCCL on solver buffers work; the other order
# leaves torch unable to initialise its device.  Recorded here so the torch helpers can say so.
loaded_before_torch = False

EXPORTS = ["usvmpc_model_dims", "usvmpc_default_options", "usvmpc_hpipm_profile", "usvmpc_create", "usvmpc_destroy",
           "usvmpc_set", "usvmpc_get", "usvmpc_get_int", "usvmpc_solve", "usvmpc_solve_sqp", "usvmpc_solve_async",
           "usvmpc_sync", "usvmpc_get_device_ptr", "usvmpc_last_kernel_ms", "usvmpc_kernel_ms", "usvmpc_tick_ms", "usvmpc_fail_counts", "usvmpc_unconverged_counts", "usvmpc_unconverged_total", "usvmpc_handover_counts", "usvmpc_handover_co_counts", "usvmpc_followup_ms", "usvmpc_pipeline_stats", "usvmpc_last_mapping",
           "usvmpc_advance", "usvmpc_set_stream", "usvmpc_set_option", "usvmpc_calibrate_traffic", "usvmpc_guidance_reset", "usvmpc_guidance_prepare", "usvmpc_guidance_sense",
           "usvmpc_guidance_publish", "usvmpc_guidance_state", "usvmpc_device_bytes", "usvmpc_last_error",
           "usvmpc_debug_model_eval", "usvmpc_debug_obstacle_eval"]


_libs = {}


def load(path):
    """Load (once) a build of the solver library and declare its C ABI."""
    global loaded_before_torch
    path = os.path.abspath(path)
    if path in _libs:
        return _libs[path]
    if not os.path.exists(path):
        raise RuntimeError(
            "HIP library %s not found - build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(there is no CPU fallback)" % path)
    if not _libs:
        # torch shares its HIP runtime with this library only when it was imported FIRST; zero-copy tensor views and RCCL on solver buffers
        # need that (sharding.device_tensor says so if the order was the other way round).  The library does not pull torch in on its own -
        # seconds of start-up and a large footprint for every consumer, the single-instance drop-in faces included - unless asked to:
        # USVMPC_PRELOAD_TORCH=1 (bench.py, the tests and the tools import torch themselves before this package).
        if "torch" not in sys.modules and os.environ.get("USVMPC_PRELOAD_TORCH") == "1":
            import torch  # noqa: F401
        loaded_before_torch = "torch" not in sys.modules
    L = C.CDLL(path)
    L.usvmpc_model_dims.argtypes = [C.c_int, _ip, _ip]
    L.usvmpc_default_options.argtypes = [C.POINTER(Desc)]
    L.usvmpc_default_options.restype = None
    L.usvmpc_hpipm_profile.argtypes = [C.POINTER(Desc), C.c_int]
    L.usvmpc_create.argtypes = [C.POINTER(Desc), C.POINTER(C.c_void_p)]
    L.usvmpc_destroy.argtypes = [C.c_void_p]
    L.usvmpc_set.argtypes = [C.c_void_p, C.c_char_p, C.c_int, _dp, C.c_size_t]
    L.usvmpc_get.argtypes = [C.c_void_p, C.c_char_p, C.c_int, _dp, C.c_size_t]
    L.usvmpc_get_int.argtypes = [C.c_void_p, C.c_char_p, _ip]
    L.usvmpc_solve.argtypes = [C.c_void_p, _ip]
    L.usvmpc_solve_sqp.argtypes = [C.c_void_p, _ip]
    L.usvmpc_solve_async.argtypes = [C.c_void_p]
    L.usvmpc_sync.argtypes = [C.c_void_p]
    L.usvmpc_get_device_ptr.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_void_p)]
    L.usvmpc_last_kernel_ms.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    L.usvmpc_kernel_ms.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    L.usvmpc_tick_ms.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_float)]
    L.usvmpc_fail_counts.argtypes = [C.c_void_p, C.c_int, _ip]
    L.usvmpc_unconverged_counts.argtypes = [C.c_void_p, C.c_int, _ip]
    L.usvmpc_handover_counts.argtypes = [C.c_void_p, C.c_int, _ip]
    L.usvmpc_unconverged_total.argtypes = [C.c_void_p, C.POINTER(C.c_longlong)]
    L.usvmpc_handover_co_counts.argtypes = [C.c_void_p, C.c_int, _ip, _ip]
    L.usvmpc_followup_ms.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_float)]
    L.usvmpc_pipeline_stats.argtypes = [C.c_void_p, C.POINTER(C.c_long), C.POINTER(C.c_long)]
    L.usvmpc_last_mapping.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
    L.usvmpc_advance.argtypes = [C.c_void_p, C.c_double, C.c_ulonglong]
    L.usvmpc_set_stream.argtypes = [C.c_void_p, C.c_void_p]
    L.usvmpc_set_option.argtypes = [C.c_void_p, C.c_char_p, C.c_double]
    L.usvmpc_calibrate_traffic.argtypes = [C.c_void_p, C.c_int, _dp, _dp]
    L.usvmpc_guidance_reset.argtypes = [C.c_void_p, _dp, C.c_int, _dp]
    L.usvmpc_guidance_prepare.argtypes = [C.c_void_p, _dp, _dp, _dp, _ip, C.c_int]
    L.usvmpc_guidance_sense.argtypes = [C.c_void_p, _dp, _dp, C.c_int, C.c_double, _dp, _ip]
    L.usvmpc_guidance_publish.argtypes = [C.c_void_p, _dp, _dp, _dp, _dp, _ip]
    L.usvmpc_guidance_state.argtypes = [C.c_void_p, _ip, C.POINTER(C.c_float)]
    L.usvmpc_debug_model_eval.argtypes = [C.c_int, C.c_int, C.c_int, _dp, _dp, _dp, _dp]
    L.usvmpc_debug_obstacle_eval.argtypes = [C.c_int, C.c_int, C.c_int, _dp, _dp, _dp, _dp]
    L.usvmpc_device_bytes.argtypes = [C.c_void_p]
    L.usvmpc_device_bytes.restype = C.c_size_t
    L.usvmpc_last_error.argtypes = [C.c_void_p]
    L.usvmpc_last_error.restype = C.c_char_p
    _libs[path] = L
    return L


def lib_sha256(path=None):
    """sha256 (first 16 hex digits) of a solver library file: ties a measurement to the binary that produced it."""
    import hashlib
    h = hashlib.sha256()
    with open(path or lib_path(), "rb") as f:
        for chunk in iter(lambda: f.read(1 << 20), b""):
            h.update(chunk)
    return h.hexdigest()[:16]


def lib():
    """The stock library (hand-written models); raises if it has not been built (no fallback)."""
    return load(lib_path())


def _fill(dst, src, n, what="array"):
    a = np.ascontiguousarray(src, dtype=np.float64).reshape(-1)
    if a.size != n:
        raise Exception("mismatching dimension for %s: expected %d values, got %d" % (what, n, a.size))
    for i in range(n):
        dst[i] = a[i]


def hpipm_profile(d, mode):
    """The fields of the descriptor a QP solver profile governs (mirrors usvmpc_hpipm_profile / host_spec.hpp); mode: a name of
    HPIPM_MODES or its number."""
    m = HPIPM_MODES.get(str(mode).upper(), mode) if not isinstance(mode, int) else mode
    if m not in HPIPM_MODES.values():
        raise Exception("hpipm_mode must be one of %s" % ", ".join(HPIPM_MODES))
    r04 = m == HPIPM_MODES["R04"]
    d.hpipm_mode = m
    d.qp_iter_max = 50
    d.tol_stat, d.tol_eq, d.tol_ineq, d.tol_comp = 1e-6, 1e-8, 1e-8, 1e-8
    d.mu0 = 10.0 if r04 else 1.0
    d.alpha_min = 1e-12 if r04 else 1e-8
    d.cond_pred_corr = 0 if r04 else 1
    d.cpc_factor = 2.0
    return d


def default_options(d):
    """Solver option defaults (mirrors usvmpc_default_options so that it works without the .so)."""
    hpipm_profile(d, HPIPM_MODES["BALANCE"])
    d.thr0 = 0.1
    d.sim_num_steps = 1
    d.nlp_max_iter = 100
    d.nlp_tol_stat = d.nlp_tol_eq = d.nlp_tol_ineq = d.nlp_tol_comp = 1e-6
    return d


MODEL_GENERATED = 3


def desc_from_ocp(ocp, batch=1, device=0, generated=False):
    """Translate an AcadosOcp look-alike into the C description; validates like acados'
    make_consistent (dimension mismatches raise Exception).  generated: the model is compiled from its
    symbolic definition (model id 3) instead of coming from the registry of hand-written models."""
    name = ocp.model.name
    opts, cost, con = ocp.solver_options, ocp.cost, ocp.constraints
    if generated:
        mid, nx, nu = MODEL_GENERATED, ocp.model.x.size()[0], ocp.model.u.size()[0]
    else:
        if name not in MODEL_IDS:
            raise Exception("model '%s' is not in the registry %s and carries no symbolic definition" % (name, sorted(MODEL_IDS)))
        mid = MODEL_IDS[name]
        nx, nu = MODEL_DIMS[mid]
        if ocp.model.x.size()[0] != nx or ocp.model.u.size()[0] != nu:
            raise Exception("model dimensions do not match registry entry '%s'" % name)
    for fld, want in (("cost_type", "LINEAR_LS"), ("cost_type_e", "LINEAR_LS")):
        if getattr(cost, fld) != want:
            raise Exception("only %s = %s is supported" % (fld, want))
    if opts.nlp_solver_type not in ("SQP_RTI", "SQP"):
        raise Exception("nlp_solver_type must be SQP_RTI or SQP")
    if getattr(opts, "sim_method_num_stages", 4) not in (None, 4):
        raise Exception("sim_method_num_stages must be 4 (the ERK4 tableau); use sim_method_num_steps to refine")
    if opts.integrator_type != "ERK" or opts.hessian_approx != "GAUSS_NEWTON":
        raise Exception("integrator_type must be ERK and hessian_approx GAUSS_NEWTON")
    if opts.qp_solver not in ("PARTIAL_CONDENSING_HPIPM", "FULL_CONDENSING_HPIPM"):
        raise Exception("qp_solver must be an HPIPM variant")
    if getattr(opts, "qp_solver_warm_start", 0) not in (0, None):
        raise Exception("qp_solver_warm_start must be 0: every QP is cold-started (the acados default)")
    if getattr(opts, "nlp_solver_step_length", 1.0) not in (1.0, 1, None):
        raise Exception("nlp_solver_step_length must be 1.0 (full steps)")
    cn = getattr(opts, "qp_solver_cond_N", None)
    if cn is not None and not (1 <= int(cn) <= int(ocp.dims.N)):
        raise Exception("qp_solver_cond_N must lie in 1..N")
    if opts.tf is None or ocp.dims.N is None:
        raise Exception("solver_options.tf and dims.N must be set")
    ny, ny_e = nx + nu, nx
    d = Desc()
    default_options(d)
    d.model, d.N, d.Tf, d.batch, d.device = mid, int(ocp.dims.N), float(opts.tf), int(batch), int(device)
    npar = ocp.model.p.size()[0]
    K = 0 if ocp.model.con_h_expr is None else ocp.model.con_h_expr.size()[0]
    if K and npar != 2 * K:
        raise Exception("parameter vector must hold (ox, oy) per obstacle row: np = %d, nh = %d" % (npar, K))
    d.K = K
    _fill(d.W, cost.W, ny * ny, "cost.W (ny x ny)")
    _fill(d.W_e, cost.W_e, ny_e * ny_e, "cost.W_e (ny_e x ny_e)")
    _fill(d.Vx, cost.Vx, ny * nx, "cost.Vx (ny x nx)")
    _fill(d.Vu, cost.Vu, ny * nu, "cost.Vu (ny x nu)")
    _fill(d.Vx_e, cost.Vx_e, ny_e * nx, "cost.Vx_e (ny_e x nx)")
    idxbu = np.asarray(con.idxbu, dtype=int).reshape(-1)
    d.nbu = idxbu.size
    if np.asarray(con.lbu).size != d.nbu or np.asarray(con.ubu).size != d.nbu:
        raise Exception("lbu/ubu/idxbu dimension mismatch")
    for i in range(d.nbu):
        d.idxbu[i], d.lbu[i], d.ubu[i] = int(idxbu[i]), float(con.lbu[i]), float(con.ubu[i])
    idxbx = np.asarray(con.idxbx, dtype=int).reshape(-1)
    d.nbx = idxbx.size
    if np.asarray(con.lbx).size != d.nbx or np.asarray(con.ubx).size != d.nbx:
        raise Exception("lbx/ubx/idxbx dimension mismatch")
    for i in range(d.nbx):
        d.idxbx[i], d.lbx[i], d.ubx[i] = int(idxbx[i]), float(con.lbx[i]), float(con.ubx[i])
    if K:
        if np.asarray(con.lh).size != K or np.asarray(con.uh).size != K:
            raise Exception("lh/uh must have nh = %d entries" % K)
        for i in range(K):
            d.uh[i] = float(con.uh[i])
    idxsh = np.asarray(con.idxsh, dtype=int).reshape(-1)
    idxsbx = np.asarray(getattr(con, "idxsbx", []), dtype=int).reshape(-1)
    nsbx = idxsbx.size
    ns = nsbx + idxsh.size   # acados orders the slack penalty vectors [sbx.., sh..]
    if ns:
        for nm in ("zl", "zu", "Zl", "Zu"):
            if np.asarray(getattr(cost, nm)).size != ns:
                raise Exception("cost.%s must have ns = %d entries" % (nm, ns))
    if nsbx:
        nbx = np.asarray(con.idxbx).size
        if np.unique(idxsbx).size != nsbx or idxsbx.min() < 0 or idxsbx.max() >= nbx:
            raise Exception("idxsbx must hold distinct positions in the bx list (0..nbx-1)")
        for nm in ("lsbx", "usbx"):
            if np.asarray(getattr(con, nm)).size != nsbx:
                raise Exception("%s must have nsbx = %d entries" % (nm, nsbx))
        for j, i in enumerate(idxsbx):
            d.sbx[i] = 1
            d.lsbx[i], d.usbx[i] = float(con.lsbx[j]), float(con.usbx[j])
            d.zl_bx[i], d.zu_bx[i] = float(cost.zl[j]), float(cost.zu[j])
            d.Zl_bx[i], d.Zu_bx[i] = float(cost.Zl[j]), float(cost.Zu[j])
    if idxsh.size:
        if sorted(idxsh.tolist()) != list(range(K)):
            raise Exception("soft constraints are supported for all h rows at once (idxsh = 0..nh-1)")
        for nm in ("lsh", "ush"):
            if np.asarray(getattr(con, nm)).size != K:
                raise Exception("%s must have nsh = %d entries" % (nm, K))
        d.soft = 1
        for i in range(K):
            d.lsh[i], d.ush[i] = float(con.lsh[i]), float(con.ush[i])
            d.zl[i], d.zu[i] = float(cost.zl[nsbx + i]), float(cost.zu[nsbx + i])
            d.Zl[i], d.Zu[i] = float(cost.Zl[nsbx + i]), float(cost.Zu[nsbx + i])
    if generated and d.soft == 0 and idxsh.size == 0:
        pass
    if mid == 1 and K and not d.soft:
        raise Exception("usv_model_guidance_ca1 is built with soft obstacle rows")
    if mid == 2 and d.soft:
        raise Exception("usv_model_pf_ca is built with hard obstacle rows")
    if getattr(opts, "hpipm_mode", None) is not None:   # (before the individual knobs: they override the profile's values)
        hpipm_profile(d, opts.hpipm_mode)
    if opts.qp_solver_iter_max is not None:
        d.qp_iter_max = int(opts.qp_solver_iter_max)
    for src, dst in (("qp_solver_tol_stat", "tol_stat"), ("qp_solver_tol_eq", "tol_eq"),
                     ("qp_solver_tol_ineq", "tol_ineq"), ("qp_solver_tol_comp", "tol_comp")):
        if getattr(opts, src) is not None:
            setattr(d, dst, float(getattr(opts, src)))
    if getattr(opts, "sim_method_num_steps", None) is not None:
        d.sim_num_steps = int(opts.sim_method_num_steps)
        if d.sim_num_steps < 1:
            raise Exception("sim_method_num_steps must be >= 1")
    if getattr(opts, "nlp_solver_max_iter", None) is not None:
        d.nlp_max_iter = int(opts.nlp_solver_max_iter)
    for src, dst in (("nlp_solver_tol_stat", "nlp_tol_stat"), ("nlp_solver_tol_eq", "nlp_tol_eq"),
                     ("nlp_solver_tol_ineq", "nlp_tol_ineq"), ("nlp_solver_tol_comp", "nlp_tol_comp")):
        if getattr(opts, src, None) is not None:
            setattr(d, dst, float(getattr(opts, src)))
    return d
