"""Look-alike of the slice of `acados_template` that the reference uses.

The reference builds its solvers with (paths relative to
/root/reference/catkin_ws/src/nmpc_ca/scripts/):

    from acados_template import AcadosModel, AcadosOcp, AcadosOcpSolver   usv_guidance_ca1/acados_settings.py:36
    ocp = AcadosOcp(); model_ac = AcadosModel(); ...                       :42-62
    acados_solver = AcadosOcpSolver(ocp, json_file="acados_ocp.json")      :207
    acados_solver.set(stage, field, value) / constraints_set / solve / get usv_guidance_ca1/main.py:111-175

The classes below keep those names, attributes, argument meaning and error behaviour (a wrong
length raises Exception, get returns a fresh array, solve returns the status and never raises),
but the solver behind them is the HIP library (batch of 1).  `BatchOcpSolver` is the same object
for B independent instances.
"""
import ctypes as C

import numpy as np

from . import _capi


class SymVec:
    """Stand-in for a CasADi column vector: only its length is ever used by the OCP definition
    (`model.x.size()[0]`: usv_guidance_ca1/acados_settings.py:65)."""

    def __init__(self, n, names=None):
        self.n = int(n)
        self.names = list(names) if names is not None else ["v%d" % i for i in range(self.n)]

    def size(self):
        return (self.n, 1)

    @property
    def shape(self):
        return (self.n, 1)

    def __len__(self):
        return self.n

    def __sub__(self, other):
        return self

    __rsub__ = __sub__


class AcadosModel:
    def __init__(self):
        self.name = None
        self.x = SymVec(0)
        self.xdot = SymVec(0)
        self.u = SymVec(0)
        self.z = SymVec(0)
        self.p = SymVec(0)
        self.f_impl_expr = None
        self.f_expl_expr = None
        self.con_h_expr = None


class AcadosOcpDims:
    def __init__(self):
        self.N = None


class AcadosOcpCost:
    def __init__(self):
        self.cost_type = "LINEAR_LS"
        self.cost_type_e = "LINEAR_LS"
        self.W = np.zeros((0, 0))
        self.W_e = np.zeros((0, 0))
        self.Vx = np.zeros((0, 0))
        self.Vu = np.zeros((0, 0))
        self.Vx_e = np.zeros((0, 0))
        self.yref = np.array([])
        self.yref_e = np.array([])
        self.zl = np.array([])
        self.zu = np.array([])
        self.Zl = np.array([])
        self.Zu = np.array([])


class AcadosOcpConstraints:
    def __init__(self):
        self.lbu = np.array([])
        self.ubu = np.array([])
        self.idxbu = np.array([])
        self.lbx = np.array([])
        self.ubx = np.array([])
        self.idxbx = np.array([])
        self.lh = np.array([])
        self.uh = np.array([])
        self.lsh = np.array([])
        self.ush = np.array([])
        self.idxsh = np.array([])
        # soft state bounds: positions in the bx list, lower bounds of their slacks (race_cars/acados_settings_dev.py:107-127)
        self.idxsbx = np.array([])
        self.lsbx = np.array([])
        self.usbx = np.array([])
        self.x0 = None


class AcadosOcpOptions:
    def __init__(self):
        self.tf = None
        self.qp_solver = "PARTIAL_CONDENSING_HPIPM"
        self.nlp_solver_type = "SQP_RTI"
        self.hessian_approx = "GAUSS_NEWTON"
        self.integrator_type = "ERK"
        # acados' `hpipm_mode` ("BALANCE" - its default -, "SPEED", "ROBUST"): the QP solver's argument profile; the reference never sets it
        # (usv_pf_ca/acados_settings.py:172-186).  "R04" selects this package's behaviour up to its round 5 (include/usvmpc.h, USVMPC_HPIPM_*).
        # None = "BALANCE".  The knobs below override the profile's values.
        self.hpipm_mode = None
        self.qp_solver_iter_max = None
        self.qp_solver_tol_stat = None
        self.qp_solver_tol_eq = None
        self.qp_solver_tol_ineq = None
        self.qp_solver_tol_comp = None
        # integrator refinement and full SQP (the knobs scripts/usv_guidance_ca1/acados_settings.py:192-204
        # mentions and leaves commented; scripts/race_cars/acados_settings_dev.py:157-164 sets them)
        self.sim_method_num_stages = 4
        self.sim_method_num_steps = 1
        self.nlp_solver_max_iter = 100
        self.nlp_solver_tol_stat = None
        self.nlp_solver_tol_eq = None
        self.nlp_solver_tol_ineq = None
        self.nlp_solver_tol_comp = None
        # acados' qp_solver_cond_N: None / N (what the reference leaves it at) = blocks of one stage, i.e. the Riccati sweep over
        # the N stages; a smaller value condenses the QP to that many dense stages on the device first (blocks as HPIPM partitions them)
        # (csrc/cond_ipm.hpp; not for soft state bounds - DESIGN.md section 6 has the measured comparison of the two)
        self.qp_solver_cond_N = None
        self.qp_solver_warm_start = 0   # 0 (cold start of every QP, the acados default) is the only mode built
        self.nlp_solver_step_length = 1.0
        self.print_level = 0
        self.model_source = None  # "symbolic": compile the model from its expressions even if the registry has it


class AcadosOcp:
    def __init__(self):
        self.model = AcadosModel()
        self.dims = AcadosOcpDims()
        self.cost = AcadosOcpCost()
        self.constraints = AcadosOcpConstraints()
        self.solver_options = AcadosOcpOptions()
        self.parameter_values = np.array([])


def _as_batch(value, B, n, what):
    a = np.ascontiguousarray(value, dtype=np.float64)
    if a.size == n and B != 1:
        a = np.tile(a.reshape(1, n), (B, 1))
    if a.size != B * n:
        raise Exception('mismatching dimension for field "%s" with dimension %d (you have %d)'
                        % (what, n, a.size // max(B, 1) if a.size % max(B, 1) == 0 else a.size))
    return np.ascontiguousarray(a.reshape(B, n))


class BatchOcpSolver:
    """B independent instances of one OCP on one MI355X; array-valued set/get replace the
    3N+4 ctypes round trips per tick of the reference loop (usv_guidance_ca1/main.py:123-130)."""

    def __init__(self, ocp, batch, device=0):
        self.ocp = ocp
        self.B = int(batch)
        # Route: hand-written device model from the registry (by model name and dimensions), or - when the
        # model carries a symbolic definition that the registry does not cover, or the caller asks for it
        # with solver_options.model_source = "symbolic" - a library generated and compiled for this model,
        # as acados does at this point.
        from . import casadi_lite
        m = ocp.model
        symbolic = isinstance(getattr(m, "f_expl_expr", None), casadi_lite.MXVec)
        known = m.name in _capi.MODEL_IDS and (m.x.size()[0], m.u.size()[0]) == _capi.MODEL_DIMS[_capi.MODEL_IDS[m.name]]
        self.generated = symbolic and (not known or getattr(ocp.solver_options, "model_source", None) == "symbolic")
        self._desc = _capi.desc_from_ocp(ocp, batch=self.B, device=device, generated=self.generated)
        self.N = self._desc.N
        self.K = self._desc.K
        if self.generated:
            from . import codegen, genbuild
            info = codegen.analyse(m)
            self.nx, self.nu = info.nx, info.nu
            self._lib = _capi.load(genbuild.build_device_lib(info, (self.K + 15) // 16, bool(self._desc.soft)))
        else:
            self.nx, self.nu = _capi.MODEL_DIMS[self._desc.model]
            self._lib = _capi.lib()
        self.ny, self.ny_e = self.nx + self.nu, self.nx
        d = self._desc
        # rows of a stage in acados' order [bu.., bx.., h..] and slack rows [sbx.., sh..]: the layout of get("lam" | "t")
        self.nrow = d.nbu + d.nbx + d.K
        self.ns = sum(1 for i in range(d.nbx) if d.sbx[i]) + (d.K if d.soft else 0)
        self.nlam = 2 * (self.nrow + self.ns)
        # (acados' make_consistent rejects a reference of the wrong length; an unset one - None or empty - stays zero)
        for nm, want in (("yref", self.ny), ("yref_e", self.ny_e)):
            v = getattr(ocp.cost, nm, None)
            if v is not None and np.asarray(v).size not in (0, want):
                raise Exception("inconsistent dimension: cost.%s has %d entries, the cost has ny%s = %d"
                                % (nm, np.asarray(v).size, "_e" if nm == "yref_e" else "", want))
        h = C.c_void_p()
        rc = self._lib.usvmpc_create(C.byref(self._desc), C.byref(h))
        if rc != 0:
            raise RuntimeError("usvmpc_create failed (%d): no usable HIP device or bad description" % rc)
        self._h = h
        cn = getattr(ocp.solver_options, "qp_solver_cond_N", None)
        if cn is not None and int(cn) != self.N:
            # (a condensed stage has nx + (N / cond_N) nu variables; the kernel holds one in a wave: at most 64 - the C ABI reports the
            # same limit at the first solve)
            mb = -(-self.N // int(cn))
            if self.nx + mb * self.nu > 64:
                self.close()
                raise Exception("qp_solver_cond_N = %d: a condensed stage would have %d variables (nx + ceil(N / cond_N) nu), at most 64 are built"
                                % (int(cn), self.nx + mb * self.nu))
            rc = self._lib.usvmpc_set_option(self._h, b"qp_cond_N", float(int(cn)))
            if rc != 0:
                msg = self._lib.usvmpc_last_error(self._h).decode()
                self.close()
                raise Exception("qp_solver_cond_N = %d: %s" % (int(cn), msg))
        # acados_create(): x trajectory = constraints.x0, u = 0, yref / p / lh from the ocp
        x0 = np.zeros(self.nx) if ocp.constraints.x0 is None else np.asarray(ocp.constraints.x0, dtype=float)
        self.set_all("x", np.tile(x0, (self.B, self.N + 1, 1)))
        self.set_all("u", np.zeros((self.B, self.N, self.nu)))
        self.set("x0", 0, x0)
        if np.asarray(ocp.cost.yref).size == self.ny:
            self.set_all("yref", np.tile(np.asarray(ocp.cost.yref, dtype=float), (self.B, self.N, 1)))
        if np.asarray(ocp.cost.yref_e).size == self.ny_e:
            self.set("yref", self.N, np.asarray(ocp.cost.yref_e, dtype=float))
        if self.K:
            pv = np.asarray(ocp.parameter_values, dtype=float)
            if pv.size != 2 * self.K:
                raise Exception("parameter_values must have np = %d entries" % (2 * self.K))
            self.set_all("p", np.tile(pv, (self.B, self.N + 1, 1)))
            self.set_all("lh", np.tile(np.asarray(ocp.constraints.lh, dtype=float), (self.B, self.N, 1)))

    # -- field table: name -> (per-stage length, number of stages)
    def _field(self, field, stage):
        N = self.N
        tab = {"x": (self.nx, N + 1), "u": (self.nu, N), "p": (2 * self.K, N + 1), "lh": (self.K, N),
               "pi": (self.nx, N), "sl": (self.K, N), "su": (self.K, N), "x0": (self.nx, 1),
               "lam": (self.nlam, N + 1), "t": (self.nlam, N + 1)}
        if field == "yref":
            return (self.ny_e, 1) if stage == N else (self.ny, N)
        if field not in tab:
            raise Exception("unknown field %s" % field)
        return tab[field]

    def _check(self, rc):
        if rc < 0:
            raise Exception(self._lib.usvmpc_last_error(self._h).decode())
        return rc

    def set(self, field, stage, value):
        n, _ = self._field(field, stage)
        a = _as_batch(value, self.B, n, field)
        self._check(self._lib.usvmpc_set(self._h, field.encode(), int(stage), a.ctypes.data_as(_capi._dp), n))

    def set_all(self, field, value):
        """value: [B, n_stages, n] for every stage at once."""
        n, ns = self._field(field, 0)
        a = np.ascontiguousarray(value, dtype=np.float64)
        if a.size != self.B * ns * n:
            raise Exception('mismatching dimension for field "%s": expected %d x %d x %d values, got %d'
                            % (field, self.B, ns, n, a.size))
        self._check(self._lib.usvmpc_set(self._h, field.encode(), -1, a.ctypes.data_as(_capi._dp), n))

    def get(self, field, stage):
        if field in ("res", "nlp_res"):
            out = np.zeros((self.B, 4))
            self._check(self._lib.usvmpc_get(self._h, field.encode(), 0, out.ctypes.data_as(_capi._dp), 4))
            return out
        if field == "obs_tmin":
            out = np.zeros(self.B)
            self._check(self._lib.usvmpc_get(self._h, b"obs_tmin", 0, out.ctypes.data_as(_capi._dp), 1))
            return out
        n, _ = self._field(field, stage)
        out = np.zeros((self.B, n))
        self._check(self._lib.usvmpc_get(self._h, field.encode(), int(stage), out.ctypes.data_as(_capi._dp), n))
        return out

    def get_all(self, field):
        n, ns = self._field(field, 0)
        out = np.zeros((self.B, ns, n))
        self._check(self._lib.usvmpc_get(self._h, field.encode(), -1, out.ctypes.data_as(_capi._dp), n))
        return out

    def get_int(self, field):
        out = np.zeros(self.B, dtype=np.int32)
        self._check(self._lib.usvmpc_get_int(self._h, field.encode(), out.ctypes.data_as(_capi._ip)))
        return out

    def solve(self):
        """One SQP-RTI iteration for every instance. Returns the per-instance status array."""
        st = np.zeros(self.B, dtype=np.int32)
        self._check(self._lib.usvmpc_solve(self._h, st.ctypes.data_as(_capi._ip)))
        return st

    def solve_sqp(self):
        """Full SQP for every instance (nlp_solver_type "SQP"): iterate until the NLP residuals are below
        nlp_solver_tol_* (status 0), nlp_solver_max_iter QPs were solved (2) or a QP failed (4).  Converged
        instances are frozen while the others continue.  get_int("sqp_iter") / get("nlp_res", 0) describe the run."""
        st = np.zeros(self.B, dtype=np.int32)
        self._check(self._lib.usvmpc_solve_sqp(self._h, st.ctypes.data_as(_capi._ip)))
        return st

    def solve_async(self):
        self._check(self._lib.usvmpc_solve_async(self._h))

    def sync(self):
        self._check(self._lib.usvmpc_sync(self._h))

    def device_ptr(self, field):
        p = C.c_void_p()
        self._check(self._lib.usvmpc_get_device_ptr(self._h, field.encode(), C.byref(p)))
        return p.value

    def last_kernel_ms(self):
        a, b = C.c_float(), C.c_float()
        self._check(self._lib.usvmpc_last_kernel_ms(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def kernel_ms(self, n):
        """(linearize_ms[n], qp_ms[n]) of the last n solves, from HIP events on the solver's stream."""
        a = (C.c_float * n)()
        b = (C.c_float * n)()
        self._check(self._lib.usvmpc_kernel_ms(self._h, n, a, b))
        return np.array(a[:]), np.array(b[:])

    def tick_ms(self, n):
        """Tick-to-tick times over the last n solves (n - 1 values, oldest first): start of solve i + 1 minus start of solve i on the
        solver's stream (usvmpc_tick_ms)."""
        if n < 2:
            return np.zeros(0)
        a = (C.c_float * (n - 1))()
        self._check(self._lib.usvmpc_tick_ms(self._h, n, a))
        return np.array(a[:])

    def fail_counts(self, n):
        """Instances with status != 0 in each of the last n solves (oldest first), counted on the device."""
        a = (C.c_int * n)()
        self._check(self._lib.usvmpc_fail_counts(self._h, n, a))
        return np.array(a[:])

    def pipeline_stats(self):
        """(used, discarded): linearisations made ahead of time that the following solve used / threw away (usvmpc_pipeline_stats)."""
        a, b = C.c_long(), C.c_long()
        self._check(self._lib.usvmpc_pipeline_stats(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def last_mapping(self):
        """0: the last RTI solve ran four instances per wavefront; 1: one instance per wavefront (option "wide"); 4: one instance per
        workgroup of four wavefronts (option "wide_waves") - usvmpc_last_mapping."""
        a = C.c_int()
        self._check(self._lib.usvmpc_last_mapping(self._h, C.byref(a)))
        return a.value

    def unconverged_counts(self, n):
        """Instances whose QP did not converge to the IPM tolerances (qp_status != 0) in each of the last n solves (oldest first)."""
        a = (C.c_int * n)()
        self._check(self._lib.usvmpc_unconverged_counts(self._h, n, a))
        return np.array(a[:])

    def followup_ms(self, n):
        """Duration (ms) of the follow-up launch of each of the last n RTI solves (kernel usv_qp_resume: the instances the main launch
        handed over, option "handover_iter"), oldest first; 0 for a solve without one.  Part of kernel_ms' QP time."""
        a = (C.c_float * n)()
        self._check(self._lib.usvmpc_followup_ms(self._h, n, a))
        return np.array(a[:])

    def handover_counts(self, n):
        """Instances each of the last n RTI launches handed over to its follow-up launch (option "handover_iter"), oldest first."""
        a = (C.c_int * n)()
        self._check(self._lib.usvmpc_handover_counts(self._h, n, a))
        return np.array(a[:])

    def unconverged_total(self):
        """Instances whose QP did not converge to the IPM tolerances, summed over every RTI solve of this handle (device-side running sum)."""
        v = C.c_longlong(0)
        self._check(self._lib.usvmpc_unconverged_total(self._h, C.byref(v)))
        return int(v.value)

    def handover_co_counts(self, n):
        """(finished, timeouts) of the follow-up kernel that runs BESIDE the launch (option "handover_co") for each of the last n solves:
        instances it finished, and waits of its workgroups that ran into the spin limit."""
        a, b = (C.c_int * n)(), (C.c_int * n)()
        self._check(self._lib.usvmpc_handover_co_counts(self._h, n, a, b))
        return np.array(a[:]), np.array(b[:])

    def advance(self, sigma=0.0, seed=0):
        """Closed-loop hand-over on the device: x0 <- x_1 (+ sigma N(0,1)); asynchronous."""
        self._check(self._lib.usvmpc_advance(self._h, float(sigma), int(seed)))

    def set_option(self, name, value):
        self._check(self._lib.usvmpc_set_option(self._h, name.encode(), float(value)))

    def set_stream(self, stream_ptr):
        self._check(self._lib.usvmpc_set_stream(self._h, C.c_void_p(stream_ptr)))

    def device_bytes(self):
        return int(self._lib.usvmpc_device_bytes(self._h))

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.usvmpc_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class AcadosOcpSolver:
    """Single-instance solver with acados_template's calling convention."""

    def __init__(self, acados_ocp, json_file="acados_ocp.json", device=0):
        self.acados_ocp = acados_ocp
        self.json_file = json_file  # accepted for signature compatibility; nothing is rendered
        self._b = BatchOcpSolver(acados_ocp, 1, device=device)
        self.N = self._b.N
        self.status = 0
        self._sqp = acados_ocp.solver_options.nlp_solver_type == "SQP"

    def _vec(self, value, n, field):
        a = np.ascontiguousarray(value, dtype=np.float64).reshape(-1)
        if a.size != n:
            raise Exception('AcadosOcpSolver.set(): mismatching dimension for field "{}" with dimension {} (you have {})'
                            .format(field, n, a.size))
        return a

    def set(self, stage_, field_, value_):
        b = self._b
        if field_ in ("lbx", "ubx"):
            if stage_ != 0:
                raise Exception("lbx/ubx can be set at stage 0 only (x0 embedding); state bounds are part of the OCP definition")
            v = self._vec(value_, b.nx, field_)
            setattr(self, "_" + field_, v)
            # the reference always writes lbx = ubx = x0 (usv_guidance_ca1/main.py:111-112): every write moves x0, and
            # solve() refuses to run while the two bounds disagree (a genuine stage-0 box is not supported)
            b.set("x0", 0, v)
            return
        if field_ == "yref":
            n = b.ny_e if stage_ == b.N else b.ny
            b.set("yref", stage_, self._vec(value_, n, field_))
        elif field_ == "p":
            b.set("p", stage_, self._vec(value_, 2 * b.K, field_))
        elif field_ in ("x", "u"):
            b.set(field_, stage_, self._vec(value_, b.nx if field_ == "x" else b.nu, field_))
        elif field_ == "lh":
            b.set("lh", stage_, self._vec(value_, b.K, field_))
        else:
            raise Exception("AcadosOcpSolver.set(): {} is not a valid argument.".format(field_))

    def cost_set(self, stage_, field_, value_):
        if field_ != "yref":
            raise Exception("AcadosOcpSolver.cost_set(): only yref can be changed at run time")
        self.set(stage_, "yref", value_)

    def constraints_set(self, stage_, field_, value_):
        if field_ in ("lbx", "ubx", "lh"):
            self.set(stage_, field_, value_)
        else:
            raise Exception("AcadosOcpSolver.constraints_set(): {} is not a valid argument.".format(field_))

    def solve(self):
        lo, hi = getattr(self, "_lbx", None), getattr(self, "_ubx", None)
        if lo is not None and hi is not None and not np.array_equal(lo, hi):
            # solve() never raises (acados returns a status): a genuine stage-0 box is not supported - this solver embeds the
            # initial state as lbx = ubx = x0 - so the call fails like a QP failure, iterate untouched, with one warning
            import warnings
            warnings.warn("AcadosOcpSolver.solve(): lbx and ubx of stage 0 differ - this solver embeds the initial state as "
                          "lbx = ubx = x0 and does not support a stage-0 box; returning status 4 without solving")
            self.status = 4
            return self.status
        if self._sqp:
            self.status = int(self._b.solve_sqp()[0])
        else:
            self.status = int(self._b.solve()[0])
        return self.status

    def get(self, stage_, field_):
        if field_ not in ("x", "u", "pi", "sl", "su"):
            raise Exception("AcadosOcpSolver.get(): {} is an invalid argument.".format(field_))
        if field_ == "pi":
            return self._b.get("pi", stage_ + 1)[0].copy()  # acados: pi of stage k couples k -> k+1
        return self._b.get(field_, stage_)[0].copy()

    def get_stats(self, field_):
        if field_ == "qp_iter":
            return int(self._b.get_int("qp_iter")[0])
        if field_ == "sqp_iter":
            return int(self._b.get_int("sqp_iter")[0]) if self._sqp else 1
        if field_ == "residuals":  # acados: the NLP residuals for SQP; the QP's for an RTI iteration
            return self._b.get("nlp_res" if self._sqp else "res", 0)[0].copy()
        raise Exception("AcadosOcpSolver.get_stats(): {} is not a valid argument.".format(field_))
