"""Where do the cycles of the partial-condensing kernel go?  (development aid)
  python tools/cond_timing.py build [name [flags]]   here: build_ab/libusvmpc_<name>.so = the shipped objects + cond_kernels.hip built with the flags (name "timing": -DUSV_COND_TIMING)
  USVMPC_LIB=... python tools/cond_timing.py run [B]   on the GPU box: BASELINE configs[4]'s shape with qp_cond_N = 10, cycles of thread 0 per phase
The phase numbers are the USV_TICK(n) marks of csrc/cond_ipm.hpp."""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "mpc_collisionavoidance_amd", "csrc")
OUT = os.path.join(ROOT, "build_ab", "libusvmpc_timing.so")
NAMES = {0: "condense", 1: "bf: loop head", 2: "bf: load_block", 3: "bf: pending step (3 expands)", 4: "bf: expand_rows", 5: "bf: row_pass", 6: "bf: residuals, rows_transposed x2",
         7: "bf: P [B A], stores", 8: "bf: Hessian triangle + rq", 9: "bf: elimination (nuh columns)", 10: "bf: scale, solve_forward", 11: "bf: stores, hand-over", 12: "bf: reductions",
         13: "rhs: loads", 14: "rhs: 2 expands", 15: "rhs: row_pass", 16: "rhs: transposed, rq, solve, stores", 17: "fw: loads", 18: "fw: solve_backward, dx", 19: "fw: expands",
         21: "fw: row_pass (+ dpi)", 22: "fw: reductions", 20: "finish"}

if sys.argv[1] == "build":   # build [name [flags ...]]: build_ab/libusvmpc_<name>.so; name "timing" adds -DUSV_COND_TIMING
    name = sys.argv[2] if len(sys.argv) > 2 else "timing"
    out = os.path.join(ROOT, "build_ab", "libusvmpc_%s.so" % name)
    os.makedirs(os.path.dirname(out), exist_ok=True)
    obj = os.path.join(ROOT, "build_ab", "cond_kernels_%s.o" % name)
    extra = sys.argv[3:] + (["-DUSV_COND_TIMING"] if name.startswith("timing") else [])
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-DUSV_COND_SEPARATE", "-I" + os.path.join(CSRC, "gfx950"), "-I" + CSRC] + extra +
                          ["-c", "-o", obj, os.path.join(CSRC, "cond_kernels.hip")])
    objs = [os.path.join(CSRC, "build", "usvmpc_p%d.o" % p) for p in range(6)] + [obj]
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-fPIC", "-shared", "-o", out] + objs)
    print(out)
    sys.exit(0)

sys.path.insert(0, ROOT)
import time
import numpy as np
import torch  # noqa
from mpc_collisionavoidance_amd import BatchOcpSolver, scenario, usv_models, _capi
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
name, N, K = "usv_model_pf_ca", 80, 20
wl = scenario.make_bench_batch(name, N, K, B, seed=1234, moving=True)
ocp = usv_models.make_ocp(name, N * scenario.BENCH_DT, N, K)
ocp.solver_options.sim_method_num_steps = scenario.BENCH_SIM_STEPS[name]
ocp.solver_options.qp_solver_cond_N = 10
s = BatchOcpSolver(ocp, B)
scenario.load_into(s, wl)
s.set_option("disturbance_mask", scenario.NOISE_MASK[name])
lib = _capi.lib()
has = hasattr(lib, "usvmpc_debug_cond_ticks")
buf = (ctypes.c_ulonglong * 32)()
for w in range(2):
    s.solve_async(); s.advance(1e-3, seed=1000 + w)
s.sync()
if has: lib.usvmpc_debug_cond_ticks(buf, 1)
steps = 4
t0 = time.perf_counter()
for k in range(steps):
    s.solve_async(); s.advance(1e-3, seed=2000 + k)
s.sync()
el = (time.perf_counter() - t0) / steps
qi = s.get_int("qp_iter")
print("B %d: %.2f ms per tick, %.0f solves/s, qp_iter mean %.2f, status != 0: %.4f" % (B, el * 1e3, B / el, qi.mean(), (s.get_int("status") != 0).mean()))
if has:
    lib.usvmpc_debug_cond_ticks(buf, 0)
    t = np.array(list(buf), dtype=np.float64)
    tot = t.sum()
    print("cycles of thread 0, all teams, %d ticks: %.3e (%.1f M per solve)" % (steps, tot, tot / (B * steps) / 1e6))
    for i in np.argsort(-t):
        if t[i] > 0: print("  %2d %-40s %5.1f %%   %8.0f cycles per solve" % (i, NAMES.get(int(i), "?"), 100 * t[i] / tot, t[i] / (B * steps)))
s.close()
