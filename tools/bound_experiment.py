"""What bounds usv_qp_rti on the headline workload?  One decisive experiment (GPU, timing only).

Library: build_ab/libusvmpc_timing.so (tools/dev_build.sh timing -DUSV_TIMING_EXPERIMENT): the shipped kernels with two
switches - every instance runs EXACTLY `fixed_iters` IPM iterations (no data-dependent control flow left), and the plane
addresses of group g are folded onto group g % alias_groups.  The instruction stream, the launch and the number of plane
accesses are identical in every run; only WHERE the planes live changes:
    alias 0      8192 resident instances x 131 KB = 1.07 GB in flight: HBM (the shipped behaviour)
    alias 1024   134 MB: inside the 256 MB Infinity Cache
    alias 128    16.8 MB: Infinity Cache, a few lines per L2
    alias 16     2.1 MB: inside every XCD's 4 MB L2
If the kernel waits for HBM, it speeds up materially as the window shrinks; if it is bound by the issue rate of its two waves
per SIMD it does not.  The same is repeated with ONE wave per SIMD (option max_waves): an issue-bound kernel slows down by the
lone-wave factor whatever the window, a memory-bound one hardly at alias 0.
Results (garbage numerics by construction) -> stdout; tracked copy: profiles/r03_bound_experiment.txt.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["USVMPC_LIB"] = os.path.join(ROOT, "build_ab", "libusvmpc_timing.so")
import torch  # noqa: E402,F401
from mpc_collisionavoidance_amd import BatchOcpSolver, scenario, usv_models  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "usv_model_pf_ca"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
N, K, ITERS = 40, 10, 16
wl = scenario.make_bench_batch(name, N, K, B, seed=1234)
dt, steps = scenario.BENCH_DT, scenario.BENCH_SIM_STEPS[name]
ocp = usv_models.make_ocp(name, N * dt, N, K)
ocp.solver_options.sim_method_num_steps = steps
s = BatchOcpSolver(ocp, B)
scenario.load_into(s, wl)
s.set_option("static_obstacles", 1)
s.set_option("sort_by_difficulty", 0)
s.set_option("timing_fixed_iters", ITERS)


def timed(alias, waves, reps=4):
    s.set_option("timing_alias_groups", alias)
    s.set_option("max_waves", waves)
    ms = []
    for r in range(reps + 1):
        s.set_all("x", wl["x_init"]); s.set_all("u", wl["u_init"]); s.set("x0", 0, wl["x0"])   # (the results are garbage: reload)
        s.solve_async()
        s.sync()
        if r:
            ms.append(s.last_kernel_ms()[1])
    return float(np.mean(ms)), float(np.std(ms))


print("%s, batch %d, N=%d, K=%d, every instance %d IPM iterations; usv_qp_rti ms per launch (mean of 4, +- std)" % (name, B, N, K, ITERS))
print("%-28s %-22s %-22s" % ("window", "2 waves / SIMD", "1 wave / SIMD"))
planes_bytes = None
for alias, label in ((0, "HBM (no alias, 1.07 GB)"), (2048, "alias 2048 (268 MB)"), (1024, "alias 1024 (134 MB)"),
                     (128, "alias 128 (16.8 MB)"), (16, "alias 16 (2.1 MB, L2)")):
    a = timed(alias, 0)
    b = timed(alias, 1024)
    print("%-28s %7.2f +- %-10.2f %7.2f +- %-10.2f ratio %.2f" % (label, a[0], a[1], b[0], b[1], b[0] / a[0]), flush=True)
s.close()
