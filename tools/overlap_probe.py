"""Could the NEXT tick's lineariser hide in the tail of the QP launch?  (GPU, timing only, build_ab/libusvmpc_timing.so)
Handle A runs the normal step (lineariser + QP + hand-over) on its stream; handle B, same batch size, runs ONLY its lineariser
(timing switch) on a stream of its own, enqueued right after A's QP: its workgroups are dispatched as A's persistent waves leave.
Reported: ms per step of A alone, and of A with B's lineariser riding along (= what a pipelined lineariser would cost)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["USVMPC_LIB"] = os.path.join(ROOT, "build_ab", "libusvmpc_timing.so")
import torch  # noqa
from mpc_collisionavoidance_amd import BatchOcpSolver, scenario, usv_models
name, N, K, B, steps = "usv_model_pf_ca", 40, 10, 65536, 20
wl = scenario.make_bench_batch(name, N, K, B, seed=1234)
ocp = usv_models.make_ocp(name, N * scenario.BENCH_DT, N, K)
ocp.solver_options.sim_method_num_steps = scenario.BENCH_SIM_STEPS[name]
def mk():
    s = BatchOcpSolver(ocp, B)
    scenario.load_into(s, wl)
    s.set_option("static_obstacles", 1)
    s.set_option("disturbance_mask", scenario.NOISE_MASK[name])
    return s
A, Bh = mk(), mk()
Bh.set_option("timing_fixed_iters", -1)   # lineariser only
for w in range(3):
    A.solve_async(); A.advance(1e-3, seed=1000 + w)
A.sync()
for mode in ("A alone", "A + lineariser of B", "A alone", "A + lineariser of B"):
    t0 = time.perf_counter()
    for k in range(steps):
        A.solve_async(); A.advance(1e-3, seed=2000 + k)
        if mode != "A alone":
            Bh.solve_async()
    A.sync(); Bh.sync()
    el = (time.perf_counter() - t0) / steps * 1e3
    lin, qp = A.kernel_ms(steps)
    print("%-24s %.2f ms per step   (A: lineariser %.2f, QP %.2f)" % (mode, el, lin.mean(), qp.mean()), flush=True)
