"""Soak of the default hand-over policy at the batch sizes where the follow-up kernel runs beside the launch: per-tick wall times over many
closed-loop ticks, the follow-up kernel's timeouts (a workgroup that waited its whole bound), ticks far above the median.  (development aid, GPU)
  python tools/co_soak.py [B ...] [--ticks N] [--model usv_model_guidance_ca1] [--opt name=value ...]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa
from mpc_collisionavoidance_amd import BatchOcpSolver, scenario, usv_models
args = [a for a in sys.argv[1:]]
ticks = 2000
if "--ticks" in args:
    i = args.index("--ticks"); ticks = int(args[i + 1]); del args[i:i + 2]
opts = []
while "--opt" in args:
    i = args.index("--opt"); opts.append(args[i + 1].split("=")); del args[i:i + 2]
name, N, K = "usv_model_pf_ca", 40, 10
if "--model" in args:
    i = args.index("--model"); name = args[i + 1]; del args[i:i + 2]
for B in [int(a) for a in args] or [4096]:
    wl = scenario.make_bench_batch(name, N, K, B, seed=1234)
    ocp = usv_models.make_ocp(name, N * scenario.BENCH_DT, N, K)
    ocp.solver_options.sim_method_num_steps = scenario.BENCH_SIM_STEPS[name]
    s = BatchOcpSolver(ocp, B)
    scenario.load_into(s, wl)
    s.set_option("static_obstacles", 1)
    s.set_option("disturbance_mask", scenario.NOISE_MASK[name])
    for k_, v_ in opts:
        s.set_option(k_, float(v_))
    for w in range(5):
        s.solve_async(); s.advance(1e-3, seed=100 + w)
    s.sync()
    t = np.zeros(ticks)
    tout = 0
    for k in range(ticks):
        t0 = time.perf_counter()
        s.solve_async(); s.advance(1e-3, seed=1000 + k)
        s.sync()
        t[k] = time.perf_counter() - t0
        fin, to = s.handover_co_counts(1)
        if to[0] > 0 or t[k] > 0.05:
            lin, qp = s.kernel_ms(1)
            fu = s.followup_ms(1)
            print("  tick %d: %.2f ms, follow-up kernel finished %d instances, timeouts %d; lineariser %.2f ms, main launch until its end event %.2f ms, from there to the end of the tick's QPs %.2f ms"
                  % (k, t[k] * 1e3, fin[0], to[0], lin[0], qp[0] - fu[0], fu[0]), flush=True)
            tout += int(to[0])
    med = np.median(t)
    slow = np.nonzero(t > 3 * med)[0]
    print("B %d: %d ticks, median %.2f ms, p99 %.2f, max %.2f; ticks above 3 x the median: %d %s; timeouts %d" % (
        B, ticks, med * 1e3, np.percentile(t, 99) * 1e3, t.max() * 1e3, slow.size, [(int(i), round(t[i] * 1e3, 1)) for i in slow[:8]], tout), flush=True)
    s.close()
