"""Mid-size batches: would splitting a launch BY PREDICTED DIFFICULTY across the two mappings pay?  (GPU, development aid)
The launch of a batch that is resident from the start lasts as long as its hardest instances on a lone 16-lane row.  This probe emulates the split
with TWO handles on streams of their own: the K instances with the highest iteration counts of the last warm-up tick on the latency mapping
(option wide = 1: planes in LDS, 0.144 ms per pass), the rest on the throughput mapping - against one handle with the whole batch.
python tools/split_probe.py [model] [N] [K_obstacles] [batch] [K_hard, comma separated]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa
from mpc_collisionavoidance_amd import BatchOcpSolver, scenario, usv_models
name = sys.argv[1] if len(sys.argv) > 1 else "usv_model_pf_ca"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 40
K = int(sys.argv[3]) if len(sys.argv) > 3 else 10
B = int(sys.argv[4]) if len(sys.argv) > 4 else 8192
hard = [int(v) for v in sys.argv[5].split(",")] if len(sys.argv) > 5 else [128, 256, 512]
steps, warm = 20, 6
wl = scenario.make_bench_batch(name, N, K, B, seed=1234)
ocp = usv_models.make_ocp(name, N * scenario.BENCH_DT, N, K)
ocp.solver_options.sim_method_num_steps = scenario.BENCH_SIM_STEPS[name]


def make(w, n, opts=()):
    s = BatchOcpSolver(ocp, n)
    scenario.load_into(s, w)
    s.set_option("static_obstacles", 1)
    s.set_option("disturbance_mask", scenario.NOISE_MASK[name])
    for k, v in opts:
        s.set_option(k, v)
    return s


def loop(hs, n):
    for s in hs:
        s.sync()
    t0 = time.perf_counter()
    for k in range(n):
        for s in hs:
            s.solve_async()
        for s in hs:
            s.advance(1e-3, seed=2000 + k)
    for s in hs:
        s.sync()
    return (time.perf_counter() - t0) / n * 1e3


one = make(wl, B)
loop([one], warm)
it = one.get_int("qp_iter").copy()
x, u, x0 = one.get_all("x"), one.get_all("u"), one.get("x0", 0).copy()
t_one = loop([one], steps)
it2 = one.get_int("qp_iter")
print("%s N=%d K=%d B=%d | one handle: %.2f ms per step (mapping %d); qp_iter mean %.1f, >= 30: %d, at the cap: %d; tick-to-tick correlation of the counts %.2f"
      % (name, N, K, B, t_one, one.last_mapping(), it.mean(), (it >= 30).sum(), (it >= 50).sum(), np.corrcoef(it, it2)[0, 1]), flush=True)
one.close()
order = np.argsort(-it, kind="stable")
for kh in hard:
    ia, ib = np.sort(order[:kh]), np.sort(order[kh:])
    hs = []
    for idx, opts in ((ia, (("wide", 1),)), (ib, (("wide", 0),))):
        w = {k: (v[idx] if isinstance(v, np.ndarray) and v.shape[:1] == (B,) else v) for k, v in wl.items()}
        s = make(w, len(idx), opts)
        s.set_all("x", x[idx]); s.set_all("u", u[idx]); s.set("x0", 0, x0[idx])
        hs.append(s)
    loop(hs, 2)
    t = loop(hs, steps)
    ta, tb = loop(hs[:1], 5), loop(hs[1:], 5)
    print("  %4d hardest (>= %d iterations) on the latency mapping [%d] + %d on the throughput mapping [%d]: %.2f ms per step = %.2f x; alone: %.2f / %.2f ms"
          % (kh, it[order[kh - 1]], hs[0].last_mapping(), len(ib), hs[1].last_mapping(), t, t_one / t, ta, tb), flush=True)
    for s in hs:
        s.close()
