"""Closed-loop drift probe (CPU oracle): replays bench.py's loop (solve + x0 <- x1) on a sample of the bench
workload and prints, per tick, how many instances fail / hit the iteration cap and why."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpc_collisionavoidance_amd import scenario
from oracle import binding as ob

name = sys.argv[1] if len(sys.argv) > 1 else "usv_model_pf_ca"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
T = int(sys.argv[3]) if len(sys.argv) > 3 else 28
N, K = 40, 10
dt = float(sys.argv[4]) if len(sys.argv) > 4 else scenario.DT[name]
mid = {"usv_model": 0, "usv_model_guidance_ca1": 1, "usv_model_pf_ca": 2}[name]
wl = scenario.make_batch(name, N, K, B, dt=dt, seed=1234)
spec = ob.spec(mid, N, N * dt, K)
x, u = wl["x_init"].copy(), wl["u_init"].copy()
x0 = wl["x0"].copy()
for t in range(T):
    t0 = time.time()
    st, it = ob.rti_batch(spec, x, u, x0, wl["yref"], wl["yref_e"], wl["p"], wl["lh"], threads=8)
    x0 = x[:, 1].copy()
    print("tick %2d  status!=0 %5.2f%%  cap %5.2f%%  iter mean %.1f p99 %d max %d  (%.1fs)" % (
        t, 100 * (st != 0).mean(), 100 * (it >= 50).mean(), it.mean(), np.percentile(it, 99), it.max(), time.time() - t0), flush=True)
np.savez("/tmp/drift_%s.npz" % name, st=st, it=it, x=x, u=u, x0=x0)
