"""Quick timing probe (development aid): python tools/quick_bench.py [name] [B] [N] [K] [iters]"""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from mpc_collisionavoidance_amd import BatchOcpSolver, scenario, usv_models

name = sys.argv[1] if len(sys.argv) > 1 else "usv_model_pf_ca"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
N = int(sys.argv[3]) if len(sys.argv) > 3 else 40
K = int(sys.argv[4]) if len(sys.argv) > 4 else 10
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 5
dt = scenario.DT[name]
ocp = usv_models.make_ocp(name, N * dt, N, None if name == "usv_model" else K)
wl = scenario.make_batch(name, N, K if name != "usv_model" else 0, B, dt=dt)
t0 = time.time()
s = BatchOcpSolver(ocp, B)
scenario.load_into(s, wl)
import os
if os.environ.get("USV_STATIC"): s.set_option("static_obstacles", 1)
if os.environ.get("USV_NOPACK"): s.set_option("pack_box_rows", 0)
print("setup %.2fs, device MB %.1f" % (time.time() - t0, s.device_bytes() / 1e6), flush=True)
for it in range(iters):
    t0 = time.time()
    st = s.solve()
    dtw = time.time() - t0
    lin, qp = s.last_kernel_ms()
    qi = s.get_int("qp_iter")
    print("iter %d: wall %.1f ms  lin %.2f ms  qp %.2f ms  -> %.0f solves/s | status!=0: %d  qp_iter mean %.1f max %d"
          % (it, dtw * 1e3, lin, qp, B / ((lin + qp) * 1e-3), int((st != 0).sum()), qi.mean(), qi.max()), flush=True)
