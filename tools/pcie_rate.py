"""PCIe-inclusive tick rate: the caller hands x0 / p / lh over from host memory every tick and reads u0 and x1
back (the reference's per-tick protocol, array-valued), versus the device-resident closed loop of bench.py."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from mpc_collisionavoidance_amd import BatchOcpSolver, scenario, usv_models
name, B, N, K = "usv_model_pf_ca", 65536, 40, 10
dt = scenario.DT[name]
ocp = usv_models.make_ocp(name, N * dt, N, K)
wl = scenario.make_batch(name, N, K, B)
s = BatchOcpSolver(ocp, B)
scenario.load_into(s, wl)
s.set_option("static_obstacles", 1)
for it in range(3):
    s.solve(); s.advance()
s.sync()
ticks = 8
t0 = time.perf_counter()
for it in range(ticks):
    s.solve_async(); s.advance()
s.sync()
dev = (time.perf_counter() - t0) / ticks
x1 = s.get("x", 1)
t0 = time.perf_counter()
for it in range(ticks):
    s.set("x0", 0, x1)                 # set(0,"lbx",x0)
    s.set("p", 0, wl["p"][:, 0])       # acados_update_params (stage-independent obstacle set)
    s.set("lh", 0, wl["lh"][:, 0])     # constraints_set(.,"lh",.)
    s.solve_async(); s.sync()
    u0 = s.get("u", 0)                 # get(0,"u")
    x1 = s.get("x", 1)                 # get(1,"x")
host = (time.perf_counter() - t0) / ticks
mb = (x1.nbytes * 2 + wl["p"][:, 0].nbytes + wl["lh"][:, 0].nbytes + u0.nbytes) / 1e6
print("device-resident tick %.1f ms (%.0f solves/s) | host hand-over tick %.1f ms (%.0f solves/s), %.1f MB over PCIe per tick"
      % (dev * 1e3, B / dev, host * 1e3, B / host, mb))
