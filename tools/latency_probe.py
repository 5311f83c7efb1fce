"""Small-batch latency probe (GPU, development aid): ms per RTI solve of the whole batch for a range of batch sizes, with the
workspace in HBM and in LDS.   python tools/latency_probe.py [model] [N] [K]"""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpc_collisionavoidance_amd import BatchOcpSolver, scenario, usv_models
name = sys.argv[1] if len(sys.argv) > 1 else "usv_model_pf_ca"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 20
K = int(sys.argv[3]) if len(sys.argv) > 3 else 3
for B in (1, 64, 256, 512, 1024, 2048, 4096):
    wl = scenario.make_bench_batch(name, N, K, B)
    ocp = usv_models.make_ocp(name, N * scenario.BENCH_DT, N, K)
    ocp.solver_options.sim_method_num_steps = scenario.BENCH_SIM_STEPS[name]
    res = {}
    for mode in (0, 1):
        s = BatchOcpSolver(ocp, B)
        scenario.load_into(s, wl)
        s.set_option("static_obstacles", 1)
        s.set_option("disturbance_mask", scenario.NOISE_MASK[name])
        s.set_option("lds_workspace", mode)
        for t in range(3):
            s.solve_async(); s.advance(1e-3, seed=t)
        s.sync()
        t0 = time.perf_counter()
        for t in range(10):
            s.solve_async(); s.advance(1e-3, seed=10 + t)
        s.sync()
        wall = (time.perf_counter() - t0) / 10 * 1e3
        lin, qp = s.kernel_ms(10)
        res[mode] = (wall, qp.mean(), s.get_all("x"), s.get_int("qp_iter").mean())
        s.close()
    same = np.array_equal(res[0][2], res[1][2])
    print("B %5d  HBM: %.3f ms/tick (qp %.3f)   LDS: %.3f ms/tick (qp %.3f)   identical %s  iters %.1f" % (B, res[0][0], res[0][1], res[1][0], res[1][1], same, res[0][3]), flush=True)
