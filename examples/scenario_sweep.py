#!/usr/bin/env python3
"""A batched LiDAR-scenario sweep with the whole tick on the device: the ROS pipeline of the reference
(obstacle simulator -> NMPC node callbacks / waypoint manager -> acados_solve -> published set-points,
catkin_ws/src/simulation/scripts/obstacle_sim_node.py + catkin_ws/src/nmpc_ca/src/nmpc_guidance_ca1.cpp) run for B
random obstacle fields at once.  As in the reference's own main.py the plant is the model prediction: the next
pose / velocity are read from x_1 of the solution.

    python examples/scenario_sweep.py --batch 4096 --ticks 300
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401,E402  (before the solver library: one HIP runtime for both)
from mpc_collisionavoidance_amd import BatchOcpSolver, usv_models  # noqa: E402
from mpc_collisionavoidance_amd.guidance import GuidanceFrontEnd  # noqa: E402


def make_worlds(B, L, rng):
    """L obstacles per scenario along the first leg (4,-5) -> (4,25): one every 25/L metres (jittered), up to
    1.5 m off the path to either side, radius 0.3..0.8 m - the reference scenario's density
    (usv_guidance_ca1/main.py: four obstacles on a 30 m leg), randomised."""
    y = -1.0 + (np.arange(L)[None, :] + rng.uniform(0.2, 0.8, (B, L))) * (25.0 / L)
    x = 4.0 + rng.uniform(-1.5, 1.5, (B, L))
    r = rng.uniform(0.3, 0.8, (B, L))
    return np.stack([x, y, r], axis=2)


def run(B=1024, ticks=600, N=100, K=8, L=5, seed=0, quiet=False):
    rng = np.random.default_rng(seed)
    dt = 0.05
    ocp = usv_models.make_ocp("usv_model_guidance_ca1", N * dt, N, K)
    s = BatchOcpSolver(ocp, B)
    fe = GuidanceFrontEnd(s)
    wps = np.array([[4.0, -5.0], [4.0, 25.0], [10.0, 30.0]])
    world = make_worlds(B, L, rng)
    pose = np.column_stack([4.0 + rng.uniform(-1, 1, B), np.full(B, -5.0), np.full(B, np.pi / 2) + rng.uniform(-0.3, 0.3, B)])
    vel = np.column_stack([np.full(B, 0.7), np.zeros(B)])
    fe.reset(wps, pose[:, 2])
    min_clear = np.full(B, np.inf)
    bad = np.zeros(B, dtype=bool)
    t0 = time.perf_counter()
    for i in range(ticks):
        fe.sense(pose, world, max_radius=100.0)      # obstacle_sim_node.simulate()
        fe.prepare(vel, pose)                        # obstaclesCallback + waypoint_manager + control() inputs
        st = s.solve()                               # acados_solve()
        out = fe.publish()                           # desired heading / r / speed
        bad |= (st != 0) & (out["active"] != 0)
        x1 = s.get("x", 1)
        vel, pose = x1[:, 0:2].copy(), x1[:, 5:8].copy()
        d = np.sqrt((pose[:, None, 0] - world[:, :, 0]) ** 2 + (pose[:, None, 1] - world[:, :, 1]) ** 2) - (world[:, :, 2] + 0.5)
        min_clear = np.minimum(min_clear, d.min(axis=1))
    el = time.perf_counter() - t0
    k, _ = fe.state()
    res = dict(ticks_per_s=ticks / el, scenario_ticks_per_s=B * ticks / el, min_clearance=min_clear, solver_failures=bad,
               waypoint_index=k, final_pose=pose)
    if not quiet:
        print("%d scenarios x %d ticks in %.2f s (%.0f scenario-ticks/s)" % (B, ticks, el, B * ticks / el))
        print("minimum clearance to the keep-out circle (R + 0.5 m): worst %.3f m, 1st percentile %.3f m, median %.3f m"
              % (min_clear.min(), np.percentile(min_clear, 1), np.median(min_clear)))
        print("scenarios with a solver failure: %d; reached the second leg: %d" % (bad.sum(), (k >= 2).sum()))
    s.close()
    return res


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--ticks", type=int, default=600)
    ap.add_argument("--horizon", type=int, default=100, help="the reference uses N = 100, Tf = 5 s; shorter horizons see the obstacles too late")
    a = ap.parse_args()
    run(a.batch, a.ticks, a.horizon)
