#!/usr/bin/env python3
"""The reference's three closed-loop simulations on this stack, with its calling sequence unchanged
(single instance, acados_template look-alike, plant = model prediction, no trajectory shift):

  --variant usv_acados        catkin_ws/src/nmpc_ca/scripts/usv_acados/main.py:52-112
  --variant usv_guidance_ca1  catkin_ws/src/nmpc_ca/scripts/usv_guidance_ca1/main.py:54-205
  --variant usv_pf_ca         catkin_ws/src/nmpc_ca/scripts/usv_pf_ca/main.py:54-190

Prints what the reference prints (average / maximum solve time, tracking-error statistics); plotting
(plotFcn.py) is out of scope.  Needs an MI355X (there is no CPU fallback).
"""
import argparse
import sys
import time

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from mpc_collisionavoidance_amd.usv_models import acados_settings  # noqa: E402


def usv_acados(ticks):
    Tf, N = 1.0, 20
    constraint, model, acados_solver = acados_settings(Tf, N, name="usv_model")
    nx, nu = model.x.size()[0], model.U.size()[0]
    Nsim = ticks or int(10.0 * N / Tf)
    simX, simU = np.ndarray((Nsim, nx)), np.ndarray((Nsim, nu))
    tcomp_sum = tcomp_max = 0.0
    uref = 1.3
    for i in range(Nsim):
        for j in range(N):
            acados_solver.set(j, "yref", np.array([uref, 0, 0, 0, 0, 0, 0]))
        acados_solver.set(N, "yref", np.array([uref, 0, 0, 0, 0]))
        t = time.time()
        status = acados_solver.solve()
        if status != 0:
            print("acados returned status {} in closed loop iteration {}.".format(status, i))
        elapsed = time.time() - t
        tcomp_sum += elapsed
        tcomp_max = max(tcomp_max, elapsed)
        simX[i], simU[i] = acados_solver.get(0, "x"), acados_solver.get(0, "u")
        x0 = acados_solver.get(1, "x")
        acados_solver.set(0, "lbx", x0)
        acados_solver.set(0, "ubx", x0)
    print("final surge speed u = %.4f (reference %.2f)" % (simX[-1, 0], uref))
    return tcomp_sum / Nsim, tcomp_max


def usv_guidance_ca1(ticks):
    Tf, N = 5.0, 100
    constraint, model, acados_solver = acados_settings(Tf, N, name="usv_model_guidance_ca1")
    nx, nu = model.x.size()[0], model.U.size()[0]
    Nsim = ticks or int(50.0 * N / Tf)
    simX, simU, simError = np.ndarray((Nsim, nx)), np.ndarray((Nsim, nu)), np.ndarray((Nsim, 3))
    obsx, obsy = np.array([4, 4, 4, 4]), np.array([4, 7.0, 12, 20])
    radius = np.array([1.5, 1.5, 1.5, 1.5, 0, 0, 0, 0])
    pobs, robs = np.ones(16) * 100, np.zeros(8)
    tcomp_sum = tcomp_max = psi_mae = ye_mae = psi_mse = ye_mse = 0.0
    nedx = nedy = psi = 0.0
    u, v = 0.7, 0.0
    x1, y1, x2, y2 = 4.0, -5.0, 4.0, 25.0
    ak = np.arctan2(y2 - y1, x2 - x1)
    ye = -(nedx - x1) * np.sin(ak) + (nedy - y1) * np.cos(ak)
    psie = psi - ak
    x0 = np.array([u, v, ye, psie, psie, nedx, nedy, psi])
    acados_solver.set(0, "lbx", x0)
    acados_solver.set(0, "ubx", x0)
    nstat = 0
    for i in range(Nsim):
        for ii in range(len(obsx)):
            pobs[2 * ii], pobs[2 * ii + 1], robs[ii] = obsx[ii], obsy[ii], radius[ii]
        for j in range(N):
            acados_solver.set(j, "yref", np.zeros(9))
            acados_solver.set(j, "p", pobs)
            acados_solver.constraints_set(j, "lh", robs)
        acados_solver.set(N, "yref", np.zeros(8))
        acados_solver.set(N, "p", pobs)
        t = time.time()
        status = acados_solver.solve()
        if status != 0:
            print("acados returned status {} in closed loop iteration {}.".format(status, i))
        elapsed = time.time() - t
        tcomp_sum += elapsed
        tcomp_max = max(tcomp_max, elapsed)
        x0, u0 = acados_solver.get(0, "x"), acados_solver.get(0, "u")
        simX[i], simU[i] = x0, u0
        simError[i, 0], simError[i, 1] = x0[3], x0[2]
        if i > 400:
            nstat += 1
            psi_mae += abs(x0[3]); ye_mae += abs(x0[2]); psi_mse += x0[3] ** 2; ye_mse += x0[2] ** 2
        x0 = acados_solver.get(1, "x")
        acados_solver.set(0, "lbx", x0)
        acados_solver.set(0, "ubx", x0)
    clear = min(np.hypot(simX[:, 5] - ox, simX[:, 6] - oy).min() - 1.5 for ox, oy in zip(obsx, obsy))
    print("minimum clearance to an obstacle: %.3f m (nominal 0.2 from lsh)" % clear)
    if nstat:
        print("psi MAE %.4f  ye MAE %.4f  psi MSE %.5f  ye MSE %.5f" % (psi_mae / nstat, ye_mae / nstat, psi_mse / nstat, ye_mse / nstat))
    return tcomp_sum / Nsim, tcomp_max


def usv_pf_ca(ticks):
    Tf, N = 1.0, 100
    constraint, model, acados_solver = acados_settings(Tf, N, name="usv_model_pf_ca")
    nx, nu = model.x.size()[0], model.U.size()[0]
    Nsim = ticks or int(30.0 * N / Tf)
    simX = np.ndarray((Nsim, nx))
    obsx, obsy, radius = np.array([3, 4, 3.7, 4.2]), np.array([2, 8, 16, 20]), np.array([0.5] * 4)
    pobs, robs = np.zeros(8), np.zeros(4)
    x1, y1, x2, y2 = 4.0, -5.0, 4.0, 25.0
    ak = np.arctan2(y2 - y1, x2 - x1)
    nedx = nedy = psi = 0.0
    ye = -(nedx - x1) * np.sin(ak) + (nedy - y1) * np.cos(ak)
    x0 = np.array([psi, np.sin(psi), np.cos(psi), 0.001, 0, 0, ye, x1, y1, ak, nedx, nedy, 0, 0])
    acados_solver.set(0, "lbx", x0)
    acados_solver.set(0, "ubx", x0)
    tcomp_sum = tcomp_max = 0.0
    for i in range(Nsim):
        for ii in range(4):
            pobs[2 * ii], pobs[2 * ii + 1], robs[ii] = obsx[ii], obsy[ii], radius[ii] + 0.2
        for j in range(N):
            yref = np.zeros(16)
            yref[1], yref[2], yref[3] = np.sin(ak), np.cos(ak), 0.7
            acados_solver.set(j, "yref", yref)
            acados_solver.set(j, "p", pobs)
            acados_solver.constraints_set(j, "lh", robs)
        acados_solver.set(N, "yref", yref[:14])
        acados_solver.set(N, "p", pobs)
        t = time.time()
        status = acados_solver.solve()
        if status != 0:
            print("acados returned status {} in closed loop iteration {}.".format(status, i))
        elapsed = time.time() - t
        tcomp_sum += elapsed
        tcomp_max = max(tcomp_max, elapsed)
        simX[i] = acados_solver.get(0, "x")
        x0 = acados_solver.get(1, "x")
        acados_solver.set(0, "lbx", x0)
        acados_solver.set(0, "ubx", x0)
    clear = min(np.hypot(simX[:, 10] - ox, simX[:, 11] - oy).min() - 0.7 for ox, oy in zip(obsx, obsy))
    print("final surge speed u = %.4f (reference 0.7), minimum margin to the keep-out circles %.3f m" % (simX[-1, 3], clear))
    return tcomp_sum / Nsim, tcomp_max


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--variant", default="usv_guidance_ca1", choices=["usv_acados", "usv_guidance_ca1", "usv_pf_ca"])
    ap.add_argument("--ticks", type=int, default=0, help="0 = the reference's simulation length")
    a = ap.parse_args()
    avg, mx = {"usv_acados": usv_acados, "usv_guidance_ca1": usv_guidance_ca1, "usv_pf_ca": usv_pf_ca}[a.variant](a.ticks)
    print("Average computation time: {}".format(avg))
    print("Maximum computation time: {}".format(mx))
