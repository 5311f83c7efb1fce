"""Batched counterpart of the reference's obstacle-avoidance ROS node around the solver
(class NMPC, /root/reference/catkin_ws/src/nmpc_ca/src/nmpc_guidance_ca1.cpp): waypoint manager,
LiDAR obstacle selection / body->NED transform, x0 assembly, and the published set-points.  All
arithmetic runs on the device (csrc/guidance.hpp); this class only moves arrays."""
import ctypes as C

import numpy as np

from . import _capi


class GuidanceFrontEnd:
    def __init__(self, solver):
        if solver.ocp.model.name != "usv_model_guidance_ca1":
            raise Exception("the guidance front end belongs to usv_model_guidance_ca1")
        self.s = solver
        self.B = solver.B
        self._lib = solver._lib

    def reset(self, waypoints, psi):
        """New waypoint list: waypoints [B, npts, 2] (or [npts, 2] for all), psi [B]."""
        w = np.ascontiguousarray(waypoints, dtype=np.float64)
        if w.ndim == 2:
            w = np.tile(w[None], (self.B, 1, 1))
        w = np.ascontiguousarray(w.reshape(self.B, -1))
        psi = np.ascontiguousarray(np.broadcast_to(np.asarray(psi, dtype=np.float64), (self.B,)))
        self.s._check(self._lib.usvmpc_guidance_reset(self.s._h, w.ctypes.data_as(_capi._dp), w.shape[1] // 2,
                                                      psi.ctypes.data_as(_capi._dp)))

    def sense(self, pose, world, max_radius=100.0, fetch=False):
        """The obstacle simulator's simulate(): world [B,L,3] (or [L,3] for all) NED (X, Y, R), pose [B,3]
        (nedx, nedy, yaw).  The visible obstacles, in the body frame, stay on the device for the next
        prepare(..., obstacles=None); fetch=True also returns (obstacles [B,64,3], n [B])."""
        p = np.ascontiguousarray(pose, dtype=np.float64).reshape(self.B, 3)
        w = np.asarray(world, dtype=np.float64)
        if w.ndim == 2:
            w = np.tile(w[None], (self.B, 1, 1))
        w = np.ascontiguousarray(w.reshape(self.B, -1, 3))
        o = np.zeros((self.B, 64, 3)) if fetch else None
        n = np.zeros(self.B, dtype=np.int32) if fetch else None
        self.s._check(self._lib.usvmpc_guidance_sense(self.s._h, p.ctypes.data_as(_capi._dp), w.ctypes.data_as(_capi._dp),
                                                      w.shape[1], float(max_radius),
                                                      o.ctypes.data_as(_capi._dp) if fetch else None,
                                                      n.ctypes.data_as(_capi._ip) if fetch else None))
        return (o, n) if fetch else None

    def prepare(self, vel_uv, pose, obstacles=None, n_obstacles=None):
        """vel_uv [B,2], pose [B,3] (nedx, nedy, psi), obstacles [B,L,3] body (x, y, R), n_obstacles [B];
        obstacles=None: the lists sense() left on the device."""
        v = np.ascontiguousarray(vel_uv, dtype=np.float64).reshape(self.B, 2)
        p = np.ascontiguousarray(pose, dtype=np.float64).reshape(self.B, 3)
        if obstacles is None:
            self.s._check(self._lib.usvmpc_guidance_prepare(self.s._h, v.ctypes.data_as(_capi._dp), p.ctypes.data_as(_capi._dp),
                                                            None, None, 0))
            return
        o = np.ascontiguousarray(obstacles, dtype=np.float64).reshape(self.B, -1, 3)
        n = np.ascontiguousarray(n_obstacles, dtype=np.int32).reshape(self.B)
        self.s._check(self._lib.usvmpc_guidance_prepare(self.s._h, v.ctypes.data_as(_capi._dp), p.ctypes.data_as(_capi._dp),
                                                        o.ctypes.data_as(_capi._dp), n.ctypes.data_as(_capi._ip), o.shape[1]))

    def publish(self):
        """After solve(): dict(heading, r, speed, ye, active) - the node's published set-points."""
        h, r, sp, ye = (np.zeros(self.B) for _ in range(4))
        act = np.zeros(self.B, dtype=np.int32)
        self.s._check(self._lib.usvmpc_guidance_publish(self.s._h, h.ctypes.data_as(_capi._dp), r.ctypes.data_as(_capi._dp),
                                                        sp.ctypes.data_as(_capi._dp), ye.ctypes.data_as(_capi._dp),
                                                        act.ctypes.data_as(_capi._ip)))
        return dict(heading=h, r=r, speed=sp, ye=ye, active=act)

    def state(self):
        k = np.zeros(self.B, dtype=np.int32)
        pp = np.zeros(self.B, dtype=np.float32)
        self.s._check(self._lib.usvmpc_guidance_state(self.s._h, k.ctypes.data_as(_capi._ip),
                                                      pp.ctypes.data_as(C.POINTER(C.c_float))))
        return k, pp
