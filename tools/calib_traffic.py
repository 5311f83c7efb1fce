"""Run under `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE`: streams a known byte count through the
solver's plane-access path (kernel usv_calib_stream) so the counters can be calibrated."""
import ctypes as C, sys
sys.path.insert(0, ".")
from mpc_collisionavoidance_amd import BatchOcpSolver, usv_models
ocp = usv_models.make_ocp("usv_model_pf_ca", 0.4, 40, 10)
s = BatchOcpSolver(ocp, 65536)
for n in (64, 256):
    r, w = C.c_double(), C.c_double()
    s._check(s._lib.usvmpc_calibrate_traffic(s._h, n, C.byref(r), C.byref(w)))
    print("calib planes %d: read %.0f KiB written %.0f KiB" % (n, r.value / 1024, w.value / 1024), flush=True)
