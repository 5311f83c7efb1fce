"""HPIPM's conditional predictor-corrector (option "cond_pred_corr", on in the default QP solver profile: include/usvmpc.h USVMPC_HPIPM_*;
qp_ipm.hpp QpIpm::solve) is built into EVERY mapping of the QP kernel.  Here, on the lane emulator, with a factor low enough that corrected
steps are refused all the time: the latency mapping (one / two / four waves per instance, planes in LDS and in HBM), the hand-over to the
follow-up pass (whose record carries whether the pending step is a centring-only one) and the launches of a full SQP return the 16-lane
sweeps' bits, and those differ from the run without the option (the refusals did happen)."""
import ctypes as C

import numpy as np
import pytest

from mpc_collisionavoidance_amd import _capi, scenario, usv_models
from tests.test_emu_kernels import emu_rti, _d

FACTOR = 0.6   # (HPIPM: 2; here the corrected step is refused whenever it leaves mu above 0.6 x the predictor's: most iterations)


def _setup(emu, name, N, K, B, seed):
    wl = scenario.make_batch(name, N, K, B, dt=0.05, seed=seed, generator="survey", sim_steps=scenario.BENCH_SIM_STEPS[name], clip_time=0.1)
    ocp = usv_models.make_ocp(name, N * 0.05, N, K)
    ocp.solver_options.sim_method_num_steps = scenario.BENCH_SIM_STEPS[name]
    desc = _capi.desc_from_ocp(ocp, batch=B)
    soft = name == "usv_model_guidance_ca1"
    nlam = 2 * (desc.nbu + desc.nbx + K + (K if soft else 0))
    emu.usv_emu_set_wide.argtypes = [C.c_int]
    emu.usv_emu_set_mode.argtypes = [C.c_int, C.c_long]
    emu.usv_emu_set_export.argtypes = [_capi._dp, _capi._dp]
    emu.usv_emu_set_export.restype = None
    emu.usv_emu_set_cpc.argtypes = [C.c_int, C.c_double]
    emu.usv_emu_set_handover.argtypes = [C.c_int]
    emu.usv_emu_set_handover_lds.argtypes = [C.c_int]
    emu.usv_emu_handed.restype = C.c_long
    emu.usv_emu_set_aux.argtypes = [C.c_int]
    return wl, desc, nlam


def _two_ticks(emu, desc, wl, B, N, nlam):
    lam, t = np.zeros((B, N + 1, nlam)), np.zeros((B, N + 1, nlam))
    emu.usv_emu_set_export(_d(lam), _d(t))
    r = emu_rti(emu, desc, wl, wl["x_init"], wl["u_init"])
    r2 = emu_rti(emu, desc, wl, r["x"], r["u"])
    return (r2["x"], r2["u"], r2["status"], r2["qp_status"], r2["qp_iter"], r2["sl"], r2["su"], r2["pi"], r2["res"], lam.copy(), t.copy())


def _reset(emu):
    emu.usv_emu_set_cpc(-1, 2.0)
    emu.usv_emu_set_wide(0)
    emu.usv_emu_set_mode(0, 2)
    emu.usv_emu_set_handover(0)
    emu.usv_emu_set_handover_lds(0)
    emu.usv_emu_set_aux(0)
    emu.usv_emu_set_export(None, None)


@pytest.mark.parametrize("name,N,K,rows", [("usv_model_pf_ca", 8, 3, 2), ("usv_model_pf_ca", 7, 10, 0), ("usv_model_guidance_ca1", 7, 8, 2),
                                           ("usv_model_guidance_ca1", 6, 16, 0), ("usv_model", 7, 0, 2), ("usv_model_pf_ca", 6, 20, 2),
                                           ("usv_model_guidance_ca1", 5, 32, 0), ("usv_model_pf_ca", 6, 15, 1)])
@pytest.mark.parametrize("lds,ww", [(1, 1), (0, 1), (1, 4), (0, 4), (0, 2)])
def test_refused_steps_on_the_latency_mapping_equal_the_16_lane_sweeps(emu, name, N, K, rows, lds, ww):
    B = 5
    wl, desc, nlam = _setup(emu, name, N, K, B, 17)
    try:
        emu.usv_emu_set_mode(lds, rows)
        emu.usv_emu_set_cpc(0, 2.0)
        off = _two_ticks(emu, desc, wl, B, N, nlam)
        emu.usv_emu_set_cpc(1, FACTOR)
        out = []
        for wide in (0, ww):
            emu.usv_emu_set_wide(wide)
            out.append(_two_ticks(emu, desc, wl, B, N, nlam))
    finally:
        _reset(emu)
    assert (out[0][2] == 0).any() and out[0][4].max() >= 3
    assert not np.array_equal(off[0], out[0][0]) or not np.array_equal(off[4], out[0][4])   # steps were refused: another iteration path
    for n, (a, b) in enumerate(zip(out[0], out[1])):
        assert np.array_equal(a, b), (n, np.abs(np.asarray(a, float) - np.asarray(b, float)).max())


@pytest.mark.parametrize("name,N,K", [("usv_model_pf_ca", 8, 3), ("usv_model_pf_ca", 7, 10), ("usv_model_guidance_ca1", 7, 8), ("usv_model", 7, 0)])
@pytest.mark.parametrize("hand_it", [1, 2, 3])
def test_refused_steps_survive_the_hand_over(emu, name, N, K, hand_it):
    """The suspended solve's pending step may be a centring-only one: the follow-up pass replays it as such (the flag rides in the sign of
    the record's iteration count: QpIpm::suspend)."""
    B = 6
    wl, desc, nlam = _setup(emu, name, N, K, B, 23)
    out = []
    try:
        emu.usv_emu_set_cpc(1, FACTOR)
        emu.usv_emu_set_aux(1 if K > 0 else 0)
        for hand, lds, rows in ((0, 0, 2), (hand_it, 0, 2), (hand_it, 1, 2), (hand_it, 1, 0)):
            emu.usv_emu_set_mode(0, rows)
            emu.usv_emu_set_handover(hand)
            emu.usv_emu_set_handover_lds(lds)
            out.append(_two_ticks(emu, desc, wl, B, N, nlam))
            if hand:
                assert emu.usv_emu_handed() >= 2
    finally:
        _reset(emu)
    for v in (1, 2, 3):
        for n, (a, b) in enumerate(zip(out[0], out[v])):
            assert np.array_equal(a, b), (v, n, np.abs(np.asarray(a, float) - np.asarray(b, float)).max())


@pytest.mark.parametrize("name,N,K", [("usv_model_pf_ca", 6, 3), ("usv_model_guidance_ca1", 6, 4)])
def test_refused_steps_in_the_launches_of_a_full_sqp(emu, name, N, K):
    from tests.test_sqp_options import emu_sqp
    B = 4
    wl, desc, nlam = _setup(emu, name, N, K, B, 31)
    desc.nlp_max_iter = 6
    out = []
    try:
        emu.usv_emu_set_cpc(1, FACTOR)
        for wide in (0, 1):
            emu.usv_emu_set_wide(wide)
            r = emu_sqp(emu, desc, wl, wl["x_init"], wl["u_init"])
            out.append((r["x"], r["u"], r["status"], r["sqp_iter"], r["nlp_res"]))
    finally:
        _reset(emu)
    for n, (a, b) in enumerate(zip(out[0], out[1])):
        assert np.array_equal(a, b), n
