"""The WIDE mapping of the QP kernel (qp_ipm.hpp: ONE instance per wave - the four rows of the wave share out the stage-local row work
of four consecutive stages, the Riccati / forward recursion runs in all rows alike) on the lane emulator, a whole wave of 64 fibers
per instance: iterates, statuses, iteration counts, residuals, slacks and multipliers equal the 16-lane sweeps' BIT FOR BIT (every
sum is taken in the same order), with and without the work queue, horizons that are and are not a multiple of the block of four."""
import ctypes as C

import numpy as np
import pytest

from mpc_collisionavoidance_amd import _capi, scenario, usv_models
from tests.test_emu_kernels import emu_rti, _d


# (K = 10 / 11 for usv_model_pf_ca, 15 / 16 for usv_model_guidance_ca1: box rows that do not fit the idle obstacle lanes - the two-pass form)
@pytest.mark.parametrize("name,N,K,rows", [("usv_model_pf_ca", 8, 3, 2), ("usv_model_pf_ca", 7, 4, 0), ("usv_model_pf_ca", 6, 9, 2),
                                           ("usv_model_pf_ca", 5, 1, 1), ("usv_model_guidance_ca1", 7, 8, 2), ("usv_model_guidance_ca1", 6, 3, 0),
                                           ("usv_model_pf_ca", 7, 10, 2), ("usv_model_pf_ca", 6, 11, 0), ("usv_model_pf_ca", 5, 10, 1),
                                           ("usv_model_guidance_ca1", 6, 16, 2), ("usv_model_guidance_ca1", 5, 15, 0),
                                           ("usv_model", 7, 0, 2), ("usv_model", 6, 0, 0),   # (no obstacle rows: box rows in planes of their own)
                                           # two obstacle chunks (K = 17 .. 32; BASELINE configs[4]'s OCP has K = 20): one row pass (K = 20 + 4 box rows ride in
                                           # the second chunk's idle lanes) and two (usv_model_pf_ca at K = 26: 10 + 7 rows do not fit)
                                           ("usv_model_pf_ca", 6, 20, 2), ("usv_model_pf_ca", 5, 26, 0), ("usv_model_guidance_ca1", 6, 20, 2),
                                           ("usv_model_guidance_ca1", 5, 32, 0), ("usv_model_pf_ca", 9, 17, 1),
                                           # obstacle rows that leave the box rows no idle lanes: box rows in planes of their own (unpacked)
                                           ("usv_model_pf_ca", 6, 15, 2), ("usv_model_pf_ca", 5, 32, 0), ("usv_model_guidance_ca1", 6, 16, 1)])
@pytest.mark.parametrize("lds,ww", [(1, 1), (0, 1), (1, 4), (0, 2), (1, 2), (0, 4)])
def test_wide_mapping_equals_the_16_lane_sweeps_bit_for_bit(emu, name, N, K, rows, lds, ww):
    """lds = 1: the solver's planes in (emulated) LDS; 0: in HBM - the variant for horizons that do not fit a CU's LDS (the row planes
    of the next block and the recursion's planes of the next stage asked for ahead of their use).
    ww: waves per instance (qp_ipm.hpp WW) - a workgroup of ww waves shares out the row work of 4 ww consecutive stages."""
    B = 5
    wl = scenario.make_batch(name, N, K, B, dt=0.05, seed=17, generator="survey", sim_steps=scenario.BENCH_SIM_STEPS[name], clip_time=0.1)
    ocp = usv_models.make_ocp(name, N * 0.05, N, K)
    ocp.solver_options.sim_method_num_steps = scenario.BENCH_SIM_STEPS[name]
    desc = _capi.desc_from_ocp(ocp, batch=B)
    soft = name == "usv_model_guidance_ca1"
    nlam = 2 * (desc.nbu + desc.nbx + K + (K if soft else 0))
    emu.usv_emu_set_wide.argtypes = [C.c_int]
    emu.usv_emu_set_mode.argtypes = [C.c_int, C.c_long]
    emu.usv_emu_set_export.argtypes = [_capi._dp, _capi._dp]
    emu.usv_emu_set_export.restype = None
    out = []
    try:
        emu.usv_emu_set_mode(lds, rows)
        for wide in (0, ww):
            emu.usv_emu_set_wide(wide)
            lam, t = np.zeros((B, N + 1, nlam)), np.zeros((B, N + 1, nlam))
            emu.usv_emu_set_export(_d(lam), _d(t))
            r = emu_rti(emu, desc, wl, wl["x_init"], wl["u_init"])
            r2 = emu_rti(emu, desc, wl, r["x"], r["u"])
            if wide:
                emu.usv_emu_wide_runs.restype = C.c_long
                assert emu.usv_emu_wide_runs() == 2 * (rows if 0 < rows < B else (B + 3) // 4 * 4)   # (the wide sweeps did run)
            out.append((r2["x"], r2["u"], r2["status"], r2["qp_status"], r2["qp_iter"], r2["sl"], r2["su"], r2["pi"], r2["res"], lam.copy(), t.copy()))
    finally:
        emu.usv_emu_set_wide(0)
        emu.usv_emu_set_mode(0, 2)
        emu.usv_emu_set_export(None, None)
    assert (out[0][2] == 0).any() and out[0][9].any() and out[0][4].max() >= 3
    for n, (a, b) in enumerate(zip(out[0], out[1])):
        assert np.array_equal(a, b), (n, np.abs(np.asarray(a, float) - np.asarray(b, float)).max())


@pytest.mark.parametrize("name,N,K", [("usv_model_pf_ca", 8, 3), ("usv_model_pf_ca", 7, 10), ("usv_model_guidance_ca1", 7, 8),
                                      ("usv_model_guidance_ca1", 6, 16), ("usv_model", 7, 0)])
@pytest.mark.parametrize("aux", [0, 1])
@pytest.mark.parametrize("hand_it", [1, 3])
def test_handed_over_instances_finish_with_the_same_bits(emu, name, N, K, aux, hand_it):
    """Option "handover_iter" (QpIpm::suspend / solve phase 3, usvmpc.hip usv_qp_resume): once the queue of a launch is empty a 16-lane row
    whose instance has passed hand_it IPM iterations leaves it - state in the workspace planes, four scalars in its record - and the
    follow-up pass finishes it on the WIDE mapping over the same planes.  Scheduling only: every output equals the plain run's bit for
    bit (aux = 1: the suspending row keeps its aux plane in LDS and writes it out on the way)."""
    B = 6
    wl = scenario.make_batch(name, N, K, B, dt=0.05, seed=23, generator="survey", sim_steps=scenario.BENCH_SIM_STEPS[name], clip_time=0.1)
    ocp = usv_models.make_ocp(name, N * 0.05, N, K)
    ocp.solver_options.sim_method_num_steps = scenario.BENCH_SIM_STEPS[name]
    desc = _capi.desc_from_ocp(ocp, batch=B)
    soft = name == "usv_model_guidance_ca1"
    nlam = 2 * (desc.nbu + desc.nbx + K + (K if soft else 0))
    emu.usv_emu_set_mode.argtypes = [C.c_int, C.c_long]
    emu.usv_emu_set_export.argtypes = [_capi._dp, _capi._dp]
    emu.usv_emu_set_export.restype = None
    emu.usv_emu_set_handover.argtypes = [C.c_int]
    emu.usv_emu_set_handover_lds.argtypes = [C.c_int]
    emu.usv_emu_handed.restype = C.c_long
    emu.usv_emu_set_aux.argtypes = [C.c_int]
    out = []
    try:
        emu.usv_emu_set_mode(0, 2)          # two persistent rows, the other four instances through the queue
        emu.usv_emu_set_aux(aux if K > 0 else 0)
        # (plain; hand-over with the follow-up pass over the planes in "HBM"; ... with the planes copied into LDS first - QpIpm::copy_in;
        # ... and from a launch WITHOUT a queue - every instance resident from the start, the case of mid-size batches)
        for hand, lds, rows in ((0, 0, 2), (hand_it, 0, 2), (hand_it, 1, 2), (hand_it, 1, 0)):
            emu.usv_emu_set_mode(0, rows)
            emu.usv_emu_set_handover(hand)
            emu.usv_emu_set_handover_lds(lds)
            lam, t = np.zeros((B, N + 1, nlam)), np.zeros((B, N + 1, nlam))
            emu.usv_emu_set_export(_d(lam), _d(t))
            r = emu_rti(emu, desc, wl, wl["x_init"], wl["u_init"])
            r2 = emu_rti(emu, desc, wl, r["x"], r["u"])
            if hand:
                assert emu.usv_emu_handed() >= 2, emu.usv_emu_handed()   # (the instances the two rows were on when the queue ran dry)
            out.append((r2["x"], r2["u"], r2["status"], r2["qp_status"], r2["qp_iter"], r2["sl"], r2["su"], r2["pi"], r2["res"], lam.copy(), t.copy()))
    finally:
        emu.usv_emu_set_handover(0)
        emu.usv_emu_set_handover_lds(0)
        emu.usv_emu_set_mode(0, 2)
        emu.usv_emu_set_aux(0)
        emu.usv_emu_set_export(None, None)
    assert (out[0][2] == 0).any() and out[0][4].max() >= hand_it
    for v in (1, 2, 3):
        for n, (a, b) in enumerate(zip(out[0], out[v])):
            assert np.array_equal(a, b), (v, n, np.abs(np.asarray(a, float) - np.asarray(b, float)).max())
