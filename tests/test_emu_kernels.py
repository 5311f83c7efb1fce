"""Runs the UNMODIFIED HIP kernel bodies (csrc/linearize.hpp, csrc/qp_ipm.hpp) on the CPU through the
16-fiber lane emulator (tests/emu) and checks them against the oracle.  This pins the cross-lane
algorithm (DPP broadcast/rotate patterns, lane ownership, plane layout) without a GPU; the emulator
aborts on divergent control flow around a cross-lane op.  Test infrastructure only - the product
library is never routed through it."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from mpc_collisionavoidance_amd import _capi, scenario, usv_models
from tests import util

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU = os.path.join(ROOT, "tests", "emu")
CSRC = os.path.join(ROOT, "mpc_collisionavoidance_amd", "csrc")


def _d(a):
    return a.ctypes.data_as(_capi._dp)


def _i(a):
    return a.ctypes.data_as(_capi._ip)


def emu_rti(emu, desc, wl, x, u, dbg=False):
    B, N = x.shape[0], desc.N
    nx, K = x.shape[2], desc.K
    Bp = (B + 3) // 4 * 4
    sl, su, pi = np.zeros((B, N, max(K, 1))), np.zeros((B, N, max(K, 1))), np.zeros((B, N, nx))
    st, qs, qi, res = np.zeros(B, np.int32), np.zeros(B, np.int32), np.zeros(B, np.int32), np.zeros((B, 4))
    BAt = np.zeros((N, nx, Bp, 16)) if dbg else None
    rb0 = np.zeros((N, Bp, 16)) if dbg else None
    gq = np.zeros((N + 1, Bp, 16)) if dbg else None
    x, u = x.copy(), u.copy()
    rc = emu.usv_emu_solve(C.byref(desc), _d(x), _d(u), _d(wl["x0"]), _d(wl["yref"]), _d(wl["yref_e"]), _d(wl["p"]),
                           _d(wl["lh"]), _d(sl), _d(su), _d(pi), _i(st), _i(qs), _i(qi), _d(res),
                           _d(BAt) if dbg else None, _d(rb0) if dbg else None, _d(gq) if dbg else None)
    assert rc == 0
    return dict(x=x, u=u, sl=sl[:, :, :K], su=su[:, :, :K], pi=pi, status=st, qp_status=qs, qp_iter=qi, res=res,
                BAt=BAt, rb0=rb0, gq=gq)


CASES = [("usv_model", 8, 0, 3), ("usv_model_guidance_ca1", 8, 5, 3), ("usv_model_pf_ca", 8, 4, 5),
         ("usv_model_guidance_ca1", 6, 20, 2), ("usv_model_pf_ca", 6, 18, 2),
         # box rows packed into the idle obstacle lanes (+ dense plane): 4 slots + 3 dense, no fit, full chunk, 6 + 1
         ("usv_model_pf_ca", 5, 12, 2), ("usv_model_pf_ca", 5, 14, 2), ("usv_model_guidance_ca1", 5, 16, 2),
         ("usv_model_pf_ca", 5, 10, 2)]


@pytest.mark.parametrize("name,N,K,B", CASES)
def test_kernel_bodies_match_oracle(oracle, emu, name, N, K, B):
    ocp, wl = util.make(name, N, K, B, seed=31)
    desc = _capi.desc_from_ocp(ocp, batch=B)
    spec = util.oracle_spec(oracle, name, N, scenario.DT[name], K)
    xo, uo = wl["x_init"].copy(), wl["u_init"].copy()
    xe, ue = xo.copy(), uo.copy()
    for it in range(2):
        r = emu_rti(emu, desc, wl, xe, ue)
        xe, ue = r["x"], r["u"]
        xo, uo, sto, ito = util.oracle_rti(oracle, spec, wl, xo, uo)
        assert np.array_equal(r["status"], sto)
        assert np.abs(r["qp_iter"] - ito).max() <= 1
        # classical (kernel) vs square-root (oracle) Riccati: 1e-8 relative is ample for FP64
        assert util.rel_err(xe, xo) < 1e-8 and util.rel_err(ue, uo) < 1e-8, (util.rel_err(xe, xo), util.rel_err(ue, uo))
        assert (r["qp_status"] == 0).all()


@pytest.mark.parametrize("name,N,K", [("usv_model", 5, 0), ("usv_model_guidance_ca1", 5, 3), ("usv_model_pf_ca", 5, 3)])
def test_linearize_planes_match_oracle(oracle, emu, name, N, K):
    """Lane r of the (b, k) group must hold row r of [B_k A_k]' in the BAt planes, the x lanes the
    dynamics residual b_k, and every lane its entry of the cost gradient."""
    B = 3  # not a multiple of 4: exercises the padded groups
    ocp, wl = util.make(name, N, K, B, seed=8)
    desc = _capi.desc_from_ocp(ocp, batch=B)
    spec = util.oracle_spec(oracle, name, N, scenario.DT[name], K)
    r = emu_rti(emu, desc, wl, wl["x_init"], wl["u_init"], dbg=True)
    nx, nu = wl["nx"], wl["nu"]
    for b in range(B):
        qp, _ = oracle.linearize_and_solve(spec, wl["x_init"][b], wl["u_init"][b], wl["x0"][b], wl["yref"][b],
                                           wl["yref_e"][b], wl["p"][b], wl["lh"][b], solve=False)
        for k in range(N):
            BA = np.hstack([qp["B"][k], qp["A"][k]])          # nx x nz
            got = r["BAt"][k, :, b, :nu + nx]                    # [j, lane r] = BAt[r][j]
            # planes that are structurally unit vectors (csrc/models.hpp OUT_UNIT) are never written
            unit = {"usv_model": [], "usv_model_guidance_ca1": [0, 1], "usv_model_pf_ca": [7, 8, 9]}[name]
            live = [j for j in range(nx) if j not in unit]
            assert np.allclose(got[live], BA[live], rtol=1e-12, atol=1e-14)
            for j in unit:
                e = np.zeros(nu + nx); e[nu + j] = 1.0
                assert np.array_equal(BA[j], e) and not got[j].any()
            assert np.allclose(r["rb0"][k, b, nu:nu + nx], qp["b"][k], rtol=1e-12, atol=1e-14)
            # the gradient plane holds the reference part -M yref only; the kernel adds H (zbar + z) itself
            zbar = np.concatenate([wl["u_init"][b, k], wl["x_init"][b, k]])
            assert np.allclose(r["gq"][k, b, :nu + nx] + qp["H"][k] @ zbar, qp["g"][k], rtol=1e-12, atol=1e-12)
        zN = np.concatenate([np.zeros(nu), wl["x_init"][b, N]])
        assert np.allclose((r["gq"][N, b, :nu + nx] + qp["H"][N] @ zN)[nu:], qp["g"][N][nu:], rtol=1e-12, atol=1e-12)
        assert np.all(r["BAt"][:, :, b, nu + nx:] == 0.0)      # idle lanes stay zero


def test_soft_slacks_and_multipliers_exported(oracle, emu):
    name, N, K, B = "usv_model_guidance_ca1", 10, 4, 2
    ocp, wl = util.make(name, N, K, B, seed=77)
    desc = _capi.desc_from_ocp(ocp, batch=B)
    spec = util.oracle_spec(oracle, name, N, 0.05, K)
    r = emu_rti(emu, desc, wl, wl["x_init"], wl["u_init"])
    for b in range(B):
        o = oracle.rti(spec, wl["x_init"][b], wl["u_init"][b], wl["x0"][b], wl["yref"][b], wl["yref_e"][b], wl["p"][b], wl["lh"][b])
        assert np.allclose(r["sl"][b], o["sl"], atol=1e-7) and np.allclose(r["su"][b], o["su"], atol=1e-7)
        assert np.allclose(r["pi"][b], o["pi"], rtol=1e-6, atol=1e-7)
        # stage 0: the rows take no part in the QP, their slacks are the constants max(lsh, lh - h(x0)) = lsh here
        assert np.allclose(r["sl"][b][0], -0.2) and np.allclose(r["su"][b][0], 0.0)


@pytest.mark.parametrize("name,K", [("usv_model_guidance_ca1", 5), ("usv_model_pf_ca", 20)])
def test_stage_dependent_obstacle_set(oracle, emu, name, K):
    """Moving obstacles (p and lh differ from stage to stage): the lineariser hands them to the QP kernel as
    planes instead of the per-lane constants of the stage-independent case."""
    N, B = 6, 3
    ocp, wl = util.make(name, N, K, B, seed=17, moving=True)
    assert np.ptp(wl["p"], axis=1).max() > 0
    desc = _capi.desc_from_ocp(ocp, batch=B)
    spec = util.oracle_spec(oracle, name, N, scenario.DT[name], K)
    r = emu_rti(emu, desc, wl, wl["x_init"], wl["u_init"])
    xo, uo, sto, ito = util.oracle_rti(oracle, spec, wl, wl["x_init"], wl["u_init"])
    assert np.array_equal(r["status"], sto)
    ok = sto == 0
    assert util.rel_err(r["x"][ok], xo[ok]) < 1e-8 and util.rel_err(r["u"][ok], uo[ok]) < 1e-8


def test_stage0_obstacle_rows(oracle, emu):
    """acados applies the nh rows at stages 0..N-1.  At stage 0 they depend on no free variable (x_0 is pinned to x0), so
    they are not rows of the QP - but x0 inside a HARD keep-out circle makes acados' QP infeasible: status 4 and an
    untouched iterate (here without iterating, qp_status 4); a SOFT stage-0 row reports the constant slack
    max(lsh, lh - h(x0)).  Kernel bodies and oracle agree on both."""
    # hard rows: instance 1 starts inside its first circle, the other instances do not
    name, N, K, B = "usv_model_pf_ca", 6, 4, 3
    ocp, wl = util.make(name, N, K, B, seed=31)
    wl["p"][1, :, 0:2] = wl["x0"][1, 10:12] + np.array([0.3, 0.0])     # centre 0.3 m away, lh >= 1.0
    desc = _capi.desc_from_ocp(ocp, batch=B)
    spec = util.oracle_spec(oracle, name, N, scenario.DT[name], K)
    r = emu_rti(emu, desc, wl, wl["x_init"], wl["u_init"])
    xo, uo, sto, ito = util.oracle_rti(oracle, spec, wl, wl["x_init"], wl["u_init"])
    assert list(r["status"]) == list(sto) == [0, 4, 0]
    assert r["qp_status"][1] == 4 and r["qp_iter"][1] == 0 and ito[1] == 0
    assert np.array_equal(r["x"][1], wl["x_init"][1]) and np.array_equal(xo[1], wl["x_init"][1])
    assert util.rel_err(r["x"][[0, 2]], xo[[0, 2]]) < 1e-8
    # soft rows: the stage-0 slacks
    name, N, K, B = "usv_model_guidance_ca1", 6, 5, 3
    ocp, wl = util.make(name, N, K, B, seed=31)
    wl["p"][2, :, 0:2] = wl["x0"][2, 5:7] + np.array([0.2, 0.1])       # instance 2 starts inside obstacle 0
    desc = _capi.desc_from_ocp(ocp, batch=B)
    spec = util.oracle_spec(oracle, name, N, scenario.DT[name], K)
    r = emu_rti(emu, desc, wl, wl["x_init"], wl["u_init"])
    for b in range(B):
        ro = oracle.rti(spec, wl["x_init"][b], wl["u_init"][b], wl["x0"][b], wl["yref"][b], wl["yref_e"][b], wl["p"][b], wl["lh"][b])
        assert r["status"][b] == ro["status"] == 0
        assert np.allclose(r["sl"][b], ro["sl"], rtol=0, atol=1e-8) and np.allclose(r["su"][b], ro["su"], rtol=0, atol=1e-8)
        h0 = np.hypot(*(wl["x0"][b, 5:7] - wl["p"][b, 0].reshape(K, 2)).T)
        assert np.allclose(ro["sl"][0], np.maximum(-0.2, wl["lh"][b, 0] - h0), rtol=0, atol=1e-12)
    assert r["sl"][2, 0, 0] > 0.5       # the violated row's slack is the violation itself


@pytest.mark.parametrize("name,N,K", [("usv_model_pf_ca", 8, 4), ("usv_model_guidance_ca1", 7, 5), ("usv_model", 6, 0)])
def test_scheduling_and_workspace_placement_do_not_change_results(emu, name, N, K):
    """The same arithmetic whatever carries it: one row per instance, two persistent rows pulling instances from the queue,
    and the workspace in (emulated) LDS instead of HBM return bit-identical iterates, statuses and iteration counts."""
    B = 7
    ocp, wl = util.make(name, N, K, B, seed=3)
    desc = _capi.desc_from_ocp(ocp, batch=B)
    emu.usv_emu_set_mode.argtypes = [C.c_int, C.c_long]
    out = []
    try:
        for lds, rows in ((0, 0), (0, 2), (1, 2), (1, 0)):
            emu.usv_emu_set_mode(lds, rows)
            r = emu_rti(emu, desc, wl, wl["x_init"], wl["u_init"])
            r2 = emu_rti(emu, desc, wl, r["x"], r["u"])
            out.append((r2["x"], r2["u"], r2["status"], r2["qp_iter"], r2["sl"], r2["pi"]))
    finally:
        emu.usv_emu_set_mode(0, 2)
    for o in out[1:]:
        for a, b in zip(o, out[0]):
            assert np.array_equal(a, b)


@pytest.mark.parametrize("name,N,K", [("usv_model_pf_ca", 8, 4), ("usv_model_pf_ca", 6, 9), ("usv_model_pf_ca", 5, 20),
                                      ("usv_model_guidance_ca1", 7, 10), ("usv_model_guidance_ca1", 5, 15), ("usv_model_guidance_ca1", 5, 19)])
def test_one_row_pass_equals_two(emu, name, N, K):
    """Box rows processed where they are stored - as rows of the last obstacle chunk (MERGE: whenever all of them ride in its idle
    lanes) - against the two-pass form that gathers them to their variables' lanes: the same rows, the same arithmetic per row;
    only the lane in which a row's share of the complementarity sums is accumulated differs, so the comparison is to rounding,
    statuses and iteration counts exactly."""
    B = 6
    ocp, wl = util.make(name, N, K, B, seed=11)
    desc = _capi.desc_from_ocp(ocp, batch=B)
    emu.usv_emu_set_merge.argtypes = [C.c_int]
    out = []
    try:
        for merge in (0, 1):
            emu.usv_emu_set_merge(merge)
            r = emu_rti(emu, desc, wl, wl["x_init"], wl["u_init"])
            r2 = emu_rti(emu, desc, wl, r["x"], r["u"])
            out.append(r2)
    finally:
        emu.usv_emu_set_merge(1)
    a, b = out
    assert np.array_equal(a["status"], b["status"]) and np.array_equal(a["qp_iter"], b["qp_iter"])
    for f in ("x", "u", "pi", "sl", "su"):
        assert util.rel_err(a[f], b[f]) <= 1e-9, (f, util.rel_err(a[f], b[f]))


@pytest.mark.parametrize("name,N,K", [("usv_model_pf_ca", 8, 10), ("usv_model_pf_ca", 6, 9), ("usv_model_pf_ca", 7, 12),
                                      ("usv_model_guidance_ca1", 7, 10), ("usv_model_guidance_ca1", 6, 3)])
def test_aux_plane_in_lds_equals_aux_plane_in_hbm(emu, name, N, K):
    """AUXLDS: the aux plane (dense box rows, linearisation point, r_g, l_u) kept in the wave's LDS for the whole launch - the same
    values at the same places of the same arithmetic, so iterates, statuses, iteration counts, slacks and multipliers are
    bit-identical; the read-back (export_rows) finds the dense box rows in the HBM plane finish() copies them to.
    K = 10: one dense row + six slot rows (the headline layout); 9: no dense row, one row pass; 12: two dense rows."""
    B = 6
    wl = scenario.make_batch(name, N, K, B, dt=0.05, seed=13, generator="survey", sim_steps=scenario.BENCH_SIM_STEPS[name], clip_time=0.1)
    from mpc_collisionavoidance_amd import usv_models
    ocp = usv_models.make_ocp(name, N * 0.05, N, K)
    ocp.solver_options.sim_method_num_steps = scenario.BENCH_SIM_STEPS[name]
    desc = _capi.desc_from_ocp(ocp, batch=B)
    soft = name == "usv_model_guidance_ca1"
    nlam = 2 * (desc.nbu + desc.nbx + K + (K if soft else 0))
    emu.usv_emu_set_aux.argtypes = [C.c_int]
    emu.usv_emu_set_export.argtypes = [_capi._dp, _capi._dp]
    emu.usv_emu_set_export.restype = None
    out = []
    try:
        for aux in (0, 1):
            emu.usv_emu_set_aux(aux)
            lam, t = np.zeros((B, N + 1, nlam)), np.zeros((B, N + 1, nlam))
            emu.usv_emu_set_export(_d(lam), _d(t))
            r = emu_rti(emu, desc, wl, wl["x_init"], wl["u_init"])
            r2 = emu_rti(emu, desc, wl, r["x"], r["u"])
            out.append((r2["x"], r2["u"], r2["status"], r2["qp_iter"], r2["sl"], r2["su"], r2["pi"], r2["res"], lam.copy(), t.copy()))
    finally:
        emu.usv_emu_set_aux(0)
        emu.usv_emu_set_export(None, None)
    assert (out[0][2] == 0).any() and out[0][8].any()
    for a, b in zip(out[0], out[1]):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("name,N,K", [("usv_model_pf_ca", 6, 3), ("usv_model_guidance_ca1", 5, 2), ("usv_model", 7, 0)])
def test_pipelined_lineariser_modes_cover_every_group_exactly(emu, name, N, K):
    """linearize.hpp MODE 1 (ahead of time, beside a running QP launch) + MODE 2 (fix-up) must leave exactly what MODE 0 writes, for any
    pattern of finished / unfinished instances and any pair of maps: a group is linearised ahead only if its own instance AND the
    instance that still owns its planes under the running launch's map are final; everything else is marked per (instance, stage)
    and done by the fix-up - nothing twice, nothing never."""
    B = 9
    ocp, wl = util.make(name, N, K, B, seed=4)
    desc = _capi.desc_from_ocp(ocp, batch=B)
    Bp = (B + 3) // 4 * 4
    rng = np.random.default_rng(0)
    emu.usv_emu_lin_modes.argtypes = [C.POINTER(_capi.Desc)] + [_capi._dp] * 4 + [_capi._ip] * 3 + [_capi._dp] * 2 + [_capi._ip]
    words = (N + 32) // 32
    for trial in range(4):
        ready = (rng.uniform(size=B) < (0.0, 0.5, 0.8, 1.0)[trial]).astype(np.int32)
        pn, pc = rng.permutation(B).astype(np.int32), rng.permutation(B).astype(np.int32)
        wsa, wsb = np.zeros((N + 1, Bp, 64, 16)), np.zeros((N + 1, Bp, 64, 16))   # (64 >= planes per stage of any layout)
        redo = np.zeros((B, words), dtype=np.int32)
        npt = emu.usv_emu_lin_modes(C.byref(desc), _d(wl["x_init"]), _d(wl["u_init"]), _d(wl["yref"]), _d(wl["yref_e"]), _i(ready), _i(pn), _i(pc),
                                    _d(wsa), _d(wsb), _i(redo))
        assert npt > 0
        a = wsa.reshape(-1)[: (N + 1) * Bp * npt * 16]
        b = wsb.reshape(-1)[: (N + 1) * Bp * npt * 16]
        assert np.array_equal(a, b), trial
        assert np.abs(a).max() > 0
        # the mask: stage k of instance pn[g] was left to the fix-up iff that instance or the planes' current owner pc[g] was not final
        # (padded groups replay the last instance: they can only add bits for an instance that is marked anyway or whose owner is late)
        for g in range(B):
            inst, owner = pn[g], pc[g]
            late = not (ready[inst] and ready[owner])
            bits = [(redo[inst, k // 32] >> (k % 32)) & 1 for k in range(N + 1)]
            if late:
                assert all(bits), (trial, g)
        if trial == 3:
            assert not redo.any()


def test_conditional_predictor_corrector_kernel_bodies_match_oracle(oracle, emu):
    """Option "cond_pred_corr" (HPIPM's conditional predictor-corrector: DESIGN.md section 2, qp_ipm.hpp QpIpm::solve, usv_opts.cond_pred_corr):
    the kernel bodies on the lane emulator against the oracle with the same option, on a closed loop of the hard-row bench workload where the
    fallback does fire (it moves 5 % of the instances by up to 1e-2: another path into the tolerance ball of a QP with control weight R = 0) -
    statuses equal, iteration counts equal on all but a handful, iterates as close as without the option; a factor nothing can exceed
    leaves the plain iteration's bits."""
    name, N, K, B = "usv_model_pf_ca", 20, 4, 40
    wl = scenario.make_bench_batch(name, N, K, B, seed=1234)
    dt, steps = scenario.BENCH_DT, scenario.BENCH_SIM_STEPS[name]
    ocp = usv_models.make_ocp(name, N * dt, N, K)
    ocp.solver_options.sim_method_num_steps = steps
    desc = _capi.desc_from_ocp(ocp, batch=B)
    # (both sides on the default profile but for the option; the oracle without its iterative refinement, which the kernels do not have)
    plain = oracle.spec(util.MODEL_ID[name], N, N * dt, K, sim_steps=steps, cond_pred_corr=0, itref_corr_max=0)
    cpc = oracle.spec(util.MODEL_ID[name], N, N * dt, K, sim_steps=steps, cond_pred_corr=1, itref_corr_max=0)
    emu.usv_emu_set_cpc.argtypes = [C.c_int, C.c_double]
    x, u = wl["x_init"].copy(), wl["u_init"].copy()
    w = dict(wl)
    rng = np.random.default_rng(5)
    moved = agree = total = 0
    try:
        for t in range(3):
            emu.usv_emu_set_cpc(0, 2.0)
            e0 = emu_rti(emu, desc, w, x, u)
            emu.usv_emu_set_cpc(1, 1e30)
            e_never = emu_rti(emu, desc, w, x, u)
            for f in ("x", "u", "pi", "qp_iter", "status"):
                assert np.array_equal(e0[f], e_never[f]), (t, f)     # the CPC code path with so = 1 everywhere: the same bits
            emu.usv_emu_set_cpc(1, 2.0)
            e1 = emu_rti(emu, desc, w, x, u)
            xo, uo = x.copy(), u.copy()
            sto, ito = oracle.rti_batch(cpc, xo, uo, w["x0"], w["yref"], w["yref_e"], w["p"], w["lh"], threads=0)
            xp, up = x.copy(), u.copy()
            stp, itp = oracle.rti_batch(plain, xp, up, w["x0"], w["yref"], w["yref_e"], w["p"], w["lh"], threads=0)
            assert np.array_equal(e1["status"], sto)
            ok = (sto == 0) & (e1["qp_status"] == 0) & (ito < 50)
            same = ok & (e1["qp_iter"] == ito)
            agree += int(same.sum()); total += int(ok.sum())
            e = np.maximum(util.rel_err_per_instance(e1["x"][same], xo[same]), util.rel_err_per_instance(e1["u"][same], uo[same]))
            assert np.median(e) <= 1e-9 and e.max() <= 5e-3, (t, e.max())
            fired = ok & (stp == 0) & (np.maximum(util.rel_err_per_instance(xo, xp), util.rel_err_per_instance(uo, up)) > 1e-9)
            moved += int(fired.sum())
            # where the option changes the oracle's answer it changes the kernels' too
            ek = np.maximum(util.rel_err_per_instance(e1["x"], e0["x"]), util.rel_err_per_instance(e1["u"], e0["u"]))
            assert ((ek > 1e-9) == fired)[ok & (stp == 0) & (e0["qp_status"] == 0)].mean() >= 0.9
            x, u = e0["x"], e0["u"]
            x0 = x[:, 1].copy()
            x0[:, 3] += 1e-3 * rng.standard_normal(B); x0[:, 5] += 1e-3 * rng.standard_normal(B)
            w["x0"] = x0
    finally:
        emu.usv_emu_set_cpc(-1, 2.0)   # (back to what the descriptor says)
    assert moved >= 2, moved                       # the fallback did fire
    assert agree >= 0.95 * total, (agree, total)   # same iteration counts as the oracle with the option
