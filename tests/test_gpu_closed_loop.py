"""GPU closed-loop parity (run on a real MI355X with `pytest -m gpu`).

* the bench workload itself (BASELINE configs[2] as SURVEY.md 8(d) spells it out: usv_model_pf_ca, N=40, Tf=2 s,
  10 static obstacles, closed loop x0 <- x1 + disturbance), B distinct instances, 25 ticks of solve + advance with
  the bench's options (static obstacle set in registers, difficulty binning on), against the CPU oracle TICK BY TICK.
  Two oracle runs follow the device:
    - "same inputs": before every tick the oracle is handed the iterate and the x0 the device starts that tick from, so
      each tick compares ONE application of the solver on identical inputs - status of every instance, the failing set,
      IPM iteration counts, and the iterate with the per-component relative error of tests/util.rel_err.
      usv_model_guidance_ca1: every instance <= 1e-7 (measured ~1e-10).  usv_model_pf_ca, on every tick: median <= 1e-9, at least 90 % of the
      instances <= 1e-7 in states and controls (measured over 25 ticks of 512 instances: median ~1e-12, 97th percentile
      between 1e-9 and 1.4e-7), every instance <= 1e-5 (north_star) except isolated ones (at most 0.4 % per tick; measured 1e-4 of
      the solves) which must then carry an independent certificate - the device's point satisfies the KKT conditions of its QP
      as evaluated by tests/kkt.py - and stay within 5e-3.
      The outliers (up to ~3e-4 on a thrust rate) are this model's conditioning,
      not slack in the kernels: its control weight is R = 0 (scripts/usv_pf_ca/acados_settings.py:93-99), the
      thrust-rate profile is fixed only through the barrier terms, and the QP solution itself is known no better than
      3e-2 (controls) / 9e-4 (states) at the default IPM tolerances - that is how far BOTH implementations sit from the
      oracle converged to 1e-11 (profiles/r02_parity_probe.txt, profiles/r03_parity_tail.txt: tools/parity_tail.py).  The two implementations
      follow the same iteration path and so agree ~7 orders of magnitude better than that, except where round-off
      moves an iterate across one of the path's kinks;
    - "free running": the oracle keeps its own iterate and only receives the device's x0.  The closed-loop map of the
      hard-row, bang-bang model amplifies the per-tick differences (measured: up to 6e-3 on a thrust rate around tick 5,
      decaying again), so this leg checks what must survive - the same status / failing set up to threshold cases - and
      bounds the divergence at 5e-2;
* the reference's own usv_pf_ca scenario at its exact settings (N=100, Tf=1, 4 obstacles,
  /root/reference/catkin_ws/src/nmpc_ca/scripts/usv_pf_ca/main.py:54-55,73-75,106-133) through the AcadosOcpSolver
  look-alike, 25 closed-loop ticks;
* a batch solved as two handles on the shards sharding.shard_bounds gives two ranks returns bit-for-bit what one
  handle returns for the whole batch (what bench.py --gpus 2 relies on).
"""
import numpy as np
import pytest

from mpc_collisionavoidance_amd import BatchOcpSolver, scenario, sharding, usv_models
from tests import parity_rule, util

pytestmark = pytest.mark.gpu
TOL = 1e-7


def _closed_loop(oracle, name, N, K, B, ticks, sigma, tol_max, seed=1234, wide=None):
    wl = scenario.make_bench_batch(name, N, K, B, seed=seed)
    dt, steps = scenario.BENCH_DT, scenario.BENCH_SIM_STEPS[name]
    ocp = usv_models.make_ocp(name, N * dt, N, K)
    ocp.solver_options.sim_method_num_steps = steps
    s = BatchOcpSolver(ocp, B)
    scenario.load_into(s, wl)
    s.set_option("static_obstacles", 1)
    s.set_option("disturbance_mask", scenario.NOISE_MASK[name])
    if wide is not None:   # which mapping (usvmpc_last_mapping): 0 = four instances per wavefront - the bench's kernel -, 1 = one
        s.set_option("wide", wide)
    spec = oracle.spec(util.MODEL_ID[name], N, N * dt, K, sim_steps=steps)
    data = (wl["yref"], wl["yref_e"], wl["p"], wl["lh"])
    xf, uf = wl["x_init"].copy(), wl["u_init"].copy()      # free-running oracle iterate
    xg, ug, x0 = wl["x_init"].copy(), wl["u_init"].copy(), wl["x0"].copy()
    good_f = np.ones(B, dtype=bool)     # free run: converged on both sides in every tick so far
    slack = max(1, int(0.01 * B))       # instances allowed to sit on a threshold (iteration cap / step-length floor)
    out = dict(fail_g=0, fail_o=0, worst_x=0.0, worst_u=0.0, p90=0.0, p50=0.0, worst_free=0.0, above=0)
    for t in range(ticks):
        xs, us = xg.copy(), ug.copy()   # same-inputs oracle: the device's iterate before this tick
        xin, uin = xg.copy(), ug.copy()
        s.solve_async()
        s.sync()
        sts, its = oracle.rti_batch(spec, xs, us, x0, *data, threads=8)
        stf, itf = oracle.rti_batch(spec, xf, uf, x0, *data, threads=8)
        stg, qs, qi = s.get_int("status"), s.get_int("qp_status"), s.get_int("qp_iter")
        assert wide is None or (s.last_mapping() > 0) == (wide > 0)   # (1: one wave per instance, 4: four - small batches)
        assert int(s.fail_counts(1)[0]) == int((stg != 0).sum())          # the on-device audit counter
        xg, ug = s.get_all("x"), s.get_all("u")
        # ---- same inputs: status of every instance, failing set, iteration counts, iterate
        assert (stg != sts).sum() <= slack, (name, t, np.where(stg != sts)[0])
        conv_g, conv_s = qs == 0, (sts == 0) & (its < spec.opts.qp_iter_max)
        assert (conv_g != conv_s).sum() <= slack, (name, t)
        ok = conv_g & conv_s
        assert ok.mean() >= 0.97, (name, t, ok.mean())
        ex, eu = util.rel_err_per_instance(xg[ok], xs[ok]), util.rel_err_per_instance(ug[ok], us[ok])
        out["worst_x"], out["worst_u"] = max(out["worst_x"], ex.max()), max(out["worst_u"], eu.max())
        out["p90"] = max(out["p90"], np.percentile(ex, 90), np.percentile(eu, 90))
        out["p50"] = max(out["p50"], np.percentile(ex, 50), np.percentile(eu, 50))
        assert np.percentile(ex, 90) <= TOL and np.percentile(eu, 90) <= TOL, (name, t, np.percentile(ex, 90), np.percentile(eu, 90))
        assert np.percentile(ex, 50) <= 1e-9 and np.percentile(eu, 50) <= 1e-9, (name, t)
        dit = np.abs(qi - its)[ok]
        # north_star's 1e-5 on every instance - or, for the isolated instance above it (measured: 2 of 20 442 solves of this
        # workload, 3.8e-4 and 1.3e-5, profiles/r03_parity_tail.txt: QPs the oracle itself cannot converge to 1e-11, unchanged by an
        # IEEE-division build of the kernels), an independent certificate: the device's point must satisfy the KKT conditions of
        # its QP (tests/kkt.py) and stay within 5e-3 of the oracle's
        e = np.maximum(ex, eu)
        if tol_max < 1e-5:
            # (soft-row model) instances that took the oracle's number of iterations: EVERY one within tol_max (1e-7) - the bound of rounds
            # 1 - 4 again (round 5 had widened it to 1e-5 for the maximum after one instance at 1.7e-6 under the R04 profile;
            # under the default profile the worst over 512 x 25 + 256 x 25 solves is 1.5e-8).  An instance whose exit test is passed by a
            # hair's breadth on one side only stops an iteration apart: at most `slack` of them (asserted below), each within
            # SOFT_DIT_CAP = 1e-3 of the oracle AND carrying the KKT certificate when above 1e-5
            same = dit == 0
            if same.any():
                assert e[same].max() <= tol_max, (name, t, np.percentile(e[same], 99), e[same].max())
            if (~same).any():
                assert e[~same].max() <= parity_rule.SOFT_DIT_CAP, (name, t, e[~same].max())
                okd = ok.copy()
                okd[np.where(ok)[0][same]] = False
                r = parity_rule.check(oracle, spec, s, okd, e[~same], xin, uin, x0, data, soft=name == "usv_model_guidance_ca1",
                                      cap=parity_rule.SOFT_DIT_CAP)
                out["above"] += r["above"]
                assert not r["violations"], (name, t, r)
        else:   # tests/parity_rule.py
            r = parity_rule.check(oracle, spec, s, ok, e, xin, uin, x0, data, soft=name == "usv_model_guidance_ca1")
            out["above"] += r["above"]
            assert not r["violations"], (name, t, r)
        # the same iteration count on all but a handful of instances (an instance whose path round-off moves across a kink
        # can take a very different number of iterations to the same tolerance: seen once in 512 x 25, 20 vs 45)
        assert (dit > 0).sum() <= slack and (dit > 1).sum() <= 2, (name, t, dit.max(), (dit > 0).sum())
        # a failed solve leaves its iterate untouched on both sides (acados: status 4 returns before the update)
        bad = (stg != 0) & (sts != 0)
        assert np.array_equal(xg[bad], xs[bad]) and np.array_equal(ug[bad], us[bad])
        out["fail_g"] += int((stg != 0).sum())
        out["fail_o"] += int((sts != 0).sum())
        # ---- free running: statuses and bounded divergence
        differ = (stg != stf) & good_f
        assert differ.sum() <= slack, (name, t, np.where(differ)[0])
        good_f &= conv_g & (stf == 0) & (itf < spec.opts.qp_iter_max)
        assert good_f.mean() >= 0.9, (name, t, good_f.mean())
        ef = max(util.rel_err(xg[good_f], xf[good_f]), util.rel_err(ug[good_f], uf[good_f]))
        out["worst_free"] = max(out["worst_free"], ef)
        assert ef <= 5e-2, (name, t, ef)
        # ---- hand-over: x1 of the device's iterate plus the disturbance on the masked states only
        s.advance(sigma, seed=2000 + t)
        s.sync()
        x0 = s.get("x0", 0)
        d = x0 - xg[:, 1]
        mask = np.array([(scenario.NOISE_MASK[name] >> j) & 1 for j in range(d.shape[1])], dtype=bool)
        assert np.all(d[:, ~mask] == 0.0)
        if sigma > 0:
            assert 0.5 * sigma < d[:, mask].std() < 2.0 * sigma
    s.close()
    print(name, out)
    return out


def test_bench_workload_closed_loop_pf_ca(oracle):
    # (the bench's kernel: four instances per wavefront - a handle of 512 would take the latency mapping by default)
    r = _closed_loop(oracle, "usv_model_pf_ca", 40, 10, 512, ticks=25, sigma=1e-3, tol_max=1e-5, wide=0)
    # failures do not pile up: hard rows make some QPs infeasible for a tick or two, but the count stays small
    assert r["fail_g"] <= 0.01 * 512 * 25 and abs(r["fail_g"] - r["fail_o"]) <= 25, r


def test_bench_workload_closed_loop_guidance_ca1(oracle):
    r = _closed_loop(oracle, "usv_model_guidance_ca1", 40, 10, 512, ticks=25, sigma=1e-3, tol_max=TOL, wide=0)
    assert r["fail_g"] == r["fail_o"] == 0, r


def test_closed_loop_on_the_latency_mapping_pf_ca(oracle):
    """The same closed loop on the WIDE mapping (one instance per wavefront; K = 10: the two-pass form), the same parity rule
    (tests/parity_rule.py) - on 1024 instances, so that the rule's 0.4 % of a tick is four instances, not two."""
    r = _closed_loop(oracle, "usv_model_pf_ca", 40, 10, 1024, ticks=12, sigma=1e-3, tol_max=1e-5, wide=1)
    assert r["fail_g"] <= 0.01 * 1024 * 12 and abs(r["fail_g"] - r["fail_o"]) <= 25, r


def test_closed_loop_on_the_latency_mapping_guidance_ca1(oracle):
    r = _closed_loop(oracle, "usv_model_guidance_ca1", 40, 10, 256, ticks=25, sigma=1e-3, tol_max=TOL, wide=1)
    assert r["fail_g"] == r["fail_o"] == 0, r


def test_reference_scenario_pf_ca_exact_settings(oracle):
    """scripts/usv_pf_ca/main.py at its own settings: N=100, Tf=1, obstacles (3,2),(4,8),(3.7,16),(4.2,20) of radius
    0.5 (+0.2), start at the origin heading 0 towards the path (4,-5)->(4,25); same call sequence, 25 ticks."""
    N, Tf, K = 100, 1.0, 4
    constraint, model, acados_solver = usv_models.acados_settings(Tf, N, name="usv_model_pf_ca", n_obstacles=K)
    spec = oracle.spec(2, N, Tf, K)
    x1, y1, x2, y2 = 4.0, -5.0, 4.0, 25.0
    ak = np.arctan2(y2 - y1, x2 - x1)
    ye = -(0.0 - x1) * np.sin(ak) + (0.0 - y1) * np.cos(ak)
    x_start = np.array([0.0, 0.0, 1.0, 0.001, 0, 0, ye, x1, y1, ak, 0, 0, 0, 0])
    acados_solver.set(0, "lbx", x_start)
    acados_solver.set(0, "ubx", x_start)
    pobs = np.array([3, 2, 4, 8, 3.7, 16, 4.2, 20], dtype=float)
    robs = np.full(K, 0.7)
    yref = np.array([0, np.sin(ak), np.cos(ak), 0.7, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0])
    x0m = np.asarray(model.x0, dtype=float)
    xo, uo = np.tile(x0m, (N + 1, 1)), np.zeros((N, 2))   # acados_create: x = constraints.x0 on every stage, u = 0
    x0o = x_start.copy()
    for i in range(25):
        for j in range(N):
            acados_solver.set(j, "yref", yref)
            acados_solver.set(j, "p", pobs)
            acados_solver.constraints_set(j, "lh", robs)
        acados_solver.set(N, "yref", yref[:14])
        acados_solver.set(N, "p", pobs)
        status = acados_solver.solve()
        r = oracle.rti(spec, xo, uo, x0o, np.tile(yref, (N, 1)), yref[:14], np.tile(pobs, (N + 1, 1)), np.tile(robs, (N, 1)))
        xo, uo = r["x"], r["u"]
        assert status == r["status"] == 0, (i, status, r["status"])
        xs = np.stack([acados_solver.get(j, "x") for j in range(N + 1)])
        us = np.stack([acados_solver.get(j, "u") for j in range(N)])
        assert util.rel_err(xs, xo) <= TOL and util.rel_err(us, uo) <= TOL, (i, util.rel_err(xs, xo), util.rel_err(us, uo))
        x0o = xo[1].copy()
        x0g = acados_solver.get(1, "x")
        acados_solver.set(0, "lbx", x0g)
        acados_solver.set(0, "ubx", x0g)


@pytest.mark.parametrize("name,B", [("usv_model_pf_ca", 203), ("usv_model_guidance_ca1", 64)])
def test_two_handles_on_shards_equal_unsharded(name, B):
    N, K = 20, 6
    wl = scenario.make_bench_batch(name, N, K, B, seed=77)
    dt, steps = scenario.BENCH_DT, scenario.BENCH_SIM_STEPS[name]
    ocp = usv_models.make_ocp(name, N * dt, N, K)
    ocp.solver_options.sim_method_num_steps = steps

    def run(w, n):
        s = BatchOcpSolver(ocp, n)
        scenario.load_into(s, w)
        s.set_option("disturbance_mask", scenario.NOISE_MASK[name])
        for t in range(3):
            s.solve()
            s.advance(0.0)
        out = (s.get_all("x"), s.get_all("u"), s.get_int("status"), s.get_int("qp_iter"), s.get("x0", 0))
        s.close()
        return out

    whole = run(wl, B)
    parts = []
    for r in range(2):
        lo, hi = sharding.shard_bounds(B, 2, r)
        parts.append(run(sharding.split_workload(wl, 2, r), hi - lo))
    for i in range(5):
        assert np.array_equal(np.concatenate([parts[0][i], parts[1][i]], axis=0), whole[i]), i


@pytest.mark.parametrize("name,N,K,G,world", [("usv_model_pf_ca", 20, 3, 4096, 8), ("usv_model_guidance_ca1", 20, 3, 4096, 8),
                                               ("usv_model_pf_ca", 40, 10, 3000, 6)])
def test_shards_that_land_on_the_other_mapping_equal_the_unsharded_batch(name, N, K, G, world):
    """The mapping is chosen from the size of the handle's batch (usvmpc.hip launch_qp: the latency mapping while the batch leaves SIMDs idle):
    4096 instances in one handle run four per wavefront, the same instances as 8 shards of 512 run one per wavefront (or, soft-row OCP,
    one per workgroup of four).  An instance's result must not depend on which: bit for bit, over a closed loop (VERDICT r04
    weak 2 - the mappings used to agree to rounding only, amplified to 5e-3 by the hard-row model)."""
    wl = scenario.make_bench_batch(name, N, K, G, seed=99)
    dt, steps = scenario.BENCH_DT, scenario.BENCH_SIM_STEPS[name]
    ocp = usv_models.make_ocp(name, N * dt, N, K)
    ocp.solver_options.sim_method_num_steps = steps

    def run(w, n):
        s = BatchOcpSolver(ocp, n)
        scenario.load_into(s, w)
        s.set_option("static_obstacles", 1)
        s.set_option("disturbance_mask", scenario.NOISE_MASK[name])
        maps = []
        for t in range(3):
            s.solve()
            maps.append(s.last_mapping())
            s.advance(0.0)
        s.sync()
        out = (s.get_all("x"), s.get_all("u"), s.get_int("status"), s.get_int("qp_iter"), s.get("x0", 0), s.get_all("pi"), s.get_all("lam"), s.get_all("t"))
        s.close()
        return out, maps

    whole, maps_whole = run(wl, G)
    parts, maps_parts = [], []
    for r in range(world):
        lo, hi = sharding.shard_bounds(G, world, r)
        o, m = run(sharding.split_workload(wl, world, r), hi - lo)
        parts.append(o)
        maps_parts += m
    assert set(maps_whole) == {0} and 0 not in set(maps_parts), (maps_whole, maps_parts)   # the test is about crossing the threshold
    for i in range(len(whole)):   # (equal_nan: the multipliers of an instance whose IPM ended in NaN - status 4, iterate untouched - are NaN on both sides)
        assert np.array_equal(np.concatenate([q[i] for q in parts], axis=0), whole[i], equal_nan=i >= 5), i


def test_bench_scale_shards_equal_the_unsharded_batch():
    """bench.py --global-batch at BASELINE configs[2] scale on ONE GPU: the seed-1234 batch of 65 536 instances solved as two handles on
    the slices two ranks would get (bench.make_workload: shard b -> rank floor(b * 2 / B)) returns, bit for bit, what one handle returns for
    the whole batch - what configs[3] / [4] (one batch sharded over 8 GPUs) rely on."""
    import bench
    name, N, K, G = "usv_model_pf_ca", 40, 10, 65536

    def run(rank, world):
        wl, B, dt, steps, sigma, mask = bench.make_workload(name, N, K, 0, G, "survey", False, rank, world)
        ocp = usv_models.make_ocp(name, N * dt, N, K)
        ocp.solver_options.sim_method_num_steps = steps
        s = BatchOcpSolver(ocp, B)
        scenario.load_into(s, wl)
        s.set_option("static_obstacles", 1)
        s.set_option("disturbance_mask", mask)
        for t in range(2):
            s.solve_async()
            s.advance(0.0)      # (the disturbance stream is indexed by the instance's number inside its handle: no noise here)
        s.sync()
        out = (s.get_all("x"), s.get_all("u"), s.get_int("status"), s.get_int("qp_iter"), s.get("x0", 0))
        s.close()
        return out

    whole = run(0, 1)
    parts = [run(r, 2) for r in range(2)]
    for i in range(5):
        assert np.array_equal(np.concatenate([parts[0][i], parts[1][i]], axis=0), whole[i]), i
    assert (whole[2] == 0).mean() > 0.99
