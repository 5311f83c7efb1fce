"""The synthetic workload generators (mpc_collisionavoidance_amd/scenario.py) against SURVEY.md 8(d)."""
import numpy as np

from mpc_collisionavoidance_amd import scenario


def test_survey_generator_pf_ca_distributions_and_clip():
    N, K, B = 40, 10, 2000
    wl = scenario.make_bench_batch("usv_model_pf_ca", N, K, B)
    x0 = wl["x0"]
    assert wl["dt"] == 0.05 and wl["sim_steps"] == 5 and wl["generator"] == "survey"
    assert 2.0 <= x0[:, 10].min() and x0[:, 10].max() <= 6.0 and -5.0 <= x0[:, 11].min() and x0[:, 11].max() <= 15.0
    assert 0.3 <= x0[:, 3].min() and x0[:, 3].max() <= 1.2
    assert 0.09 < np.abs(x0[:, 4]).max() <= 0.1                      # sway v ~ U(-0.1, 0.1) as SURVEY
    assert np.ptp(wl["p"], axis=1).max() == 0.0                       # static set
    # every vehicle starts outside every keep-out circle by at least SURVEY's 0.5 m, and the course ray does not enter a
    # circle before s_min = 0.4 + 1.2 u
    pos = x0[:, 10:12]
    obs = wl["p"][:, 0].reshape(B, K, 2)
    lh = wl["lh"][:, 0]
    d = np.sqrt(((obs - pos[:, None, :]) ** 2).sum(-1))
    assert (d - lh).min() >= 0.5 - 1e-9
    course = x0[:, 0] + np.arctan2(x0[:, 4], x0[:, 3] + 0.001)
    rel = obs - pos[:, None, :]
    lon = rel[..., 0] * np.cos(course)[:, None] + rel[..., 1] * np.sin(course)[:, None]
    lat = -rel[..., 0] * np.sin(course)[:, None] + rel[..., 1] * np.cos(course)[:, None]
    hit = np.abs(lat) < lh
    s_enter = np.where(hit, lon - np.sqrt(np.maximum(lh ** 2 - lat ** 2, 0)), np.inf)
    smin = (0.4 + 1.2 * x0[:, 3])[:, None]
    assert (s_enter >= smin - 1e-6).all()
    # the initial guess is a trajectory of the model: stage 0 is x0, the parameter-like states never move
    assert np.array_equal(wl["x_init"][:, 0], x0)
    assert np.ptp(wl["x_init"][:, :, 7:10], axis=1).max() == 0.0 and np.ptp(wl["x_init"][:, :, 12:14], axis=1).max() == 0.0


def test_survey_rollout_is_the_oracles_integrator(oracle):
    N, K, B = 12, 3, 6
    wl = scenario.make_bench_batch("usv_model_pf_ca", N, K, B, seed=5)
    for b in range(B):
        x = wl["x0"][b].copy()
        for k in range(N):
            x, _, _ = oracle.erk_sens(2, wl["dt"], wl["sim_steps"], x, np.zeros(2))
            assert np.allclose(x, wl["x_init"][b, k + 1], rtol=0, atol=1e-11)


def test_legacy_generator_unchanged_defaults():
    a = scenario.make_batch("usv_model_pf_ca", 10, 4, 16)
    assert a["generator"] == "beside" and a["dt"] == 0.01 and np.abs(a["x0"][:, 4]).max() <= 0.03
    # at the 2 s look-ahead the field is SURVEY's own (6 m); other horizons scale it (3 Tf)
    b = scenario.make_bench_batch("usv_model_guidance_ca1", 40, 4, 16)
    c = scenario.make_batch("usv_model_guidance_ca1", 40, 4, 16, dt=0.05)
    assert all(np.array_equal(b[k], c[k]) for k in ("x0", "p", "lh", "x_init"))
    d = scenario.make_bench_batch("usv_model_guidance_ca1", 20, 4, 16)
    e = scenario.make_batch("usv_model_guidance_ca1", 20, 4, 16, dt=0.05, max_range=3.0)
    assert all(np.array_equal(d[k], e[k]) for k in ("x0", "p", "lh", "x_init"))


def test_survey_verbatim_is_the_survey_generator_without_the_departures():
    """bench.py --workload survey-verbatim: the same seed-1234 vehicles and obstacle draws as the default workload,
    NO obstacle clip, acados' own initial guess x_k = x0, the disturbance on every state."""
    import bench
    name, N, K, B = "usv_model_pf_ca", 40, 10, 4000
    a = scenario.make_bench_batch(name, N, K, B)
    b = scenario.make_bench_batch(name, N, K, B, verbatim=True)
    assert np.array_equal(a["x0"], b["x0"]) and np.array_equal(a["lh"], b["lh"]) and np.array_equal(a["yref"], b["yref"])
    moved = (a["p"][:, 0] != b["p"][:, 0]).reshape(B, K, 2).any(axis=2)
    assert 0.08 < moved.mean() < 0.14                      # "11 % of the obstacles" are what the clip moves
    # an unclipped obstacle sits at least as near as its clipped version
    pos = a["x0"][:, 10:12]
    da = np.linalg.norm(a["p"][:, 0].reshape(B, K, 2) - pos[:, None, :], axis=2)
    db = np.linalg.norm(b["p"][:, 0].reshape(B, K, 2) - pos[:, None, :], axis=2)
    assert (db <= da + 1e-12).all() and (db[moved] < da[moved]).all()
    assert np.array_equal(b["x_init"], np.repeat(b["x0"][:, None, :], N + 1, axis=1)) and not b["u_init"].any()
    assert b["generator"] == "survey_verbatim"
    wl, Bn, dt, steps, sigma, mask = bench.make_workload(name, N, K, 64, 0, "survey-verbatim", False, 0, 1)
    assert mask == scenario.ALL_STATES_MASK == (1 << 14) - 1 and steps == 5 and dt == 0.05 and sigma == 1e-3 and Bn == 64
    assert wl["generator"] == "survey_verbatim"
    # the soft-row model has no clip and no roll-out to drop: only the initial guess changes
    c, d = scenario.make_bench_batch("usv_model_guidance_ca1", 20, 3, 50), scenario.make_bench_batch("usv_model_guidance_ca1", 20, 3, 50, verbatim=True)
    assert np.array_equal(c["p"], d["p"]) and np.array_equal(d["x_init"], np.repeat(d["x0"][:, None, :], 21, axis=1))
