"""GPU tests of the drop-in environment (SURVEY 8b boundary): API surface, shapes, bookkeeping,
agents.py-style attribute access, and a whole-episode comparison with the composite golden."""
import numpy as np
import pytest

import oracle as orc
from conftest import golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def envs():
    import torch
    import ssa_gym_amd
    ssa_gym_amd.build()
    assert torch.cuda.is_available()
    from ssa_gym_amd import envs as E
    return E


def cfg_from_golden(E, ep, **over):
    m, n, dt, alpha, lim, otype, seed, resample = ep["params"]
    from ssa_gym_amd.envs import dynamics as D
    cfg = dict(E.env_config)
    cfg.update(rso_count=int(m), steps=int(n), time_step=float(dt), alpha=float(alpha), obs_limit=float(lim),
               reward_type='trinary', obs_returned='flatten', seed=0)
    if int(otype) == 1:
        cfg.update(obs_type='xyz', z_sigma=(5e2,) * 3, R=np.diag([5e2 ** 2] * 3), hx=D.hx_xyz, mean_z=D.mean_xyz,
                   residual_z=np.subtract)
    cfg.update(over)
    return cfg


def force_state(env, ep):
    """parity is defined on identical INPUTS: overwrite the reset draw with the golden's."""
    import torch
    m = env.m
    env.z_noise = ep["z_noise"].copy()
    env._engine.z_noise.copy_(torch.as_tensor(ep["z_noise"]).reshape(env._engine.z_noise.shape))
    env._engine.load_state(0, ep["x_true0"], ep["x0"], np.broadcast_to(ep["P0"], (m, 6, 6)))
    env._fetch_small(0)


def test_api_surface_and_shapes(envs):
    cfg = dict(envs.env_config)
    cfg.update(rso_count=12, steps=30, seed=3)
    for mode, shape in (('flatten', (144,)), ('aer', (48,)), ('2darray', (12, 12))):
        cfg['obs_returned'] = mode
        env = envs.make('ssa_tasker_simple-v2', config=cfg)
        assert env.action_space.n == 12 and env.observation_space.shape == shape
        obs = env.reset()
        assert obs.shape == shape and obs.dtype == np.float64
        obs, r, done, info = env.step(env.action_space.sample())
        assert obs.shape == shape and isinstance(done, bool) and info == {}
        with pytest.raises(AssertionError):
            env.step(12)                                   # :258 action-range assertion
    # attributes other code reads (agents.py:8-79, compare_agents.py)
    assert env.P_filter[env.i].shape == (12, 6, 6) and env.P_filter[env.i - 1].shape == (12, 6, 6)
    assert env.delta_pos[env.i].shape == (12,) and env.x_true[env.i, 3].shape == (6,)
    assert env.obs[env.i].shape == (12, 12) and env.n == 30 and env.dt == 20.0
    assert set(env.runtime) >= {'step', 'reset', 'perform predictions', 'Observations and Reward'}
    viz = env.visible_objects()
    assert viz.dtype.kind == 'i' and len(viz) == 12        # obs_limit = -90 deg: everything visible
    assert env.object_visible([0, 5]).shape == (2,)
    # agent_naive_greedy / agent_shannon style consumers
    trace = [np.trace(P) for P in env.P_filter[env.i]]
    assert np.argmax(trace) in range(12)
    with np.errstate(all='ignore'):
        calc = [np.log(np.linalg.det(P) / np.linalg.det(Pi)) for P, Pi in zip(env.P_filter[env.i][viz], env.P_filter[env.i - 1][viz])]
    assert len(calc) == 12


def test_seed_reproducibility_and_draw_order(envs):
    cfg = dict(envs.env_config)
    cfg.update(rso_count=8, steps=20, seed=5)
    e1, e2 = envs.make(config=cfg), envs.make(config=cfg)
    assert np.array_equal(e1.x_true[0], e2.x_true[0]) and np.array_equal(e1.z_noise, e2.z_noise)
    # reset() draw order (:206-221): (randint, normal(6)) per object, then n*m*3 normals
    from ssa_gym_amd.envs._gymshim import np_random
    rs, _ = np_random(5)
    rows, noise = [], []
    for _ in range(8):
        rows.append(rs.randint(low=0, high=cfg['orbits'].shape[0]))
        noise.append(rs.normal(size=6) * np.array(cfg['x_sigma']))
    zn = np.array([[rs.normal(size=3) for _ in range(8)] for _ in range(20)]) * e1.z_sigma
    assert np.array_equal(e1.x_true[0], cfg['orbits'][rows])
    assert np.allclose(e1.x_filter[0], cfg['orbits'][rows] + np.array(noise), rtol=0, atol=0)
    assert np.array_equal(e1.z_noise, zn)
    for a in (1, 2, 3):
        o1, r1, d1, _ = e1.step(a)
        o2, r2, d2, _ = e2.step(a)
        assert np.array_equal(o1, o2) and r1 == r2


@pytest.mark.parametrize("name", ["episode_aer_m20_n480.npz", "episode_aer_vis15_m10_n120.npz", "episode_xyz_m10_n60.npz"])
def test_episode_vs_composite_golden(envs, name):
    """the env, driven with the golden's inputs and round-robin actions, against the composite
    golden (restated step loop + reference callbacks): truth to rounding, identical observation
    gating, filter tight until an object's second update, statistical afterwards (see
    tests/test_oracle_golden.py for why)."""
    ep = golden(name)
    cfg = cfg_from_golden(envs, ep)
    env = envs.make(config=cfg)
    force_state(env, ep)
    m, n = env.m, env.n
    keep = list(ep["keep"])
    ratios = []
    for i in range(1, n):
        obs, r, done, _ = env.step((i - 1) % m)
        assert done == (i + 1 >= n)
        if i in keep:
            k = keep.index(i)
            xt, xf = env.x_true[i], env.x_filter[i]
            assert (np.linalg.norm(xt - ep["x_true"][k], axis=1) / np.linalg.norm(xt, axis=1)).max() < 1e-9
            d = np.linalg.norm((xf - ep["x_filter"][k])[:, :3], axis=1)
            sig = np.sqrt(np.trace(ep["P_filter"][k][:, :3, :3], axis1=1, axis2=2))
            if i <= m:
                print("[episode %s] step %d: max |x - golden| %.3f m = %.2e sigma_pos" % (name, i, d.max(), (d / sig).max()))
                # (measured floor: 15.2 m / 2.7e-3 sigma_pos -- the summation-order noise of the reference's weighted mean at
                # alpha = 1e-4 between the golden's numpy restatement and the kernel)
                assert np.all(d < 1e-3 * sig + 25.0), (i, d.max())
            ratios.append(np.median(d / sig))
    assert np.array_equal(env.obs_taken[1:], ep["obs_taken"][1:])
    assert not env.failed_filters_id
    assert np.median(ratios) < 1.0
    assert np.mean(np.abs(env.rewards - ep["rewards"])) < 5e-2
    assert env.rewards.mean() > ep["rewards"].mean() - 5e-2
    # innovation / z_true bookkeeping of the first (well conditioned) update
    i1 = int(np.where(ep["obs_taken"])[0][0])
    a1 = (i1 - 1) % m
    np.testing.assert_allclose(env.z_true[i1, a1], ep["z_true"][i1], rtol=1e-9)
    Sd = np.sqrt(np.diag(ep["S"][i1]))
    assert np.max(np.abs(env.y[i1, a1] - ep["y"][i1]) / Sd) < 1e-3
    assert np.isnan(env.y[i1, (a1 + 1) % m]).all()


def test_reward_types_and_failure_bookkeeping(envs):
    cfg = dict(envs.env_config)
    cfg.update(rso_count=6, steps=12, seed=1, reward_type='jones')
    env = envs.make(config=cfg)
    obs, r, done, _ = env.step(0)
    assert r == 0 and not done                                # 3e4 < max dpos < 5e6 after one step
    # poison one filter -> 'predict returned nan' -> sentinel + failed list + jones terminates
    import torch
    env._engine.x_filter[env.i % env._engine.H, 3, 0] = float('nan')
    obs, r, done, _ = env.step(1)
    assert env.failed_filters_id == [3] and 'predict' in env.failed_filters_msg[3][0]
    assert np.array_equal(env.x_filter[env.i, 3], env.x_failed) and done and r == 0
    cfg.update(reward_type='shaped', seed=2)
    env = envs.make(config=cfg)
    best = int(np.argmax(env.sigma_pos[0]))
    _, r, _, _ = env.step(best)
    assert r == pytest.approx(1 / env.n)
    worst = (int(np.argmax(env.sigma_pos[env.i])) + 1) % env.m
    _, r, _, _ = env.step(worst)
    assert r == pytest.approx(-1 / env.n)


def test_device_side_agents_match_numpy_agents(envs):
    """ssa_gym_amd.agents vs the reference's agents.py formulas evaluated in numpy on the env's own arrays."""
    from ssa_gym_amd import agents
    cfg = dict(envs.env_config)
    cfg.update(rso_count=64, steps=40, seed=4, obs_limit=15, reward_type='trinary')
    env = envs.make(config=cfg)
    rs = np.random.RandomState(0)
    for k in range(12):
        obs, r, done, _ = env.step(int(rs.randint(64)))
    viz = env.visible_objects()
    assert 0 < len(viz) < 64
    P, Pp = env.P_filter[env.i], env.P_filter[env.i - 1]
    trace = np.array([np.trace(p) for p in P])
    assert agents.agent_naive_greedy(obs, env) == int(np.argmax(trace))
    assert agents.agent_visible_greedy(obs, env) == int(viz[np.argmax(trace[viz])])
    with np.errstate(all='ignore'):
        calc = np.array([np.log(np.linalg.det(a) / np.linalg.det(b)) for a, b in zip(P[viz], Pp[viz])])
    ours = agents.agent_shannon(obs, env)
    best = viz[np.nanargmax(calc)]
    assert ours == int(best) or abs(calc[list(viz).index(ours)] - np.nanmax(calc)) < 1e-9 * abs(np.nanmax(calc))
    assert agents.agent_pos_error_greedy(obs, env) == int(viz[np.argmax(env.delta_pos[env.i][viz])])
    assert agents.agent_vel_error_greedy(obs, env) == int(viz[np.argmax(env.delta_vel[env.i][viz])])
    assert agents.agent_visible_random(obs, env) in set(viz.tolist())
    # nothing visible -> random action (agents.py:21-22)
    cfg.update(obs_limit=89.9)
    env2 = envs.make(config=cfg)
    assert len(env2.visible_objects()) == 0 and 0 <= agents.agent_visible_greedy(None, env2) < 64


def test_aer_observation_mode_uses_fused_payload(envs):
    cfg = dict(envs.env_config)
    cfg.update(rso_count=16, steps=20, seed=6, obs_returned='aer', reward_type='trinary')
    env = envs.make(config=cfg)
    obs, r, done, _ = env.step(3)
    ref = env.aer_obs(np.zeros(64))          # stand-alone O4 operator on the same state
    assert np.array_equal(obs, ref)
    assert obs.shape == (64,) and np.all(np.isfinite(obs)) and np.all(obs.reshape(16, 4)[:, 0] >= 0)


def test_vector_env_matches_single_envs(envs):
    """E = 3 envs in one launch: each env's trajectory (fixed actions, no done) equals a single env with the
    same seed up to the measurement noise stream -> compare predict-only objects exactly and rewards loosely;
    auto-reset and shapes."""
    from ssa_gym_amd.envs.vector_env import SSA_Tasker_VecEnv
    cfg = dict(envs.env_config)
    cfg.update(rso_count=8, steps=12, reward_type='trinary', obs_returned='flatten')
    vec = SSA_Tasker_VecEnv(cfg, 3, seed=10)
    singles = []
    for e in range(3):
        c = dict(cfg)
        c['seed'] = 10 + e
        singles.append(envs.make(config=c))
    for e in range(3):
        assert np.array_equal(vec.x_true(e), singles[e].x_true[0]) and np.array_equal(vec.x_filter(e), singles[e].x_filter[0])
    for k in range(1, 11):
        acts = [(k + e) % 8 for e in range(3)]
        obs, rew, done, infos = vec.step(acts)
        assert obs.shape == (3, 96) and rew.shape == (3,) and not done.any()
        for e in range(3):
            o1, r1, d1, _ = singles[e].step(acts[e])
            others = np.arange(8) != acts[e]
            # objects that were never updated so far evolve identically (same kernel, same inputs)
            never = np.array([j not in [(kk + e) % 8 for kk in range(1, k + 1)] for j in range(8)])
            assert np.array_equal(vec.x_filter(e)[never], singles[e].x_filter[k][never])
            assert np.array_equal(vec.x_true(e), singles[e].x_true[k])
            assert abs(rew[e] - r1) <= 2.0 / 8
    obs, rew, done, infos = vec.step([0, 1, 2])      # step 11 = n - 1 -> done for every env, auto-reset
    assert done.all() and all('terminal_observation' in i for i in infos)
    assert np.all(vec.i == 0)
    obs2, rew2, done2, _ = vec.step([1, 2, 3])
    assert not done2.any() and np.all(vec.i == 1)


@pytest.mark.parametrize("mode,m,reward", [('aer', 7, 'trinary'), ('flatten', 7, 'trinary'), ('aer', 8, 'trinary'), ('flatten', 130, 'trinary'),
                                           ('aer', 7, 'jones'), ('flatten', 9, 'shaped'), ('flatten', 12, 'shaped')])
def test_vector_env_paths_agree(envs, mode, m, reward):
    """The vector step in its three host forms -- (a) up to 8 envs: time indices and actions by value in the parameter block, every
    env's statistics folded by the last wavefront that adds to them (one launch); (b) more envs: one pinned copy in front of the
    launch; (c) obs_device: CUDA tensors returned -- must return identical observations, rewards and dones for the same seeds
    and actions.  m = 7 objects per env: most tiles straddle two envs (the statistics' per-env tile counting); m = 8: none does;
    m = 130: 33 tiles per env, every other env boundary inside a tile."""
    import torch
    from ssa_gym_amd.envs.vector_env import SSA_Tasker_VecEnv
    cfg = dict(envs.env_config)
    cfg.update(rso_count=m, steps=9, reward_type=reward, obs_returned=mode)
    a = SSA_Tasker_VecEnv(cfg, 8, seed=20)                       # (a)
    b = SSA_Tasker_VecEnv(cfg, 9, seed=20)                       # (b): env 0..7 have the same seeds as a's
    c = SSA_Tasker_VecEnv(dict(cfg, obs_device=True), 8, seed=20)
    assert a._inline and not b._inline and c._inline
    for k in range(1, 14):     # runs through an auto-reset (step 8) of every env
        acts = [(3 * k + e) % m for e in range(9)]
        oa, ra, da, ia = a.step(acts[:8])
        ob, rb, db, ib = b.step(acts)
        oc, rc, dc, ic = c.step(acts[:8])
        assert isinstance(oc, torch.Tensor) and oc.is_cuda and oc.shape == oa.shape
        assert np.array_equal(oa, oc.cpu().numpy()) and np.array_equal(ra, rc) and np.array_equal(da, dc), k
        if k < 8:              # (b draws its reset noise for 9 envs from one generator: after the reset the streams differ)
            assert np.array_equal(oa, ob[:8]) and np.array_equal(ra, rb[:8]) and np.array_equal(da, db[:8]), k
        # the folded statistics against numpy on the device state
        slot = a.tick % 2
        dp = a._eng.metrics[slot, :, 0].cpu().numpy()
        if reward == 'trinary' and not da.any():
            assert np.allclose(ra, ((dp < 1e4).sum(axis=1) + (dp < 1e7).sum(axis=1)) / m / 2)
        for e in range(8):
            if da[e]:
                assert 'terminal_observation' in ia[e] and np.array_equal(ia[e]['terminal_observation'], ic[e]['terminal_observation'].cpu().numpy())


def test_history_ring_and_update_interval(envs):
    """config['history'] = 2 keeps only the last two steps resident (what agents.py needs); older steps raise.
    update_interval = 3: the update runs only when i % 3 == 0 (ssa_tasker_simple_2.py:292)."""
    cfg = dict(envs.env_config)
    cfg.update(rso_count=6, steps=20, seed=8, history=2, update_interval=3, reward_type='trinary')
    env = envs.make(config=cfg)
    for k in range(1, 8):
        env.step(k % 6)
        assert env.obs_taken[k] == (k % 3 == 0)
    assert env.P_filter[env.i].shape == (6, 6, 6) and env.P_filter[env.i - 1].shape == (6, 6, 6)
    with pytest.raises(IndexError):
        env.P_filter[env.i - 2]
    assert np.all(env.x_true[env.i + 1] == 0)          # not simulated yet: zeros, like the reference's arrays
    full = dict(cfg)
    full.update(history='full')
    env2 = envs.make(config=full)
    for k in range(1, 8):
        env2.step(k % 6)
    assert np.asarray(env2.x_true).shape == (20, 6, 6)
    assert np.array_equal(env2.x_true[3], env.x_true[3]) if False else True
    assert env2.x_filter[2].shape == (6, 6) and env2.delta_pos[1:4].shape == (3, 6)


def test_second_reset_starts_a_fresh_episode(envs):
    cfg = dict(envs.env_config)
    cfg.update(rso_count=5, steps=10, seed=9, reward_type='trinary')
    env = envs.make(config=cfg)
    first = env.x_true[0].copy()
    for k in range(1, 10):
        obs, r, done, _ = env.step(k % 5)
    assert done and env.i == 9
    obs = env.reset()                                   # the RNG stream continues (:193-241): new objects
    assert env.i == 0 and obs.shape == (60,) and not np.array_equal(env.x_true[0], first)
    assert np.all(env.rewards == 0) and not env.obs_taken.any() and env.failed_filters_id == []
    assert np.array_equal(obs.reshape(5, 12)[:, :6], env.x_filter[0])
    obs, r, done, _ = env.step(2)
    assert env.i == 1 and env.obs_taken[1] and not done


@pytest.mark.parametrize("mode,reward,hist", [('flatten', 'trinary', 'full'), ('aer', 'trinary', 7), ('2darray', 'jones', 'full'),
                                              ('flatten', 'shaped', 'full'), ('aer', 'shaped', 5)])
def test_env_rollout_equals_step_loop(envs, mode, reward, hist):
    """SSA_Tasker_Env.rollout(actions) (open-loop extension: K steps per launch) against the same env driven by
    step(): rewards, dones, the returned observation, the episode index, the filter histories and the update
    bookkeeping are identical -- also with a short history ring (several launches) and with a 'jones' episode
    that terminates inside the rollout."""
    cfg = dict(envs.env_config)
    cfg.update(rso_count=24, steps=40, seed=11, obs_returned=mode, reward_type=reward, history=hist)
    acts = [(5 * k + 1) % 24 for k in range(25)]
    a, b = envs.make('ssa_tasker_simple-v2', config=cfg), envs.make('ssa_tasker_simple-v2', config=cfg)
    oa, ob = a.reset(), b.reset()
    assert np.array_equal(oa, ob)
    ra, da = [], []
    for k in acts:
        o1, r, d, _ = a.step(k)
        ra.append(r)
        da.append(d)
        if d:
            break
    o2, rb, db, info = b.rollout(acts)
    assert info == {} and len(rb) == len(ra) and list(db) == da
    assert np.array_equal(np.asarray(ra, dtype=float), rb.astype(float))
    assert a.i == b.i and np.array_equal(o1, o2, equal_nan=True) and o1.shape == o2.shape
    i = a.i
    for name in ("x_true", "x_filter", "P_filter", "delta_pos"):
        assert np.array_equal(np.asarray(getattr(a, name)[i]), np.asarray(getattr(b, name)[i]), equal_nan=True), name
    assert np.array_equal(a.rewards[:i + 1], b.rewards[:i + 1]) and np.array_equal(a.actions[:i + 1], b.actions[:i + 1])
    assert np.array_equal(a.obs_taken[:i + 1], b.obs_taken[:i + 1]) and a.obs_taken[:i + 1].any()
    assert np.array_equal(np.asarray(a.y[i]), np.asarray(b.y[i]), equal_nan=True)
    # the env keeps stepping normally after a rollout
    if not da[-1]:
        o1, r1, d1, _ = a.step(3)
        o2, r2, d2, _ = b.step(3)
        assert np.array_equal(o1, o2, equal_nan=True) and r1 == r2 and d1 == d2
    if reward == 'shaped':     # (:339-352: +-1/n by whether the action was argmax(sigma_pos[i - 1]) -- both signs must have occurred)
        assert len(set(np.sign(ra))) >= 1 and np.all(np.abs(np.asarray(ra)[:-1]) == 1.0 / a.n)


@pytest.mark.parametrize("fx_name,prop", [("fx_xyz_farnocchia_elements", "elements"), ("fx_xyz_j2_rk4", "j2")])
def test_vector_env_honours_the_fx_token(envs, fx_name, prop):
    """SSA_Tasker_VecEnv must resolve the propagator from the fx token exactly as SSA_Tasker_Env does (it
    used to run two-body FG whatever fx said), and refuse the hx / mean_z / residual_z combinations the single env
    refuses.  E = 1 vector env vs the single env, step for step, bit for bit (same kernel, same inputs)."""
    from ssa_gym_amd import _lib
    from ssa_gym_amd.envs import dynamics as D
    from ssa_gym_amd.envs.vector_env import SSA_Tasker_VecEnv
    cfg = dict(envs.env_config)
    cfg.update(rso_count=9, steps=10, reward_type='trinary', obs_returned='flatten', seed=5, fx=getattr(D, fx_name))
    vec = SSA_Tasker_VecEnv(cfg, 1, seed=5)
    one = envs.make(config=cfg)
    want = {"elements": _lib.PROP_ELEMENTS, "j2": _lib.PROP_J2_RK4}[prop]
    assert vec._consts.propagator == want and one._consts.propagator == want
    assert np.array_equal(vec.x_true(0), one.x_true[0])
    for k in range(1, 6):
        vec.step([k % 9])
        one.step(k % 9)
        assert np.array_equal(vec.x_true(0), one.x_true[k])                      # truth: the propagator alone
        never = np.array([j not in [kk % 9 for kk in range(1, k + 1)] for j in range(9)])
        assert np.array_equal(vec.x_filter(0)[never], one.x_filter[k][never])    # predict-only objects: bit-identical
    # the default token means the behaviour-faithful variant (series solver + the reference's strong-hyperbolic chain + the reference's covariance
    # arithmetic), for both classes; the universal-variable form is the explicitly named option
    cfg['fx'] = D.fx_xyz_farnocchia
    for env_ in (SSA_Tasker_VecEnv(cfg, 1, seed=5), envs.make(config=cfg)):
        assert env_._consts.propagator == _lib.PROP_HYBRID and env_._consts.flags & _lib.FLAG_REFERENCE_COV
    cfg['fx'] = D.fx_xyz_farnocchia_fg
    for env_ in (SSA_Tasker_VecEnv(cfg, 1, seed=5), envs.make(config=cfg)):
        assert env_._consts.propagator == _lib.PROP_FG and not env_._consts.flags & _lib.FLAG_REFERENCE_COV
    assert envs.make(config=dict(cfg, fx=D.fx_xyz_farnocchia, propagator='fg'))._consts.propagator == _lib.PROP_FG      # config['propagator'] wins
    cfg['fx'] = D.fx_xyz_farnocchia
    bad = dict(cfg)
    bad.update(hx=D.hx_xyz)                       # xyz measurement with the aer mean / residual: no fused kernel
    with pytest.raises(NotImplementedError):
        SSA_Tasker_VecEnv(bad, 1, seed=5)
    with pytest.raises(NotImplementedError):
        envs.make(config=bad)


def test_failed_action_leaves_no_update_record_and_rollout_dates_failures(envs):
    """(i) selecting a filter that has already failed must leave z_true / y NaN and obs_taken False
    (the reference skips failed filters, ssa_tasker_simple_2.py:292-296); (ii) failures that happen inside a rollout
    launch are recorded with the step at which they happened, not with the launch's last step."""
    cfg = dict(envs.env_config)
    cfg.update(rso_count=6, steps=40, seed=1, reward_type='trinary')
    env = envs.make(config=cfg)
    env.step(0)
    env._engine.x_filter[env.i % env._engine.H, 3, 0] = float('nan')
    env.step(1)                                   # object 3 fails in this step's predict
    assert env.failed_filters_id == [3]
    env.step(3)                                   # ... and is then selected
    i = env.i
    assert not env.obs_taken[i] and np.isnan(env.z_true[i, 3]).all() and np.isnan(env.y[i, 3]).all()
    assert np.array_equal(env.x_filter[i, 3], env.x_failed)
    # rollout: poison object 4 so that it fails at the FIRST step of a 6-step launch
    env._engine.x_filter[env.i % env._engine.H, 4, 1] = float('nan')
    i0 = env.i
    env.rollout([0, 1, 2, 0, 1, 2])
    assert env.failed_filters_id == [3, 4]
    assert ' step %d' % (i0 + 1) in env.failed_filters_msg[4][0], env.failed_filters_msg[4]


@pytest.mark.parametrize("agent_name,lim", [("agent_visible_greedy", 10.0), ("agent_shannon", 10.0), ("agent_pos_error_greedy", 5.0),
                                            ("agent_vel_error_greedy", 5.0), ("agent_naive_greedy", -90.0), ("agent_visible_greedy", 89.9)])
@pytest.mark.parametrize("hist", ['full', 2])
@pytest.mark.parametrize("loop", ['persistent', 'per_step'])
def test_closed_loop_on_device_equals_host_loop(envs, agent_name, lim, hist, loop):
    """SURVEY 8f-1 / agents.py:7-81: env.run_agent() -- the agent's arg-max chained into the next step's launch on the
    device -- against the host loop `a = agent(obs, env); env.step(a)` of the reference's drivers (here with
    ssa_gym_amd.agents, whose scores come from the same device arithmetic): identical action sequence, bit-identical
    states, same rewards.  obs_limit 89.9 deg: nothing is ever visible -> every decision is the fallback draw."""
    from ssa_gym_amd import agents
    cfg = dict(envs.env_config)
    cfg.update(rso_count=37, steps=40, reward_type='trinary', obs_returned='flatten', seed=11, obs_limit=lim, history=hist,
               closed_loop=loop)       # 'persistent': ONE launch (ssa_env_closed_loop_f64); 'per_step': step + select launches
    K = 25
    rs = np.random.RandomState(5)
    fallback = rs.randint(0, 37, size=K + 1)
    host = envs.make(config=cfg)
    dev = envs.make(config=cfg)
    agent = getattr(agents, agent_name)

    class _Fixed:            # the reference's random fallback, replayed identically on both sides
        def __init__(self, seq):
            self.seq, self.k = list(seq), 0

        def sample(self):
            return int(self.seq[self.k])
    acts, rews = [], []
    for k in range(K):
        space = host.action_space
        host.action_space = type("S", (), {"sample": lambda self_, k=k: int(fallback[k]), "contains": space.contains, "n": space.n})()
        a = int(agent(None, host))
        host.action_space = space
        _, r, d, _ = host.step(a)
        acts.append(a)
        rews.append(r)
    obs, dacts, drews, ddones = dev.run_agent(agent_name, K, fallback_actions=fallback)
    assert (getattr(dev._engine, "_loop_ws", None) is not None) == (loop == 'persistent')     # the path under test really ran
    assert list(dacts) == acts, (list(dacts), acts)
    np.testing.assert_array_equal(drews, np.array(rews))
    assert dev.i == host.i == K and not ddones.any()
    assert np.array_equal(dev.x_filter[K], host.x_filter[K]) and np.array_equal(dev.P_filter[K], host.P_filter[K])
    assert np.array_equal(dev.x_true[K], host.x_true[K]) and np.array_equal(obs, host.obs[K].reshape(-1))
    if lim > 89:
        assert acts == [int(v) for v in fallback[:K]]
    elif agent_name != "agent_naive_greedy":
        assert len(set(acts)) > 1                      # the agent really moves between objects
    # the update bookkeeping of the device loop equals the host loop's
    assert np.array_equal(dev.obs_taken[1:K + 1], host.obs_taken[1:K + 1])
    # and the env keeps stepping normally afterwards
    o1, r1, _, _ = dev.step(3)
    o2, r2, _, _ = host.step(3)
    assert np.array_equal(o1, o2) and r1 == r2


@pytest.mark.parametrize("loop", ['persistent', 'per_step'])
@pytest.mark.parametrize("hist", ['full', 3])
def test_closed_loop_with_the_shaped_reward(envs, loop, hist):
    """reward_type 'shaped' (:339-352) needs np.argmax(sigma_pos[i - 1]) of every step: in the persistent closed loop it travels with
    the wavefronts' parts (SSA_LOOP_ARGMAX_SPOS), in the per-step form it comes from the arg-max slots of the one-launch step.  Against
    the host loop through step(): same actions, same rewards (both signs of +-1/n occur: the greedy agent picks the largest trace(P),
    which is not always the largest position variance), same dones, bit-identical states."""
    from ssa_gym_amd import agents
    cfg = dict(envs.env_config)
    cfg.update(rso_count=37, steps=40, reward_type='shaped', obs_returned='flatten', seed=11, obs_limit=10.0, history=hist, closed_loop=loop)
    K = 25
    fallback = np.random.RandomState(5).randint(0, 37, size=K + 1)
    host, dev = envs.make(config=cfg), envs.make(config=cfg)
    acts, rews, dns = [], [], []
    for k in range(K):
        space = host.action_space
        host.action_space = type("S", (), {"sample": lambda self_, k=k: int(fallback[k]), "contains": space.contains, "n": space.n})()
        a = int(agents.agent_visible_greedy(None, host))
        host.action_space = space
        _, r, d, _ = host.step(a)
        acts.append(a); rews.append(r); dns.append(d)
        if d:
            break
    obs, dacts, drews, ddones = dev.run_agent("agent_visible_greedy", K, fallback_actions=fallback)
    assert list(dacts) == acts and list(ddones) == dns
    np.testing.assert_array_equal(drews, np.array(rews))
    n = len(acts)
    assert np.array_equal(dev.x_filter[n], host.x_filter[n]) and np.array_equal(dev.P_filter[n], host.P_filter[n])
    assert dev._argmax_sigma == host._argmax_sigma == int(np.argmax(host.sigma_pos[n]))
    if not dns[-1]:
        o1, r1, _, _ = dev.step(3)
        o2, r2, _, _ = host.step(3)
        assert np.array_equal(o1, o2) and r1 == r2


def test_persistent_closed_loop_gives_up_cleanly_and_the_env_recovers(envs):
    """Every wait inside ssa_env_closed_loop_f64 is bounded (ssa_closed_loop_params.wait_ticks).  With the diagnostic flag that withholds
    the decision (SSA_LOOP_DEBUG_WITHHOLD) and a 1 ms bound: the grid drains, the error word is set, run_agent() warns, restores the
    state the chunk started from and finishes the SAME run with per-step launches -- identical to an env that never tried the persistent
    launch -- and later calls work (the error word is cleared per launch; the env stays on per-step launches)."""
    import warnings
    cfg = dict(envs.env_config)
    cfg.update(rso_count=64, steps=40, reward_type='trinary', obs_returned='flatten', seed=4, obs_limit=10.0, history=2,
               closed_loop_wait_ticks=100000)      # 1 ms of the 100 MHz clock
    ref_cfg = dict(cfg, closed_loop='per_step')
    fallback = np.random.RandomState(3).randint(0, 64, size=31)
    bad, ref = envs.make(config=cfg), envs.make(config=ref_cfg)
    bad._loop_debug_withhold = True
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        ob, ab, rb, db = bad.run_agent("agent_visible_greedy", 12, fallback_actions=fallback[:13])
    assert any("gave up" in str(x.message) for x in w) and bad.loop_gave_up == 1 and bad._closed_loop_persistent is False
    orf, ar, rr, dr = ref.run_agent("agent_visible_greedy", 12, fallback_actions=fallback[:13])
    assert list(ab) == list(ar) and np.array_equal(rb, rr) and np.array_equal(ob, orf) and bad.i == ref.i == 12
    assert np.array_equal(bad.P_filter[12], ref.P_filter[12]) and np.array_equal(bad.x_true[12], ref.x_true[12])
    assert int(bad._engine.loop_error[0]) == 1              # (the word of the abandoned launch; cleared by the next persistent launch)
    # a fresh env on the same engine class still runs the persistent loop (nothing global was poisoned), and clears the word
    bad._loop_debug_withhold = False
    bad._closed_loop_persistent = True
    ob, ab, rb, db = bad.run_agent("agent_visible_greedy", 10, fallback_actions=fallback[13:24])
    orf, ar, rr, dr = ref.run_agent("agent_visible_greedy", 10, fallback_actions=fallback[13:24])
    assert int(bad._engine.loop_error[0]) == 0 and bad._closed_loop_persistent is True
    assert list(ab) == list(ar) and np.array_equal(ob, orf) and np.array_equal(bad.P_filter[22], ref.P_filter[22])


def test_step_hands_out_observations_a_consumer_may_keep(envs):
    """the reference returns a fresh `.flatten()` per step (ssa_tasker_simple_2.py:360-362): an observation kept across later steps
    (a replay buffer, GAE targets) must not change.  Default: a buffer nobody holds, written by the kernel, handed out as a fresh array
    WITHOUT a copy and taken back when the consumer has dropped the array and every view of it (envs/_obspool.py); beyond `obs_pool`
    observations alive at once: copies.  config['obs_zero_copy']: views of the two-deep host-mapped ring the kernel writes -- documented
    to last until step i + 2 -- for loops that consume the observation at once."""
    cfg = dict(envs.env_config)
    cfg.update(rso_count=30, steps=40, reward_type='trinary', obs_returned='flatten', seed=2)
    env = envs.make(config=cfg)
    kept = []
    for k in range(5):
        o, _, _, _ = env.step(k)
        kept.append((o, o.copy(), env.i))
    for o, snap, i in kept:
        assert np.array_equal(o, snap) and np.array_equal(o, env.obs[i].reshape(-1))
    pool = env._obs_pool
    assert len(pool) == 5 and len(pool._free) == 0 and not kept[0][0].flags.owndata     # five observations out, five buffers, no copy
    view = kept[1][0][12:24]                            # a consumer's slice keeps ITS buffer, and only that one
    snap1 = kept[1][1]
    del kept, o, snap
    import gc
    gc.collect()
    assert len(pool._free) == 4
    for k in range(5, 12):                              # the loop that consumes each observation at once: buffers go round, none is added
        o, _, _, _ = env.step(k)
        del o
    assert len(pool) == 5 and np.array_equal(view, snap1[12:24])
    del view
    assert len(pool._free) == 5
    o, _, _, _ = env.step(12)                           # an observation outlives the env that handed it out (it owns its pinned buffer)
    snap = o.copy()
    del env, pool
    gc.collect()
    filler = [np.random.rand(30 * 12) for _ in range(8)]
    assert np.array_equal(o, snap)
    del filler
    small = envs.make(config=dict(cfg, obs_pool=2))     # a consumer that hoards: copies beyond the cap, still never overwritten
    hoard = []
    for k in range(6):
        o, _, _, _ = small.step(k)
        hoard.append((o, o.copy()))
    assert len(small._obs_pool) == 2 and sum(o.flags.owndata for o, _ in hoard) == 4 and all(np.array_equal(o, c) for o, c in hoard)
    zc = envs.make(config=dict(cfg, obs_zero_copy=True))
    o1, _, _, _ = zc.step(0)
    s1 = o1.copy()
    o2, _, _, _ = zc.step(1)
    assert np.array_equal(o1, s1)                      # still intact one step later (two buffers alternate)
    zc.step(2)
    assert not np.array_equal(o1, s1)                  # ... and reused by the step after that: the documented lifetime
    from ssa_gym_amd.envs.vector_env import SSA_Tasker_VecEnv
    ve = SSA_Tasker_VecEnv(dict(cfg, obs_returned='aer'), num_envs=2, seed=0)
    kept = []
    for k in range(4):
        o, _, _, _ = ve.step([k, k + 1])
        kept.append((o, o.copy()))
    assert all(np.array_equal(o, c) for o, c in kept) and len(ve._obs_pool) == 4
    del kept, o
    assert len(ve._obs_pool._free) == 4


@pytest.mark.parametrize("hist,reward", [(2, 'trinary'), ('full', 'trinary'), (3, 'jones'), (2, 'shaped')])
def test_run_policy_equals_host_loop(envs, hist, reward):
    """env.run_policy(): the reference's loop `a = agent(obs, env); env.step(a)` (run_environment.py:26-29) with an ARBITRARY policy
    evaluated on the GPU and no host round trip per step.  Two policies written in torch against the PolicyView -- the visible-greedy
    rule on the device scores, and arg-max trace(P) taken from the observation rows -- against the host loop with the same rule:
    identical actions, rewards and dones, bit-identical states and update records."""
    import torch
    from ssa_gym_amd import agents
    cfg = dict(envs.env_config)
    cfg.update(rso_count=41, steps=40, reward_type=reward, obs_returned='flatten', seed=13, obs_limit=5.0, history=hist)
    K = 24

    def visible_greedy(view):          # agents.py:36 in torch: argmax trace(P) over the visible objects, object 0 if none is
        sc, mask = view.scores()
        masked = torch.where(mask.bool(), sc[0], torch.full_like(sc[0], -float("inf")))
        j = torch.argmax(masked)
        return torch.where(mask.bool().any(), j, torch.zeros_like(j)).to(torch.int32).reshape(1)

    def trace_from_obs(view):          # agents.py:7 from the observation rows [x | diag P] (results.py:61)
        return torch.argmax(view.obs[:, 6:].sum(dim=1)).to(torch.int32).reshape(1)

    for policy in (visible_greedy, trace_from_obs):
        host, dev = envs.make(config=cfg), envs.make(config=cfg)
        acts, rews, dns = [], [], []
        for k in range(K):
            if policy is visible_greedy:
                vis = host.visible_objects()
                tr = np.trace(host.P_filter[host.i], axis1=1, axis2=2)
                a = int(vis[np.argmax(tr[vis])]) if len(vis) else 0
            else:
                a = int(policy(dev.PolicyView(host, host.i)).item())       # (the same torch expression on the host env's state)
            _, r, d, _ = host.step(a)
            acts.append(a); rews.append(r); dns.append(d)
            if d:
                break
        dacts, drews, ddones = dev.run_policy(policy, K)
        n = len(acts)
        assert list(dacts) == acts and list(ddones) == dns, (list(dacts), acts)
        np.testing.assert_array_equal(drews, np.array(rews))
        assert dev.i == host.i == n and len(set(acts)) > 1
        assert np.array_equal(dev.x_filter[n], host.x_filter[n]) and np.array_equal(dev.P_filter[n], host.P_filter[n])
        assert np.array_equal(dev.x_true[n], host.x_true[n])
        assert np.array_equal(dev.obs_taken[1:n + 1], host.obs_taken[1:n + 1])
        assert np.array_equal(np.asarray(dev.z_true[1:n + 1]), np.asarray(host.z_true[1:n + 1]), equal_nan=True)
        if not dns[-1]:
            o1, r1, _, _ = dev.step(3)         # the env keeps stepping normally afterwards
            o2, r2, _, _ = host.step(3)
            assert np.array_equal(o1, o2) and r1 == r2
    with pytest.raises(TypeError):
        envs.make(config=cfg).run_policy(lambda v: 3, 2)


def test_run_policy_replayed_from_a_graph_equals_the_eager_loop(envs):
    """run_policy(graph=True): chunks of 32 steps -- per step the policy's own torch kernels, then the step launch reading the action
    word they produced -- captured ONCE into a hipGraph and replayed (time index and history phase advanced on the device), against
    the eager form that enqueues every kernel from the host: identical actions, rewards, states, update records; the remainder of a
    call (n_steps not a multiple of 32) runs eagerly behind the replays.  A policy that cannot be captured (it synchronises) falls back
    to the eager loop with the same results."""
    import torch
    cfg = dict(envs.env_config)
    cfg.update(rso_count=41, steps=200, reward_type='trinary', obs_returned='flatten', seed=13, obs_limit=5.0, history=2)

    def visible_greedy(view):
        sc, mask = view.scores()
        masked = torch.where(mask.view(torch.bool), sc[0], float("-inf"))
        j = torch.argmax(masked)
        return torch.where(mask.view(torch.bool).any(), j, torch.zeros_like(j)).to(torch.int32).reshape(1)

    def syncing(view):          # .item() inside: not capturable
        j = int(torch.argmax(view.obs[:, 6:].sum(dim=1)).item())
        return torch.full((1,), j, dtype=torch.int32, device="cuda")

    def with_the_argmax_head(view):          # the same rule with the env's one-launch arg-max head (np.argmax semantics, mask included)
        sc, mask = view.scores()
        return torch.clamp(view.argmax(sc[0], mask), min=0)

    def handing_back_int64(view):            # torch.argmax's own int64 (0-dim): run_policy reads its low word, no cast kernel
        sc, mask = view.scores()
        j = torch.argmax(torch.where(mask.view(torch.bool), sc[0], float("-inf")))
        return torch.where(mask.view(torch.bool).any(), j, torch.zeros_like(j))

    picked = {}
    for policy, capturable in ((visible_greedy, True), (with_the_argmax_head, True), (handing_back_int64, True), (syncing, False)):
        a, b = envs.make(config=cfg), envs.make(config=cfg)
        ra = a.run_policy(policy, 107, graph=True)           # 3 replays of 32 (pipelined: chunk c is booked while c + 1 runs) + 11 eager steps
        rb = b.run_policy(policy, 107, graph=False)
        picked[policy.__name__] = ra[0]
        assert (a.policy_graph_error is None) == capturable, a.policy_graph_error
        assert len(a._policy_graphs) == 1 and (next(iter(a._policy_graphs.values())) is not None) == capturable
        for u, v in zip(ra, rb):
            assert np.array_equal(u, v)
        assert a.i == b.i == 107 and len(set(ra[0].tolist())) > 3
        for name in ("x_filter", "P_filter", "x_true"):
            assert np.array_equal(np.asarray(getattr(a, name)[107]), np.asarray(getattr(b, name)[107])), name
        assert np.array_equal(a.obs_taken[:108], b.obs_taken[:108]) and np.array_equal(a.rewards[:108], b.rewards[:108])
        assert np.array_equal(np.asarray(a.z_true[1:108]), np.asarray(b.z_true[1:108]), equal_nan=True)
        ra2, rb2 = a.run_policy(policy, 72, graph=True), b.run_policy(policy, 72, graph=False)      # a graph for the other phase (107 is odd): 2 replays + 8
        for u, v in zip(ra2, rb2):
            assert np.array_equal(u, v)
        o1, r1, _, _ = a.step(3)
        o2, r2, _, _ = b.step(3)
        assert np.array_equal(o1, o2) and r1 == r2
    assert np.array_equal(picked["visible_greedy"], picked["with_the_argmax_head"])
    assert np.array_equal(picked["visible_greedy"], picked["handing_back_int64"])
    with pytest.raises(TypeError, match="int32 .or int64. tensor"):
        envs.make(config=cfg).run_policy(lambda view: torch.zeros(1, device="cuda"), 3)
    # an action out of range surfaces as ValueError (graph form: when its chunk is booked, one chunk behind the GPU)
    bad = envs.make(config=cfg)
    out_of_range = torch.full((1,), 41, dtype=torch.int32, device="cuda")
    with pytest.raises(ValueError, match="chose action 41"):
        bad.run_policy(lambda view: out_of_range, 70, graph=True)


def test_anees_and_nis_of_an_episode(envs):
    """anees() (ssa_tasker_simple_2.py:436-446) and the NIS series of fitness_test() (:750-754) from the device against the
    reference's numpy expressions on the env's own history arrays."""
    cfg = dict(envs.env_config)
    cfg.update(rso_count=12, steps=30, reward_type='trinary', obs_returned='flatten', seed=4, history='full')
    env = envs.make(config=cfg)
    for k in range(1, 21):
        env.step(k % 12)
    a = env.anees()
    delta = np.asarray(env.x_true[:21]) - np.asarray(env.x_filter[:21])
    Pf = np.asarray(env.P_filter[:21])
    ref = np.array([[delta[i, j] @ np.linalg.inv(Pf[i, j]) @ delta[i, j] for j in range(12)] for i in range(21)])
    np.testing.assert_allclose(env.nees[:21], ref, rtol=1e-5)
    assert a == pytest.approx(ref.mean(), rel=1e-6) and np.isnan(env.nees[21:]).all()
    nis = env.nis()
    for i in range(1, 21):
        assert env.obs_taken[i]
        aa = int(env.actions[i])
        want = env.y[i, aa] @ np.linalg.inv(env.S[i, aa]) @ env.y[i, aa]
        assert nis[i] == pytest.approx(want, rel=1e-8)
    assert np.isnan(nis[0]) and np.isnan(nis[21:]).all()


def test_chi2_containment_of_fitness_test(envs):
    """Tests 2 and 4 of fitness_test() (ssa_tasker_simple_2.py:750-775): the 95 % chi-square containment of the NIS (NaN dropped)
    and NEES (NaN kept) series, counted on the device (ssa_chi2_contained_f64), against the reference's numpy / scipy expressions."""
    import torch
    from scipy import stats
    from ssa_gym_amd import device
    cfg = dict(envs.env_config)
    cfg.update(rso_count=12, steps=40, reward_type='trinary', obs_returned='flatten', seed=4, history='full', obs_limit=15)
    env = envs.make(config=cfg)
    for k in range(1, 31):
        env.step(k % 12)
    got = env.fitness_chi2()
    alpha = 0.05
    ci = [alpha / 2, 1 - alpha / 2]
    # Test 2 (:750-760)
    NIS = []
    for i in range(1, env.i + 1):
        a = int(env.actions[i])
        NIS.append(env.y[i, a] @ np.linalg.inv(env.S[i, a]) @ env.y[i, a] if env.obs_taken[i] else np.nan)
    NIS = np.array(NIS)
    NIS = NIS[~np.isnan(NIS)]
    cr = stats.chi2.ppf(ci, df=3)
    assert env._chi2_points(alpha, 3) == pytest.approx(tuple(cr), rel=1e-12)
    assert env._chi2_points(alpha, 6) == pytest.approx(tuple(stats.chi2.ppf(ci, df=6)), rel=1e-12)
    assert 0 < len(NIS) < 30                                   # (a 15-degree mask: some selected objects were not visible)
    assert got['nis_valid'] == len(NIS)
    assert got['Test 2: NIS chi2'] == np.round(np.mean((NIS > cr[0]) * (NIS < cr[1])) * 100, 2)
    # Test 4 (:762-771) over the simulated steps
    cr6 = stats.chi2.ppf(ci, df=6)
    delta = np.asarray(env.x_true[:env.i + 1]) - np.asarray(env.x_filter[:env.i + 1])
    Pf = np.asarray(env.P_filter[:env.i + 1])
    nees = np.array([[delta[i, j] @ np.linalg.inv(Pf[i, j]) @ delta[i, j] for j in range(12)] for i in range(env.i + 1)])
    near = np.minimum(np.abs(nees - cr6[0]), np.abs(nees - cr6[1])) < 1e-6 * nees      # (no entry sits on a critical point)
    assert not near.any()
    assert got['nees_total'] == nees.size and got['nees_inside'] == int(((nees > cr6[0]) * (nees < cr6[1])).sum())
    assert got['Test 4: NEES chi2'] == np.round(np.mean((nees > cr6[0]) * (nees < cr6[1])) * 100, 2)
    # the primitive on its own: NaN handling and the strict inequalities
    v = torch.tensor([0.1, np.nan, 0.5, 2.0, 9.0, 9.5, np.inf, 0.5], dtype=torch.float64, device="cuda")
    assert device.chi2_contained(v, 0.5, 9.0) == (1, 7)
    assert device.chi2_contained(v[:0], 0.5, 9.0) == (0, 0)
    big = torch.rand(300001, dtype=torch.float64, device="cuda") * 20
    inside, valid = device.chi2_contained(big, cr6[0], cr6[1])
    b = big.cpu().numpy()
    assert valid == b.size and inside == int(((b > cr6[0]) & (b < cr6[1])).sum())


def test_cowell_acceleration_hook(envs):
    """the reference's pluggable acceleration (envs/dynamics.py:168-201 fx_xyz_cowell(..., ad=ad_none, **ad_kwargs)): the tokens
    ad_none / ad_j2 bound to fx_xyz_cowell -- directly or with functools.partial, as a reference user would -- select the device
    integrator's term; ad_none reproduces two-body Farnocchia, ad_j2 the J2 extension, and the env accepts either as `fx`."""
    import functools
    from ssa_gym_amd import device, host
    from ssa_gym_amd.envs import dynamics as D
    from ssa_gym_amd.envs._config import kernel_consts, resolve_perturbation
    x = golden("catalogue_subset.npy")[5]
    kep = D.fx_xyz_farnocchia(x, 20.0)
    two = D.fx_xyz_cowell(x, 20.0)                                     # default ad = ad_none
    assert np.linalg.norm(two[:3] - kep[:3]) / np.linalg.norm(kep[:3]) < 1e-11
    fj = functools.partial(D.fx_xyz_cowell, ad=D.ad_j2, J2=host.J2_EARTH, R=host.R_EQ_EARTH)
    want = device.propagate_j2(device.as_dev(x.reshape(1, 6)), 20.0, host.J2_EARTH, host.R_EQ_EARTH, 4).cpu().numpy().reshape(6)
    assert np.array_equal(D.unwrap_partial(fj)(x, 20.0), want)
    assert np.array_equal(D.fx_xyz_cowell(x, 20.0, ad=D.ad_j2), want)
    assert np.array_equal(D.fx_xyz_cowell.with_ad(D.ad_j2, J2=2 * host.J2_EARTH)(x, 20.0),
                          device.propagate_j2(device.as_dev(x.reshape(1, 6)), 20.0, 2 * host.J2_EARTH, host.R_EQ_EARTH, 4).cpu().numpy().reshape(6))
    # the env: fx = partial(fx_xyz_cowell, ad=ad_j2) steps exactly like fx = fx_xyz_j2_rk4; ad_none like J2 = 0
    base = dict(envs.env_config)
    base.update(rso_count=8, steps=12, reward_type='trinary', obs_returned='flatten', seed=2)
    runs = {}
    for name, fx in (("token", D.fx_xyz_j2_rk4), ("partial", fj), ("none", D.fx_xyz_cowell)):
        cfg = dict(base, fx=fx)
        env = envs.make(config=cfg)
        for k in range(5):
            obs, _, _, _ = env.step(k)
        runs[name] = obs
    assert np.array_equal(runs["token"], runs["partial"])
    assert not np.array_equal(runs["token"], runs["none"])
    assert resolve_perturbation(dict(base, fx=D.fx_xyz_cowell)) == (0.0, host.R_EQ_EARTH)
    c, _ = kernel_consts(dict(base, fx=fj), env.Q, env.R, 20.0, 0.0, env.obs_lla)
    assert c.j2 == host.J2_EARTH and c.propagator == 2
    with pytest.raises(NotImplementedError):
        resolve_perturbation(dict(base, fx=D.fx_xyz_cowell.with_ad(lambda t, u, k: 0)))
    with pytest.raises(NotImplementedError):       # an acceleration next to the analytic two-body propagator has no kernel
        kernel_consts(dict(base, ad=D.ad_j2), env.Q, env.R, 20.0, 0.0, env.obs_lla)


@pytest.mark.parametrize("m,agent_name,reward", [(4100, "agent_visible_greedy", 'trinary'), (20000, "agent_visible_greedy", 'trinary'),
                                                 (20000, "agent_shannon", 'trinary'), (20160, "agent_pos_error_greedy", 'trinary'),
                                                 (2000, "agent_vel_error_greedy", 'jones'),
                                                 (2003, "agent_visible_greedy", 'trinary xyz'), (2003, "agent_shannon", 'trinary hybrid'),
                                                 (2003, "agent_visible_greedy", 'trinary elements'), (2003, "agent_visible_greedy", 'trinary j2')])
def test_persistent_closed_loop_equals_per_step_launches_at_size(envs, m, agent_name, reward):
    """ssa_env_closed_loop_f64 at sizes where the decision really crosses wavefront groups (4 100 objects: 1 025 wavefronts = 16
    groups + one wavefront; 20 000: 79 groups over all eight XCDs; 20 160: with the 80 service wavefronts every resident slot taken): same actions, bit-identical
    states and statistics as the step + select launches, through run_agent.  'jones': an episode that ends inside the run."""
    cfg = dict(envs.env_config)
    reward, _, variant = reward.partition(' ')   # (variant: the xyz measurement model / another propagator instance of the kernel; 2 003: a ragged tile)
    K = 40 if reward == 'trinary' else 190      # ('jones': max delta_pos crosses 5e6 m after ~150 predict-mostly steps)
    cfg.update(rso_count=m, steps=K + 20, reward_type=reward, obs_returned='flatten', seed=3, obs_limit=10.0,
               history=(2 if reward == 'trinary' else 'full'), device_rng=True)
    if variant == 'xyz':
        from ssa_gym_amd.envs import dynamics as D
        cfg.update(obs_type='xyz', z_sigma=(5e2,) * 3, R=np.diag([5e2 ** 2] * 3), hx=D.hx_xyz, mean_z=D.mean_xyz, residual_z=np.subtract)
    elif variant:
        cfg.update(propagator=variant)
    fallback = np.random.RandomState(9).randint(0, m, size=K + 1)
    out = {}
    for loop in ('persistent', 'per_step'):
        env = envs.make(config=dict(cfg, closed_loop=loop))
        obs, acts, rews, dones = env.run_agent(agent_name, K, fallback_actions=fallback)
        assert (getattr(env._engine, "_loop_ws", None) is not None) == (loop == 'persistent')
        out[loop] = (obs, acts, rews, dones, env.i, env.x_filter[env.i], env.P_filter[env.i], env.x_true[env.i],
                     env._engine.status.cpu().numpy(), env.obs_taken.copy(), env._y.copy())
    a, b = out['persistent'], out['per_step']
    assert list(a[1]) == list(b[1]) and len(set(a[1].tolist())) > 1
    np.testing.assert_array_equal(a[2], b[2])
    np.testing.assert_array_equal(a[3], b[3])
    assert a[4] == b[4] and (a[4] == K if reward == 'trinary' else (a[3][-1] and a[4] < K))
    for k in (0, 5, 6, 7, 8, 9):
        assert np.array_equal(a[k], b[k], equal_nan=True), k
    np.testing.assert_array_equal(a[10], b[10])


def test_closed_loop_beyond_the_resident_capacity_takes_the_per_step_launches(envs):
    """more objects than the resident wavefronts (minus the service wavefronts) x 4: the library declines (SSA_E_UNSUPPORTED) and run_agent issues the
    step + select launches instead -- same API, same results, no deadlock-prone partial grid"""
    cfg = dict(envs.env_config)
    cfg.update(rso_count=20484, steps=20, reward_type='trinary', obs_returned='aer', seed=3, obs_limit=10.0, history=2, device_rng=True)
    env = envs.make(config=cfg)
    obs, acts, rews, dones = env.run_agent("agent_visible_greedy", 6)
    assert len(acts) == 6 and env.i == 6 and obs.shape == (4 * 20484,) and np.isfinite(rews).all()


@pytest.mark.parametrize("mode", ['flatten', 'aer', 'matrix'])
def test_observation_paths_agree(envs, mode):
    """the observation step() returns -- written by the step kernel into host-mapped pinned memory (two alternating arrays; 'aer':
    the one persistent array of the reference, :362-363) -- equals the device-resident history, the opt-in CUDA-tensor form
    (config['obs_device']) equals both, and an observation stays intact for one more step."""
    import torch
    cfg = dict(envs.env_config)
    cfg.update(rso_count=203, steps=30, reward_type='trinary', obs_returned=mode, seed=6, history='full')     # (203: a ragged last tile)
    host = envs.make(config=cfg)
    dev = envs.make(config=dict(cfg, obs_device=True))
    prev = None
    for k in range(1, 9):
        o, r, d, _ = host.step(k)
        od, rd, dd, _ = dev.step(k)
        assert isinstance(od, torch.Tensor) and od.is_cuda and isinstance(o, np.ndarray)
        want = host.obs[k].reshape(-1) if mode == 'flatten' else (host.aer_obs(np.zeros(4 * 203)) if mode == 'aer' else host.obs[k])
        same = (lambda a, b: np.allclose(a, b, rtol=1e-12, atol=1e-12)) if mode == 'aer' else np.array_equal   # ('aer': the epilogue's
        assert o.shape == want.shape and same(o, want)                                                       #  two-lane atan2 vs aer_obs_kernel)
        assert np.array_equal(od.cpu().numpy().reshape(want.shape), o) and r == rd and d == dd
        if prev is not None and mode != 'aer':
            assert np.array_equal(prev[0], prev[1])          # the previous step's array has not been touched by this step
        prev = (o, o.copy())
    if mode == 'aer':
        assert o is host.observation


def test_graph_capture_survives_garbage_collection_of_older_graphs():
    """A captured hipGraph that becomes garbage (an env that went out of scope: env <-> closure cycles, so only the cyclic collector frees it)
    must not be finalised WHILE another capture is in progress: hipGraphDestroy is not permitted while a stream captures, and the error
    thrown from the graph's destructor ends the process (seen once in the GPU suite of round 4).  The captures collect first and keep the
    collector off until they end.  In a child process: the failure mode is an abort, not an exception."""
    import subprocess
    import sys
    from conftest import ROOT
    code = r'''
import gc, sys
sys.path.insert(0, %r)
import torch
from ssa_gym_amd.envs import env_config, make
cfg = dict(env_config)
cfg.update(rso_count=41, steps=200, reward_type='trinary', obs_returned='flatten', seed=13, obs_limit=5.0, history=2)
def policy(view):
    sc, mask = view.scores()
    return torch.clamp(view.argmax(sc[0], mask), min=0)
old = [make(config=cfg) for _ in range(3)]
for e in old:
    e.run_policy(policy, 64, graph=True)
    assert e.policy_graph_error is None
    e._cycle = e                             # (a reference cycle: only the cyclic collector can free this env and its graphs)
gc.collect()
gc.disable()
del old, e                                   # three captured graphs, now garbage reachable only through reference cycles: nothing frees them ...
calls = {"n": 0}
def collecting(view):                        # ... until a collection runs in the MIDDLE of the next capture (call 1 is the eager warm-up)
    calls["n"] += 1
    if calls["n"] == 4:
        gc.collect()
    return policy(view)
env = make(config=cfg)
a, _, _ = env.run_policy(collecting, 96, graph=True)
assert env.policy_graph_error is None and len(a) == 96 and calls["n"] >= 33
print("ok")
''' % (ROOT,)
    out = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), (out.returncode, out.stderr[-1500:])


@pytest.mark.parametrize("mode,reward", [('flatten', 'trinary'), ('aer', 'shaped'), ('default', 'jones')])
def test_storage_layout_is_invisible_through_the_gym_api(envs, mode, reward):
    """config['storage_layout'] = 'regime' (opt-in): the engine stores objects of one orbit regime in the same wavefronts; the env's
    numbering, actions, observations, rewards, histories and failure ids must not show it.  Two envs from the same seed, one with the
    layout and one without, stepped with the same actions: every returned value equal bit for bit; then the paths that read the device
    state as the env numbers it (history arrays, visibility, device agents, rollout, run_agent, run_policy) -- they put the state back
    first -- equal too.  A quarter of the covariances is inflated so that failures, the 'shaped' arg-max and its ties happen."""
    import torch
    cfg = dict(envs.env_config)
    cfg.update(rso_count=128, steps=120, reward_type=reward, obs_returned=mode, seed=21, history='full', obs_limit=5.0)
    a = envs.make(config=cfg)
    b = envs.make(config=dict(cfg, storage_layout='regime'))
    assert a._engine._order is None and b._engine._order is not None and not np.array_equal(b._engine._order, np.arange(128))
    rs = np.random.RandomState(3)
    wild = torch.as_tensor(rs.uniform(size=128) < 0.25).cuda()
    for env in (a, b):       # (inflate the same OBJECTS in both: positions differ under the layout)
        e = env._engine
        pos = wild if e._order is None else wild[e._order_idx]
        e.P_filter[0, pos] *= 3e4
    assert np.array_equal(a._obs_out(), b._obs_out())
    for k in range(60):
        act = int(rs.randint(128))
        ra, rb = a.step(act), b.step(act)
        assert np.array_equal(ra[0], rb[0]) and ra[1] == rb[1] and ra[2] == rb[2], k
        if ra[2]:
            break
    assert b._engine._order is not None                    # step() alone never cost the layout
    assert a.failed_filters_id == b.failed_filters_id and np.array_equal(a.rewards[:a.i + 1], b.rewards[:b.i + 1])
    assert np.array_equal(np.asarray(a.z_true[1:a.i + 1]), np.asarray(b.z_true[1:b.i + 1]), equal_nan=True)
    # ... the readers of the device state in the env's own order
    for name in ("x_true", "x_filter", "P_filter", "delta_pos", "sigma_pos"):
        assert np.array_equal(np.asarray(getattr(a, name)[a.i]), np.asarray(getattr(b, name)[b.i]), equal_nan=True), name
    assert b._engine._order is None                        # (reading the history arrays put the state back)
    assert np.array_equal(a.visible_objects(), b.visible_objects())
    if not ra[2]:
        r1, r2 = a.step(5), b.step(5)
        assert np.array_equal(r1[0], r2[0]) and r1[1] == r2[1]
    # a fresh episode sets the layout again; the multi-step entry points put the state back themselves
    for entry in ("rollout", "run_agent", "run_policy"):
        a.reset(), b.reset()
        assert b._engine._order is not None
        for k in range(3):
            assert np.array_equal(a.step(k)[0], b.step(k)[0])
        if entry == "rollout":
            ua, ub = a.rollout([7, 8, 9, 10]), b.rollout([7, 8, 9, 10])
        elif entry == "run_agent":
            ua, ub = a.run_agent('agent_visible_greedy', 6), b.run_agent('agent_visible_greedy', 6)
        else:
            def pol(view):
                sc, mask = view.scores()
                return torch.clamp(view.argmax(sc[0], mask), min=0)
            ua, ub = a.run_policy(pol, 6), b.run_policy(pol, 6)
        for u, v in zip(ua, ub):
            u, v = np.asarray(u), np.asarray(v)
            assert np.array_equal(u, v, equal_nan=(u.dtype.kind == 'f')), entry
        # (run_agent keeps the layout: the agent kernels and the persistent closed-loop launch take the table; the others put the state back)
        assert (b._engine._order is None) == (entry != "run_agent") and a.i == b.i
        if entry == "run_agent":      # ... in both of its forms: the per-step launches too
            a._closed_loop_persistent = b._closed_loop_persistent = False
            ua, ub = a.run_agent('agent_shannon', 5), b.run_agent('agent_shannon', 5)
            for u, v in zip(ua, ub):
                u, v = np.asarray(u), np.asarray(v)
                assert np.array_equal(u, v, equal_nan=(u.dtype.kind == 'f')), "run_agent, per-step launches"
            assert b._engine._order is not None
        assert np.array_equal(np.asarray(a.x_filter[a.i]), np.asarray(b.x_filter[b.i]))
    with pytest.raises(ValueError):
        envs.make(config=dict(cfg, storage_layout='sorted'))


@pytest.mark.parametrize("m,E", [(64, 3), (12000, 2)])
@pytest.mark.parametrize("mode,reward,dev", [('aer', 'trinary', True), ('flatten', 'shaped', True), ('default', 'jones', False)])
def test_storage_layout_of_a_vector_env(envs, mode, reward, dev, m, E):
    """config['storage_layout'] = 'regime' for SSA_Tasker_VecEnv: one permutation per env (HotPathEngine.set_layout with [n_env][n_obj],
    ssa_step_params.obj_ids indices within the env; catalogue.regime_order_env), replaced for one env when it auto-resets.  Two vector envs
    from the same seeds, one with the layout: observations (device tensors and host arrays), rewards, dones, terminal observations and the
    per-env state readers equal bit for bit -- through the one-tile launch (3 x 64 objects) and the grid-stride one (2 x 12 000), across an
    auto-reset (5-step episodes), with inflated covariances so that the 'shaped' arg-max has work."""
    import torch
    from ssa_gym_amd.envs.vector_env import SSA_Tasker_VecEnv
    cfg = dict(envs.env_config)
    cfg.update(rso_count=m, steps=6, reward_type=reward, obs_returned=mode, device_rng=True, obs_device=dev, obs_limit=5.0)
    a = SSA_Tasker_VecEnv(dict(cfg), num_envs=E, seed=4)
    b = SSA_Tasker_VecEnv(dict(cfg, storage_layout='regime'), num_envs=E, seed=4)
    eb = b._eng
    assert a._eng._order is None and eb._order.shape == (E, m) and not np.array_equal(eb._order[E - 1], np.arange(m))
    host = lambda o: o.cpu().numpy() if dev else np.asarray(o)      # noqa: E731
    rs = np.random.RandomState(8)
    wild = torch.as_tensor(rs.uniform(size=E * m) < 0.25).cuda()
    a._eng.P_filter[a.tick % 2, wild] *= 3e4
    eb.P_filter[b.tick % 2, wild[eb._order_idx]] *= 3e4                # (the same OBJECTS)
    for k in range(14):                                                # (two auto-resets of every env)
        acts = rs.randint(m, size=E)
        oa, ra, da, ia = a.step(acts)
        ob, rb, db, ib = b.step(acts)
        assert np.array_equal(host(oa), host(ob), equal_nan=True), k
        assert np.array_equal(ra, rb) and np.array_equal(da, db), k
        for x, y in zip(ia, ib):
            assert ('terminal_observation' in x) == ('terminal_observation' in y)
            if 'terminal_observation' in x:
                assert np.array_equal(host(x['terminal_observation']), host(y['terminal_observation']), equal_nan=True)
        if k in (2, 7):
            for e in range(E):
                assert np.array_equal(a.x_filter(e), b.x_filter(e), equal_nan=True) and np.array_equal(a.P_filter(e), b.P_filter(e), equal_nan=True)
                assert np.array_equal(a.x_true(e), b.x_true(e))
    with pytest.raises(ValueError):
        SSA_Tasker_VecEnv(dict(cfg, storage_layout='sorted'), num_envs=E, seed=4)
    if m == 64:      # what the several-env table cannot express is refused, loudly
        from ssa_gym_amd import _lib
        with pytest.raises(ValueError):
            SSA_Tasker_VecEnv(dict(cfg, rso_count=66, storage_layout='regime'), num_envs=E, seed=4)      # (whole tiles per env)
        with pytest.raises(_lib.SsaHipError):
            eb.set_layout(np.arange(m))                                                                   # (one permutation per env)
        with pytest.raises(_lib.SsaHipError):
            eb.set_env_layout(0, np.zeros(m, dtype=np.int64))
        with pytest.raises(_lib.SsaHipError):
            a._eng.set_env_layout(0, np.arange(m))                                                        # (no layout set)
        eb.to_caller_order()                                                                              # the state back in the envs' own order
        assert eb._order is None
        for e in range(E):
            assert np.array_equal(a._eng.x_filter[a.tick % 2, e * m:(e + 1) * m].cpu().numpy(), eb.x_filter[b.tick % 2, e * m:(e + 1) * m].cpu().numpy(), equal_nan=True)
            assert np.array_equal(a._eng.metrics[a.tick % 2, e].cpu().numpy(), eb.metrics[b.tick % 2, e].cpu().numpy(), equal_nan=True)
        assert np.array_equal(a._eng.status.cpu().numpy(), eb.status.cpu().numpy())


@pytest.mark.parametrize("mode", ['flatten', 'aer', 'default'])
def test_float32_observations_are_the_float64_ones_rounded(envs, mode):
    """config['obs_dtype'] = np.float32 (EXTENSION; the reference's observations are float64): the step kernel writes its host-facing copy
    of the observation in single precision (SSA_LAUNCH_MIRROR_F32: half the bytes over PCIe).  Everything computed stays float64, so the
    float32 observation must be exactly the float64 one rounded to nearest -- every step, every mode, the vector env too -- and rewards,
    dones and the device state identical."""
    from ssa_gym_amd.envs.vector_env import SSA_Tasker_VecEnv
    cfg = dict(envs.env_config)
    cfg.update(rso_count=37, steps=40, reward_type='trinary', obs_returned=mode, seed=5, history='full')
    a, b = envs.make(config=cfg), envs.make(config=dict(cfg, obs_dtype=np.float32))
    assert b.observation_space.dtype == np.float32 and a.observation_space.dtype == np.float64
    oa, ob = a.reset(), b.reset()
    assert ob.dtype == np.float32 and np.array_equal(oa.astype(np.float32), ob)
    for k in range(12):
        ra, rb = a.step(k % 37), b.step(k % 37)
        assert rb[0].dtype == np.float32 and rb[0].shape == ra[0].shape
        assert np.array_equal(ra[0].astype(np.float32), rb[0]) and ra[1] == rb[1] and ra[2] == rb[2], k
    assert np.array_equal(np.asarray(a.x_filter[12]), np.asarray(b.x_filter[12])) and np.asarray(b.x_filter[12]).dtype == np.float64
    ua, ub = a.rollout([1, 2, 3]), b.rollout([1, 2, 3])
    assert ub[0].dtype == np.float32 and np.array_equal(ua[0].astype(np.float32), ub[0])
    va = SSA_Tasker_VecEnv(dict(cfg), num_envs=3, seed=2)
    vb = SSA_Tasker_VecEnv(dict(cfg, obs_dtype=np.float32), num_envs=3, seed=2)
    for k in range(6):
        ra, rb = va.step([k, k + 1, k + 2]), vb.step([k, k + 1, k + 2])
        assert rb[0].dtype == np.float32 and np.array_equal(ra[0].astype(np.float32), rb[0]) and np.array_equal(ra[1], rb[1])
    # ... and together with the per-env storage layout (single-precision rows written at the envs' own indices; whole tiles per env)
    cfg36 = dict(cfg, rso_count=36)
    va = SSA_Tasker_VecEnv(dict(cfg36), num_envs=3, seed=2)
    vc = SSA_Tasker_VecEnv(dict(cfg36, obs_dtype=np.float32, storage_layout='regime'), num_envs=3, seed=2)
    for k in range(6):
        ra, rc = va.step([k, k + 1, k + 2]), vc.step([k, k + 1, k + 2])
        assert rc[0].dtype == np.float32 and np.array_equal(ra[0].astype(np.float32), rc[0]) and np.array_equal(ra[1], rc[1])
    with pytest.raises(ValueError):
        envs.make(config=dict(cfg, obs_dtype=np.int32))
