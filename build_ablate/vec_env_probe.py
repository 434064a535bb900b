"""where the time of SSA_Tasker_VecEnv.step() goes (8 envs x 20 000 objects, 'aer'): the whole call, then the pieces of it."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from ssa_gym_amd import _lib
from ssa_gym_amd.envs import env_config
from ssa_gym_amd.envs.vector_env import SSA_Tasker_VecEnv
m, E, N = 20000, 8, 200
cfg = dict(env_config)
cfg.update(rso_count=m, steps=480, reward_type='trinary', obs_returned='aer', seed=0, device_rng=True)
env = SSA_Tasker_VecEnv(cfg, num_envs=E, seed=0)
acts = lambda k: [(k * 7 + 13 * e) % m for e in range(E)]    # noqa: E731
for k in range(10):
    env.step(acts(k))


def timeit(name, fn, n=N):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(n):
        fn(k)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print("%-86s %7.1f us" % (name, dt * 1e6), flush=True)


timeit("env.step(): the whole vector step", lambda k: env.step(acts(k + 10)))
e = env._eng
cur = torch.cuda.current_stream()
dev_aer = torch.zeros((E * m, 4), dtype=torch.float64, device="cuda")
state = {"tick": env.tick}


def piece(aer_ptr, copy, sync=True, stats_host=True):
    def f(k):
        state["tick"] += 1
        t = state["tick"]
        if copy:
            e.time_actions.copy_(env._ta_host, non_blocking=True)
        e.launch_step((t - 1) % 2, t % 2, 0, aer_out=aer_ptr, stats_out=env._stats_host.data_ptr() if stats_host else 0,
                      stream=cur.cuda_stream, fast_stats=True)
        if sync:
            cur.synchronize()
    return f


env._time_np[:] = 5


def inline_piece(aer_ptr, sync=True):
    def f(k):
        state["tick"] += 1
        t = state["tick"]
        e.launch_step((t - 1) % 2, t % 2, 0, aer_out=aer_ptr, stats_out=env._stats_host.data_ptr(), stream=cur.cuda_stream,
                      fast_stats=True, fold_inside=True, env_words=([5] * E, acts(k)))
        if sync:
            cur.synchronize()
    return f


timeit("ONE launch (words by value, folds inside; obs -> host-mapped) + sync", inline_piece(env._obs_ring_ptr[0]))
timeit("ONE launch (words by value, folds inside; obs -> device) + sync", inline_piece(dev_aer.data_ptr()))
timeit("ONE launch (words by value, folds inside; obs -> host-mapped), no per-step sync", inline_piece(env._obs_ring_ptr[0], sync=False))
timeit("ONE launch (words by value, folds inside; obs -> device), no per-step sync", inline_piece(dev_aer.data_ptr(), sync=False))
timeit("copy + launch (obs -> host-mapped) + sync", piece(env._obs_ring_ptr[0], True))
timeit("       launch (obs -> host-mapped) + sync", piece(env._obs_ring_ptr[0], False))
timeit("copy + launch (obs -> device) + sync", piece(dev_aer.data_ptr(), True))
timeit("       launch (obs -> device) + sync", piece(dev_aer.data_ptr(), False))
timeit("       launch (no aer payload) + sync", piece(0, False))
timeit("       launch (obs -> device), no per-step sync", piece(dev_aer.data_ptr(), False, sync=False))
timeit("       launch (obs -> host-mapped), no per-step sync", piece(env._obs_ring_ptr[0], False, sync=False))
a = np.asarray(acts(3), dtype=np.int64)


def host_only(k):
    actions = np.asarray(acts(k), dtype=np.int64).reshape(E)
    assert np.all((actions >= 0) & (actions < m))
    env._time_np[:] = env.i
    env._act_np[:] = actions
    st = env._stats_np.copy()
    ap = st[:, _lib.STAT_ARGMAX_SPOS].astype(np.int64)
    mx = st[:, _lib.STAT_MAX_DPOS]
    last = env.i + 1 >= env.n
    rewards = (st[:, _lib.STAT_CNT_LT_1E4] + st[:, _lib.STAT_CNT_LT_1E7]) / m / 2
    dones = last.copy()
    env.rewards_sum += rewards
    infos = [{} for _ in range(E)]
    dones.any()
    rewards = np.where(np.isfinite(rewards), rewards, 0.5)


timeit("host arithmetic of step() alone (numpy on 8-vectors, no GPU call)", host_only)
timeit("acts(k) (the bench's action list)", lambda k: acts(k))
