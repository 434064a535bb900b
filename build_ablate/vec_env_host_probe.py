"""Host side of a vector-env step (8 envs x 20 000 objects, observations on the GPU): how long the launch call, the stream synchronisation
and the rest of step() take, over the early part of an episode."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from ssa_gym_amd.envs import env_config
from ssa_gym_amd.envs.vector_env import SSA_Tasker_VecEnv
m, E = 20000, 8
cfg = dict(env_config)
cfg.update(rso_count=m, steps=480, reward_type='trinary', obs_returned=os.environ.get('MODE', 'aer'), seed=0, device_rng=True,
           obs_device=os.environ.get("DEV", "1") == "1", storage_layout='regime' if os.environ.get("LAYOUT") == "1" else None)
env = SSA_Tasker_VecEnv(cfg, num_envs=E, seed=0)
acts = [np.array([(k * 7 + 13 * e) % m for e in range(E)]) for k in range(300)]
for k in range(10):
    env.step(acts[k])
env.reset()
acc = {"launch": 0.0, "sync": 0.0}
real_launch = env._eng.launch_step
def launch(*a, **kw):
    t = time.perf_counter(); r = real_launch(*a, **kw); acc["launch"] += time.perf_counter() - t; return r
env._eng.launch_step = launch
class S:
    def __init__(self, s): self.s = s; self.cuda_stream = s.cuda_stream
    def synchronize(self):
        t = time.perf_counter(); self.s.synchronize(); acc["sync"] += time.perf_counter() - t
env._stream = S(env._stream)
N = 200
torch.cuda.synchronize()
t0 = time.perf_counter()
for k in range(N):
    env.step(acts[k])
tot = time.perf_counter() - t0
print("per vector step: total %.1f us = launch call %.1f + synchronize %.1f + the rest of step() %.1f" % (
    1e6 * tot / N, 1e6 * acc["launch"] / N, 1e6 * acc["sync"] / N, 1e6 * (tot - acc["launch"] - acc["sync"]) / N))
import cProfile, pstats
env.reset()
pr = cProfile.Profile(); pr.enable()
for k in range(N):
    env.step(acts[k])
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(12)
