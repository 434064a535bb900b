"""Host side of SSA_Tasker_Env.step() at 20 000 objects ('flatten', fresh arrays from the pool): launch call, synchronisation, the rest."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from ssa_gym_amd.envs import env_config, make
m = 20000
cfg = dict(env_config)
cfg.update(rso_count=m, steps=480, reward_type='trinary', obs_returned=os.environ.get('MODE', 'flatten'), seed=0, history=2, device_rng=True,
           obs_device=os.environ.get("DEV") == "1")
env = make(config=cfg)
for k in range(20):
    env.step(k % m)
acc = {"launch": 0.0, "sync": 0.0}
real_launch = env._engine.launch_step
def launch(*a, **kw):
    t = time.perf_counter(); r = real_launch(*a, **kw); acc["launch"] += time.perf_counter() - t; return r
env._engine.launch_step = launch
class S:
    def __init__(self, s): self.s = s; self.cuda_stream = s.cuda_stream
    def synchronize(self):
        t = time.perf_counter(); self.s.synchronize(); acc["sync"] += time.perf_counter() - t
env._stream = S(env._stream)
N = 200
torch.cuda.synchronize()
t0 = time.perf_counter()
for k in range(N):
    o = env.step(k % m)
tot = time.perf_counter() - t0
print("per step: total %.1f us = launch call %.1f + synchronize %.1f + the rest of step() %.1f" % (
    1e6 * tot / N, 1e6 * acc["launch"] / N, 1e6 * acc["sync"] / N, 1e6 * (tot - acc["launch"] - acc["sync"]) / N))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for k in range(N):
    o = env.step(k % m)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
