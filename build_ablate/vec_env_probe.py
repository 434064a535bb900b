"""Where a vector-env step's time goes (8 envs x 20 000 objects, observations left on the GPU): wall time per vector step over the
early (steps 1-200) and the late (steps 300-470) part of an episode; under rocprofv3 --kernel-trace the kernels' own durations stand next
to it.   LAYOUT=1: config['storage_layout'] = 'regime'."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from ssa_gym_amd.envs import env_config
from ssa_gym_amd.envs.vector_env import SSA_Tasker_VecEnv
m, E = 20000, 8
cfg = dict(env_config)
cfg.update(rso_count=m, steps=480, reward_type='trinary', obs_returned=os.environ.get('MODE', 'aer'), seed=0, device_rng=True, obs_device=True,
           storage_layout='regime' if os.environ.get("LAYOUT") == "1" else None)
env = SSA_Tasker_VecEnv(cfg, num_envs=E, seed=0)
acts = lambda k: [(k * 7 + 13 * e) % m for e in range(E)]    # noqa: E731
for k in range(10):
    env.step(acts(k))
for ep in range(int(os.environ.get("EPISODES", 2))):
    env.reset()
    marks = {}
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(1, 471):
        env.step(acts(k))
        if k in (200, 300, 470):
            marks[k] = time.perf_counter() - t0
    print("episode %d: steps 1-200 %.1f us per vector step, steps 301-470 %.1f us" % (ep, 1e6 * marks[200] / 200, 1e6 * (marks[470] - marks[300]) / 170), flush=True)
