"""env.step() rate through the gym API (host in the loop: launch, sync, stats + observation to the host)."""
import os, sys, time, numpy as np
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,ROOT)
from ssa_gym_amd.envs import env_config, make
for m, mode in ((20,'flatten'),(2000,'flatten'),(20000,'flatten'),(20000,'aer')):
    cfg=dict(env_config); cfg.update(rso_count=m, steps=480, reward_type='trinary', obs_returned=mode, seed=0, history=2)
    env=make(config=cfg); env.reset()
    for k in range(20): env.step(k % m)
    t0=time.perf_counter(); n=300
    for k in range(n): env.step((20+k) % m)
    dt=(time.perf_counter()-t0)/n
    print('m=%d obs=%s: %.1f us/step = %.0f env-steps/s (gym API, PCIe + sync inclusive)'%(m,mode,dt*1e6,1/dt), flush=True)
