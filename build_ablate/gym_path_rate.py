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
# BASELINE config 5 per GPU: 8 env instances x 20 000 objects in ONE launch per vector step
from ssa_gym_amd.envs.vector_env import SSA_Tasker_VecEnv
for E, m, mode in ((8, 20000, 'aer'), (8, 20000, 'flatten'), (64, 2000, 'aer')):
    cfg = dict(env_config); cfg.update(rso_count=m, steps=480, reward_type='trinary', obs_returned=mode, seed=0)
    venv = SSA_Tasker_VecEnv(cfg, E, seed=0)
    for k in range(10): venv.step([(k + e) % m for e in range(E)])
    t0 = time.perf_counter(); n = 100
    for k in range(n): venv.step([(10 + k + e) % m for e in range(E)])
    dt = (time.perf_counter() - t0) / n
    print('vec E=%d m=%d obs=%s: %.1f us/vector-step = %.0f env-steps/s = %.0f 20k-object env-steps/s (gym vector API, PCIe + sync inclusive)'
          % (E, m, mode, dt * 1e6, E / dt, E / dt * m / 20000), flush=True)
