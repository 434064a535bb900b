"""where env.step() spends its time at 20 000 objects (cProfile + the env's own runtime counters; diagnostic)."""
import cProfile, pstats, io, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
from ssa_gym_amd.envs import env_config, make
m = 20000
for mode in ('flatten', 'aer'):
    cfg = dict(env_config)
    cfg.update(rso_count=m, steps=480, reward_type='trinary', obs_returned=mode, seed=0, history=2, device_rng=True)
    env = make(config=cfg)
    for k in range(int(os.environ.get('START', 20))):
        env.step(k % m)
    pr = cProfile.Profile()
    t0 = time.perf_counter()
    pr.enable()
    N = int(os.environ.get('N', 300))
    for k in range(N):
        env.step((20 + k) % m)
    pr.disable()
    dt = (time.perf_counter() - t0) / N
    print("==== %s: %.1f us per step (with profiler)" % (mode, dt * 1e6))
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(14); print(s.getvalue()[:3000])
    print({k: round(v / 320 * 1e6, 1) for k, v in env.runtime.items()})
