import os, sys, time, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from ssa_gym_amd.envs import env_config, make
for m in (20, 2000, 20000):
    for dev in (False, True):
        cfg = dict(env_config); cfg.update(rso_count=m, steps=480, seed=0, history=2, device_rng=dev)
        env = make(config=cfg)
        env.reset()
        t0 = time.perf_counter(); env.reset(); t1 = time.perf_counter()
        print("m=%d device_rng=%s: reset %.1f ms" % (m, dev, (t1 - t0) * 1e3), flush=True)
