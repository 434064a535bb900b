"""Soak of the vector env's per-env storage layout: E envs x m objects, EPISODES whole episodes each with auto-resets (every env draws a new
catalogue and gets a new permutation), every step's observations, rewards and dones hashed -- with and without config['storage_layout'] =
'regime', for the 'aer' observations on the device and the 'flatten' ones on the host.  The digests must agree."""
import hashlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from ssa_gym_amd.envs import env_config
from ssa_gym_amd.envs.vector_env import SSA_Tasker_VecEnv
m, E = int(os.environ.get("M", 20000)), int(os.environ.get("E", 8))
episodes, steps = int(os.environ.get("EPISODES", 3)), int(os.environ.get("STEPS", 480))
for mode, dev, reward in (('aer', True, 'trinary'), ('flatten', False, 'shaped'), ('default', True, 'jones')):
    digests, fails = [], []
    for layout in (None, 'regime'):
        cfg = dict(env_config)
        cfg.update(rso_count=m, steps=steps, reward_type=reward, obs_returned=mode, seed=0, device_rng=True, obs_device=dev, storage_layout=layout)
        env = SSA_Tasker_VecEnv(cfg, num_envs=E, seed=11)
        h = hashlib.sha256()
        rs = np.random.RandomState(5)
        t0 = time.perf_counter()
        resets = 0
        for k in range(episodes * (steps - 1)):
            obs, rew, done, info = env.step(rs.randint(m, size=E))
            h.update((obs.cpu().numpy() if dev else np.asarray(obs)).tobytes()); h.update(rew.tobytes()); h.update(done.tobytes())
            resets += int(done.sum())
        digests.append(h.hexdigest()[:16])
        fails.append(int((env._eng.status != 0).sum().item()))
        print("%-8s obs_device=%-5s reward=%-8s layout=%-7s %d vector steps, %d env resets, %.1f s, failed filters now %d, sha256 %s" % (
            mode, dev, reward, layout, episodes * (steps - 1), resets, time.perf_counter() - t0, fails[-1], digests[-1]), flush=True)
        del env
    assert digests[0] == digests[1] and fails[0] == fails[1], (mode, digests, fails)
print("vector env: observations, rewards and dones of every step identical with and without the per-env storage layout")
