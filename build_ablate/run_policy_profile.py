"""where env.run_policy()'s host time goes (cProfile over 400 steps at 20 000 objects)"""
import cProfile, pstats, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from ssa_gym_amd.envs import env_config, make
cfg = dict(env_config)
cfg.update(rso_count=20000, steps=480, reward_type='trinary', obs_returned='flatten', seed=0, history=2, device_rng=True, obs_limit=10.0)
env = make(config=cfg)
fixed = torch.zeros(1, dtype=torch.int32, device="cuda")


def trivial(view):
    return fixed


def policy(view):
    sc, mask = view.scores()
    masked = torch.where(mask.bool(), sc[0], torch.full_like(sc[0], -float("inf")))
    return torch.argmax(masked).to(torch.int32).reshape(1)


for name, pol in (("trivial policy (a preallocated tensor)", trivial), ("torch policy (scores + where + argmax)", policy)):
    env.reset(); env.run_policy(pol, 20)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    env.run_policy(pol, 200)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("%s: %.1f us per step" % (name, dt / 200 * 1e6), flush=True)
env.reset()
pr = cProfile.Profile(); pr.enable()
env.run_policy(policy, 200)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
