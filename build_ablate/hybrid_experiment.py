import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
from ssa_gym_amd import _lib, _build, device, host, engine
_build.LIB = os.path.join(ROOT, "build_ablate", "exp", "hybrid.so")
_lib.load()
import episode_workload as ew
sys.argv = ['bench.py']
import bench
class H: pass
hip = H(); hip.torch, hip.lib, hip.dev, hip.host, hip.engine = torch, _lib, device, host, engine
for seed in (7, 8):
    w = ew.workload(seed=seed)
    print(seed, "oracle", ew.run_oracle(w)["failed_at"], flush=True)
    r = ew.run_hip(hip, w, "elements")
    print(seed, "hybrid+refcov", r["failed_at"], r["status_mix"], r["jones_done_step"], flush=True)
print("rate", bench.local_variant_rate(20000, 958, 0, "elements"))
