"""episode-level failure statistics: oracle vs HIP elements / fg (2 000 objects, 479 round-robin steps, env defaults)"""
import json
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import ssa_gym_amd  # noqa: E402
from ssa_gym_amd import _lib, device, host, engine  # noqa: E402
import episode_workload as ew  # noqa: E402


class H:
    pass


hip = H()
hip.torch, hip.lib, hip.dev, hip.host, hip.engine = torch, _lib, device, host, engine
ssa_gym_amd.build()
_lib.load()
out = {}
for seed in (7, 8):
    w = ew.workload(seed=seed)
    for rs in (False, True):
        tag = "seed%d%s" % (seed, "_resample" if rs else "")
        out[tag] = {"oracle": ew.run_oracle(w, resample=rs), "oracle_centred": ew.run_oracle(w, centred=True, resample=rs),
                    "elements_refcov": ew.run_hip(hip, w, "elements", resample=rs, covariance='reference'),
                    "elements_centred": ew.run_hip(hip, w, "elements", resample=rs, covariance='centred'),
                    "hybrid_refcov": ew.run_hip(hip, w, "hybrid", resample=rs, covariance='reference'),
                    "fg_refcov": ew.run_hip(hip, w, "fg", resample=rs, covariance='reference'),
                    "fg_centred": ew.run_hip(hip, w, "fg", resample=rs, covariance='centred')}
        for k, v in out[tag].items():
            print(tag, k, json.dumps(v), flush=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "episode_failures.json"), "w"), indent=1)
