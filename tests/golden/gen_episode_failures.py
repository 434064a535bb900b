"""Generates tests/golden/episode_failures_oracle.json: failure statistics of one 2 000-object, 479-step round-robin episode
(env defaults, alpha = 1e-4) on the CPU oracle (oracle/ssa_oracle.c: reference order of operations).  Run from the repo root:
    python tests/golden/gen_episode_failures.py
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import episode_workload as ew  # noqa: E402
import oracle as orc  # noqa: E402

orc.build()
out = {}
for seed in (7, 8):
    w = ew.workload(m=2000, seed=seed)
    out["seed%d" % seed] = {"reference_order": ew.run_oracle(w), "centred_means": ew.run_oracle(w, centred=True),
                            "reference_order_resample": ew.run_oracle(w, resample=True)}
json.dump(out, open(os.path.join(HERE, "episode_failures_oracle.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
