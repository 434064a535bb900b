"""Generates tests/golden/episode_failures_oracle.json: failure statistics of 2 000-object, 479-step round-robin episodes (env defaults,
alpha = 1e-4; five workloads = seeds) on the CPU oracle (oracle/ssa_oracle.c: reference order of operations, and with centred means --
the oracle's two summation orders are the yardstick for how far two faithful implementations of the same arithmetic may part): counts
per 60-step window, status-code mix, the 'jones' termination step, WHICH filters failed and at which step.  Run from the repo root:
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
for seed in ew.SEEDS:
    w = ew.workload(m=2000, seed=seed)
    out["seed%d" % seed] = {"reference_order": ew.run_oracle(w), "centred_means": ew.run_oracle(w, centred=True)}
    if seed in (7, 8):
        out["seed%d" % seed]["reference_order_resample"] = ew.run_oracle(w, resample=True)
    a, b = out["seed%d" % seed]["reference_order"], out["seed%d" % seed]["centred_means"]
    print("seed %d: failed %d / %d, overlap of the two summation orders %.3f" % (seed, a["failed_at"][479], b["failed_at"][479],
                                                                                 ew.jaccard(a["failed_ids"], b["failed_ids"])), flush=True)
json.dump(out, open(os.path.join(HERE, "episode_failures_oracle.json"), "w"), indent=1)

