"""GPU timeline of env.run_policy() (rocprofv3 --kernel-trace): which kernels one step of the graph-replayed / eager closed loop with a
torch policy consists of, how long each runs and how long the GPU idles between them.

    cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/x/trace -- python3 $R/build_ablate/run_policy_trace.py
    python3 build_ablate/run_policy_trace.py --reduce gpurun_out/x/trace
"""
import os, sys, glob, csv
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)


def reduce(d):
    files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    rows = []
    for f in files:
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    # the marker launches (ssa::propagate_kernel on 7 states) bracket the sections
    marks = [i for i, r in enumerate(rows) if "propagate_kernel" in r[2]]
    names = ["graph replay, torch policy", "eager, torch policy", "graph replay, preallocated action", "eager, preallocated action",
             "graph replay, view.scores() + view.argmax() (the arg-max head of the library)"]
    for s in range(len(marks) - 1):
        seg = rows[marks[s] + 1:marks[s + 1]]
        if not seg:
            continue
        print("== %s: %d dispatches over %.1f us" % (names[s] if s < len(names) else s, len(seg), (seg[-1][1] - seg[0][0]) / 1e3))
        per = {}
        prev_end = None
        for (a, b, n) in seg:
            short = n.split("(")[0][-70:]
            e = per.setdefault(short, [0, 0.0, 0.0])
            e[0] += 1; e[1] += (b - a) / 1e3
            if prev_end is not None:
                e[2] += max(0, a - prev_end) / 1e3
            prev_end = b
        for k, (c, dur, gap) in sorted(per.items(), key=lambda kv: -kv[1][1]):
            print("   %5d x %-70s run %7.2f us  idle before it %6.2f us" % (c, k, dur / c, gap / c))
        steps = sum(c for k, (c, _, _) in per.items() if "step_fast_kernel" in k)
        if steps:
            print("   -> %.2f us per step on the GPU (%d steps), busy %.2f us" % ((seg[-1][1] - seg[0][0]) / 1e3 / steps, steps,
                                                                              sum(v[1] for v in per.values()) / steps))


if len(sys.argv) > 2 and sys.argv[1] == "--reduce":
    reduce(sys.argv[2]); sys.exit(0)

import torch
from ssa_gym_amd.envs import env_config, make
from ssa_gym_amd import device
cfg = dict(env_config)
cfg.update(rso_count=20000, steps=480, reward_type='trinary', obs_returned='flatten', seed=0, history=2, device_rng=True, obs_limit=10.0)
if os.environ.get("PROP"):
    from ssa_gym_amd.envs import dynamics
    cfg['fx'] = getattr(dynamics, os.environ["PROP"])
env = make(config=cfg)
fixed = torch.zeros(1, dtype=torch.int32, device="cuda")
marker = torch.zeros((7, 6), dtype=torch.float64, device="cuda") + 7e6


def policy(view):
    sc, mask = view.scores()
    return torch.argmax(torch.where(mask.view(torch.bool), sc[0], float("-inf"))).to(torch.int32).reshape(1)


def trivial(view):
    return fixed


def lean(view):
    sc, mask = view.scores()
    return view.argmax(sc[0], mask)


cases = [(policy, True), (policy, False), (trivial, True), (trivial, False)]
if hasattr(env.PolicyView, "argmax"):
    cases.append((lean, True))
for pol, graph in cases:
    env.reset(); env.run_policy(pol, 64, graph=graph)        # captures, first launches
    torch.cuda.synchronize()
device.propagate(marker, 1.0); torch.cuda.synchronize()
for pol, graph in cases:
    env.reset(); env.run_policy(pol, 32, graph=graph); torch.cuda.synchronize()
    env.run_policy(pol, 64, graph=graph); torch.cuda.synchronize()
    device.propagate(marker, 1.0); torch.cuda.synchronize()
