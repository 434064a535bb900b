"""closed-loop kernel: what each ingredient costs (diagnostic builds in build_ablate/cl/*.so; see build_closed_loop_variants.sh)"""
import ctypes
import glob
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.argv = ['bench.py']
import bench  # noqa: E402
from ssa_gym_amd import _lib, _build  # noqa: E402
m = int(os.environ.get('M', '20000'))
paths = sorted(glob.glob(os.path.join(ROOT, 'build_ablate', 'cl', '*.so')))
for path in paths:
    _lib._lib = None
    _build.LIB = path
    lib = _lib.load()
    for agent, name in ((0, 'naive_greedy'), (1, 'visible_greedy'), (2, 'shannon')):
        if 'nowait' in path or 'notree' in path:
            if agent == 2:
                continue
        r = bench.closed_loop_rate(m, 480, 50, agent=agent)
        print("%-28s %-16s %8.2f us/step  %9.0f env-steps/s  spread %s" % (os.path.basename(path), name, 1e3 * r["ms_per_step"], r["value"], r["value_spread"]), flush=True)
