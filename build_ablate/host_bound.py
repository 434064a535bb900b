"""is the per-step launch loop host-bound?  enqueue time vs completion time of 2000 steps (diagnostic)."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
sys.argv = ['bench.py']
import bench
from ssa_gym_amd import host, engine, parallel, _build
if os.environ.get('LIB'):
    _build.LIB = os.environ['LIB']
print('lib', _build.LIB)
m = 20000
pb = bench.build_problem(m, seed=100)
consts = host.make_consts(pb["Q"], pb["R"], 1e-4, 2.0, -3, 20.0, -np.pi / 2, pb["obs_lla"], obs_type='aer', propagator='fg')
z = torch.zeros((1, 480, m, 3), dtype=torch.float64, device='cuda')
eng = engine.HotPathEngine(consts, m, 1, pb["trans"], z, history=2)
eng.load_state(0, pb["x_true"], pb["x"], np.broadcast_to(pb["P0"], (m, 6, 6)))
for defer in (True, False):
    local = parallel.HipLocalStepper(eng, consts, fast_stats=True, defer_fold=defer)
    local.load_schedule(np.arange(4000) % m)
    snap = eng.snapshot(0)
    for rep in range(3):
        local.reset_episode(snap, 480); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(400):
            local.step(-1)
        t1 = time.perf_counter()
        local.flush(); torch.cuda.synchronize()
        t2 = time.perf_counter()
        print("defer_fold", defer, "enqueue us/step %.2f   completion us/step %.2f" % ((t1 - t0) / 400 * 1e6, (t2 - t0) / 400 * 1e6), flush=True)
# raw ctypes loop (what time_variants does)
p = eng._p; lib = eng.lib if hasattr(eng, 'lib') else None
