"""host cost of enqueuing one env step (no GPU wait) through each layer"""
import os, sys, time, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
sys.argv = ['bench.py']
import bench
from ssa_gym_amd import host, engine, parallel
for m in (2000, 20000):
    pb = bench.build_problem(m, seed=100)
    consts = host.make_consts(pb["Q"], pb["R"], 1e-4, 2.0, -3, 20.0, -np.pi / 2, pb["obs_lla"])
    z = torch.zeros((1, 480, m, 3), dtype=torch.float64, device='cuda')
    eng = engine.HotPathEngine(consts, m, 1, pb["trans"], z, history=2)
    eng.load_state(0, pb["x_true"], pb["x"], np.broadcast_to(pb["P0"], (m, 6, 6)))
    local = parallel.HipLocalStepper(eng, consts, fast_stats=True, defer_fold=True)
    local.load_schedule(np.arange(4000) % m)

    def t(f, n=200):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): f()
        t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        return round((t1 - t0) / n * 1e6, 2), round((t2 - t0) / n * 1e6, 2)
    print(m, 'local.step (deferred fold)  host / total us per step', t(lambda: local.step(-1)))
    local.flush()
    s = torch.cuda.current_stream().cuda_stream
    eng._p.launch_mask = 8
    print(m, 'ctypes ssa_env_step_f64 only  host / total us per step', t(lambda: eng._lib.ssa_env_step_f64(eng._cref, eng._pref, s)))
    eng._p.launch_mask = 0
