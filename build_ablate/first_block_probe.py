"""Why is the FIRST timed 20-step block of the bench 40 ms long?  Host time of every enqueue and of the fences around the first blocks."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
sys.argv = ['bench.py']
import bench
from ssa_gym_amd import host, engine, parallel
from ssa_gym_amd.catalogue import regime_order
m = 20000
pb = bench.build_problem(m, seed=100)
consts = host.make_consts(pb["Q"], pb["R"], 1e-4, 2.0, -3, 20.0, -np.pi / 2, pb["obs_lla"], obs_type='aer', propagator='hybrid')
gen = torch.Generator(device="cuda").manual_seed(1)
z = torch.randn((1, 480, m, 3), dtype=torch.float64, device='cuda', generator=gen) * torch.as_tensor(pb["z_sigma"], device="cuda")
eng = engine.HotPathEngine(consts, m, 1, pb["trans"], z, history=2)
eng.set_layout(regime_order(pb["x_true"]))
eng.load_state(0, pb["x_true"], pb["x"], np.broadcast_to(pb["P0"], (m, 6, 6)))
local = parallel.HipLocalStepper(eng, consts, fast_stats=True, defer_fold=True)
local.load_schedule(np.arange(479) % m)
snap = eng.snapshot(0)
def fence():
    local.flush(); torch.cuda.synchronize()
for k in range(5):
    local.step(-1)
fence()
for b in range(4):
    t0 = time.perf_counter(); ts = []
    for k in range(20):
        t1 = time.perf_counter(); local.step(-1); ts.append(time.perf_counter() - t1)
    t2 = time.perf_counter(); local.flush(); t3 = time.perf_counter(); torch.cuda.synchronize(); t4 = time.perf_counter()
    print("block %d: %.2f ms; enqueue max %.3f ms (step %d), sum %.3f; flush %.3f; synchronize %.3f" % (
        b, 1e3 * (t4 - t0), 1e3 * max(ts), int(np.argmax(ts)), 1e3 * sum(ts), 1e3 * (t3 - t2), 1e3 * (t4 - t3)), flush=True)
# what the end-of-block flush (the fold kernel of the last deferred step) costs a 20-step block
for variant in ("flush", "no flush (lower bound of folding the last step inside its own launch)", "flush", "no flush"):
    el = []
    for b in range(60):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(20):
            local.step(-1)
        if variant.startswith("flush"):
            local.flush()
        torch.cuda.synchronize()
        el.append(time.perf_counter() - t0)
        if local.tick % 479 > 440:
            local.reset_episode(snap, 479)
    print("%-70s 20-step block: median %.1f us = %.2f us per step" % (variant, 1e6 * np.median(el), 1e6 * np.median(el) / 20), flush=True)
