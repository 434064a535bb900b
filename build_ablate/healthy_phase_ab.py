"""Why is `hybrid` slower than `fg` while every filter is healthy?  Kernel time per step (event pairs) over the healthy part of an episode
(steps 1-200) and the late part (301-479) for propagator x covariance form, with the storage layout."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
sys.argv = ['bench.py']
import bench
from ssa_gym_amd import host, engine, parallel, _build
if os.environ.get('LIB'):
    _build.LIB = os.path.join(ROOT, os.environ['LIB'])
from ssa_gym_amd.catalogue import regime_order
m = 20000
pb = bench.build_problem(m, seed=100)
gen = torch.Generator(device="cuda").manual_seed(1)
z = torch.randn((1, 480, m, 3), dtype=torch.float64, device='cuda', generator=gen) * torch.as_tensor(pb["z_sigma"], device="cuda")
order = regime_order(pb["x_true"])
for rep in range(2):
    for prop in os.environ.get("PROPS", "fg,hybrid").split(","):
        for cov in os.environ.get("COVS", "centred,reference").split(","):
            consts = host.make_consts(pb["Q"], pb["R"], 1e-4, 2.0, -3, 20.0, -np.pi / 2, pb["obs_lla"], obs_type='aer', propagator=prop, covariance=cov)
            eng = engine.HotPathEngine(consts, m, 1, pb["trans"], z, history=2)
            eng.set_layout(order)
            eng.load_state(0, pb["x_true"], pb["x"], np.broadcast_to(pb["P0"], (m, 6, 6)))
            local = parallel.HipLocalStepper(eng, consts, fast_stats=True, defer_fold=True)
            local.load_schedule(np.arange(479) % m)
            for k in range(479):
                local.step(-1, profile_slot=k)
            local.flush(); torch.cuda.synchronize()
            ms = np.array([eng.profile_ms(k) for k in range(479)]) * 1e3
            print("%-7s %-9s  steps 11-200: %.2f us (min %.2f)   steps 301-479: %.2f   episode %.2f   failed %d" % (
                prop, cov, ms[10:200].mean(), ms[10:200].min(), ms[300:].mean(), ms.mean(), int((eng.status != 0).sum().item())), flush=True)
            del eng, local
