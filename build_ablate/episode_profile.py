"""step-kernel duration and filter health over one 479-step episode of the bench workload (20 000 objects)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.argv = ['bench.py']
import bench
from ssa_gym_amd import host, engine, parallel, _build, _lib
if os.environ.get("LIB"):     # a diagnostic build instead of the shipped library
    _build.LIB = os.path.join(ROOT, os.environ["LIB"])
    _lib._lib = None
    _lib.load()
m = 20000
pb = bench.build_problem(m, seed=100)
consts = host.make_consts(pb["Q"], pb["R"], 1e-4, 2.0, -3, 20.0, -np.pi / 2, pb["obs_lla"], obs_type='aer',
                          propagator=os.environ.get("PROP", "fg"))
gen = torch.Generator(device="cuda").manual_seed(1)
z = torch.randn((1, 480, m, 3), dtype=torch.float64, device='cuda', generator=gen) * torch.as_tensor(pb["z_sigma"], device="cuda")
eng = engine.HotPathEngine(consts, m, 1, pb["trans"], z, history=2)
eng.load_state(0, pb["x_true"], pb["x"], np.broadcast_to(pb["P0"], (m, 6, 6)))
local = parallel.HipLocalStepper(eng, consts, fast_stats=True, defer_fold=True)
local.load_schedule(np.arange(479) % m)
mu = 398600441800000.0
rows = []
for k in range(479):
    local.step(-1, profile_slot=k)
    if k % 60 == 59 or k == 478:
        local.flush(); torch.cuda.synchronize()
        x = eng.x_filter[local.tick % 2].cpu().numpy()
        r = np.linalg.norm(x[:, :3], axis=1); v2 = np.sum(x[:, 3:] ** 2, axis=1)
        alpha = 2.0 / r - v2 / mu
        h = np.cross(x[:, :3], x[:, 3:]); ecc = np.sqrt(np.maximum(0, 1 - np.sum(h * h, 1) * alpha / mu))
        w = np.linalg.eigvalsh(eng.P_filter[local.tick % 2].cpu().numpy())
        rows.append((k + 1, float(np.mean(alpha <= 0)), float(np.mean((ecc >= 0.99) & (ecc <= 1.01))), float(np.mean(w[:, 0] <= 0)),
                     int((eng.status != 0).sum().item())))
local.flush(); torch.cuda.synchronize()
ms = np.array([eng.profile_ms(k) for k in range(479)]) * 1e3
for lo in range(0, 479, 60):
    print("steps %3d-%3d: kernel %.2f us (min %.2f max %.2f)" % (lo + 1, min(lo + 60, 479), ms[lo:lo + 60].mean(), ms[lo:lo + 60].min(), ms[lo:lo + 60].max()))
print("episode mean %.2f us" % ms.mean())
for r_ in rows:
    print("after step %3d: hyperbolic means %.3f  near-parabolic %.3f  non-PD covariances %.3f  failed %d" % r_)
