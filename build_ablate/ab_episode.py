"""A/B of two builds over whole episodes: bit-level digests of the filter state after every step, plus the step kernel's duration.

    LIB=<library, relative to the repo> PROP=hybrid OUT=gpurun_out/x.npz python build_ablate/ab_episode.py      # record
    LIB=<other build> PROP=hybrid REF=gpurun_out/x.npz python build_ablate/ab_episode.py                        # compare

One 20 000-object, 479-step round-robin episode of the bench workload (M, STEPS to change that).  After every step the int64 bit
patterns of x_filter, P_filter, x_true are summed (wrap-around) and the status words added: equal digests = bit-identical states
(a collision needs two differences that cancel to 64 bits).  Every launch is bracketed by its dispatch's event pair; the digest
kernels run between launches, so the durations are those of isolated launches (the bench's back-to-back figure is ~0.3 us lower)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
sys.argv = ['bench.py']
import bench
from ssa_gym_amd import host, engine, _lib, _build
if os.environ.get("LIB"):
    _build.LIB = os.path.join(ROOT, os.environ["LIB"])
    import ctypes
    _probe = ctypes.CDLL(_build.LIB)           # (an older build: bind what it has, accept its ABI number -- the parameter blocks only ever grew at the end)
    _lib.SIGNATURES = {k: v for k, v in _lib.SIGNATURES.items() if hasattr(_probe, k)}
    _lib.ABI_VERSION = _probe.ssa_abi_version()
m = int(os.environ.get("M", 20000))
steps = int(os.environ.get("STEPS", 479))
prop = os.environ.get("PROP", "hybrid")
pb = bench.build_problem(m, seed=int(os.environ.get("SEED", 100)))
consts = host.make_consts(pb["Q"], pb["R"], 1e-4, 2.0, -3, 20.0, -np.pi / 2, pb["obs_lla"], obs_type='aer', propagator=prop)
gen = torch.Generator(device="cuda").manual_seed(7)
zn = torch.randn((1, 480, m, 3), dtype=torch.float64, device="cuda", generator=gen) * torch.as_tensor(pb["z_sigma"], device="cuda")
eng = engine.HotPathEngine(consts, m, 1, pb["trans"], zn, history=2)
eng.load_state(0, pb["x_true"], pb["x"], np.broadcast_to(pb["P0"], (m, 6, 6)))
sched = (torch.arange(480, dtype=torch.int32, device="cuda") * 7919) % m
dig = torch.zeros((steps + 1, 4), dtype=torch.int64, device="cuda")
for i in range(1, steps + 1):
    eng.launch_step((i - 1) % 2, i % 2, i, actions_ptr=sched.data_ptr() + 4 * i, fast_stats=True, defer_fold=True, profile_slot=i % _lib.PROFILE_SLOTS)
    s = i % 2
    dig[i, 0] = eng.x_filter[s].view(torch.int64).sum()
    dig[i, 1] = eng.P_filter[s].view(torch.int64).sum()
    dig[i, 2] = eng.x_true[s].view(torch.int64).sum()
    dig[i, 3] = eng.status.sum()
eng.flush_stats()
torch.cuda.synchronize()
ms = np.array([0.0] + [eng.profile_ms(i % _lib.PROFILE_SLOTS) for i in range(1, steps + 1)]) * 1e3
d = dig.cpu().numpy()
failed = int((eng.status != 0).sum().item())
print("%s  %s  m=%d: kernel us per 60-step window:" % (os.environ.get("LIB", "in-tree"), prop, m),
      " ".join("%.2f" % ms[lo:lo + 60].mean() for lo in range(1, steps + 1, 60)), " episode mean %.2f  step 400: %.2f  failed %d" % (
          ms[1:].mean(), ms[min(400, steps)], failed))
if os.environ.get("REF"):
    ref = np.load(os.environ["REF"])
    rd = ref["dig"]
    n = min(len(rd), len(d))
    bad = np.where((rd[:n] != d[:n]).any(axis=1))[0]
    if len(bad):
        print("A/B: states DIFFER from step %d on (%d of %d steps differ; columns x, P, x_true, status: %s)" % (bad[0], len(bad), n - 1, (rd[bad[0]] != d[bad[0]]).tolist()))
    else:
        print("A/B: bit-identical over %d steps (x_filter, P_filter, x_true, status digests); failed filters %d vs %d; kernel %.2f vs %.2f us" % (
            n - 1, failed, int(ref["failed"]), ms[1:].mean(), float(ref["ms"][1:].mean())))
if os.environ.get("OUT"):
    np.savez(os.environ["OUT"], dig=d, ms=ms, failed=failed)
