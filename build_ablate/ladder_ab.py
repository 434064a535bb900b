"""A/B of two builds over one round-robin episode (20 000 objects): first step / object where the filter states differ, and that
object's prior covariance (the matrix its ladder saw).  LIB = build, OUT = npz; second run with REF=<npz of the other build>."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
sys.argv = ['bench.py']
import bench
from ssa_gym_amd import host, engine, _lib, _build
_build.LIB = os.path.join(ROOT, os.environ["LIB"])
import ctypes
_probe = ctypes.CDLL(_build.LIB)           # (an older build: bind what it has, accept its ABI number -- the step's parameter block
_lib.SIGNATURES = {k: v for k, v in _lib.SIGNATURES.items() if hasattr(_probe, k)}      # only ever grew at the end)
_lib.ABI_VERSION = _probe.ssa_abi_version()
m = 20000
pb = bench.build_problem(m, seed=100)
consts = host.make_consts(pb["Q"], pb["R"], 1e-4, 2.0, -3, 20.0, np.radians(10.0), pb["obs_lla"], obs_type='aer', propagator='fg')
gen = torch.Generator(device="cuda").manual_seed(7)
zn = torch.randn((1, 480, m, 3), dtype=torch.float64, device="cuda", generator=gen) * torch.as_tensor(pb["z_sigma"], device="cuda")
eng = engine.HotPathEngine(consts, m, 1, pb["trans"], zn, history=2)
eng.load_state(0, pb["x_true"], pb["x"], np.broadcast_to(pb["P0"], (m, 6, 6)))
sched = (torch.arange(480, dtype=torch.int32, device="cuda") * 7919) % m
ref = np.load(os.environ["REF"]) if os.environ.get("REF") else None
sig = np.zeros((480, m)); st = np.zeros((480, m), dtype=np.int8)
for i in range(1, 480):
    Pprev = eng.P_filter[(i - 1) % 2].clone() if ref is not None else None
    xprev = eng.x_filter[(i - 1) % 2].clone() if ref is not None else None
    eng.launch_step((i - 1) % 2, i % 2, i, actions_ptr=sched.data_ptr() + 4 * i, fast_stats=True, defer_fold=True)
    torch.cuda.synchronize()
    sig[i] = eng.P_filter[i % 2].sum(dim=(1, 2)).cpu().numpy()
    st[i] = eng.status.cpu().numpy()
    if ref is not None:
        bad = np.where((sig[i] != ref["sig"][i]) & ~(np.isnan(sig[i]) & np.isnan(ref["sig"][i])))[0]
        if len(bad):
            j = int(bad[0])
            print("first difference at step %d, object %d (%d objects differ); status here %d, other build %d" % (i, j, len(bad), st[i, j], ref["st"][i, j]))
            t0 = 4 * (j // 4)          # the whole tile: the four objects one wavefront factorises side by side
            np.savez(os.environ["OUT"], P=Pprev[j].cpu().numpy(), x=xprev[j].cpu().numpy(), step=i, obj=j,
                     P_tile=Pprev[t0:t0 + 4].cpu().numpy(), x_tile=xprev[t0:t0 + 4].cpu().numpy(), status_tile=st[i - 1, t0:t0 + 4])
            from ssa_gym_amd import device
            _, _, scale = host.merwe_weights(1e-4, 2.0, -3)
            for lib_name in ("this build",):
                rung, mask, _ = device.ladder_probe(Pprev[t0:t0 + 4].contiguous(), scale)
                print("ladder probe of the tile (%s): rung %s  masks %s" % (lib_name, rung.cpu().numpy().tolist(), [format(int(m) & 0x1ffff, '017b')[::-1] for m in mask.cpu().numpy()]))
            np.set_printoptions(precision=17, linewidth=200)
            print("prior covariance of that object:\n", Pprev[j].cpu().numpy())
            break
else:
    print("no difference" if ref is not None else "recorded; failed at the end: %d" % int((st[479] != 0).sum()))
if ref is None:
    np.savez(os.environ["OUT"], sig=sig, st=st)
