"""storage layout on / off over a WHOLE 20 000-object episode (HotPathEngine.set_layout(catalogue.regime_order)): kernel time per step and --
the point -- the final state, status words and failure records compared bit for bit in the caller's order: an object's arithmetic must
not depend on who shares its wavefront."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
sys.argv = ['bench.py']
import bench
from ssa_gym_amd import host, engine, parallel, _lib
from ssa_gym_amd.catalogue import regime_order
m = int(os.environ.get("M", "20000"))
prop = os.environ.get("PROP", "hybrid")
pb = bench.build_problem(m, seed=100)
consts = host.make_consts(pb["Q"], pb["R"], 1e-4, 2.0, -3, 20.0, -np.pi / 2, pb["obs_lla"], obs_type='aer', propagator=prop)
gen = torch.Generator(device="cuda").manual_seed(1)
z = torch.randn((1, 480, m, 3), dtype=torch.float64, device='cuda', generator=gen) * torch.as_tensor(pb["z_sigma"], device="cuda")


def episode(order):
    eng = engine.HotPathEngine(consts, m, 1, pb["trans"], z, history=2)
    eng.set_layout(order)
    eng.load_state(0, pb["x_true"], pb["x"], np.broadcast_to(pb["P0"], (m, 6, 6)))
    local = parallel.HipLocalStepper(eng, consts, fast_stats=True, defer_fold=True)
    local.load_schedule(np.arange(479) % m)
    for k in range(479):
        local.step(-1, profile_slot=k)
    local.flush(); torch.cuda.synchronize()
    ms = np.array([eng.profile_ms(k) for k in range(479)]) * 1e3
    nf = int(eng.fail_count.item())
    fails = sorted((int(r[_lib.FAIL_OBJ]), int(r[_lib.FAIL_STATUS]), int(r[_lib.FAIL_TIME])) for r in eng.fail_log[:nf])
    eng.to_caller_order()
    s = local.tick % 2
    return ms, fails, (eng.x_true[s].cpu().numpy(), eng.x_filter[s].cpu().numpy(), eng.P_filter[s].cpu().numpy(), eng.status.cpu().numpy())


a = episode(None)
b = episode(regime_order(pb["x_true"]))
c = episode(np.random.RandomState(0).permutation(m))
for name, r in (("caller's order", a), ("regime layout", b), ("random layout", c)):
    print("%s m=%d  %-15s kernel %.2f us per step over the episode (steps 301-360: %.2f); failed filters %d" %
          (prop, m, name, r[0].mean(), r[0][300:360].mean(), len(r[1])))
for name, r in (("regime layout", b), ("random layout", c)):
    same = all(np.array_equal(u, v, equal_nan=True) for u, v in zip(a[2], r[2]))
    print("%s vs the caller's order: final x_true / x_filter / P_filter / status bit-identical: %s; failure records identical: %s" % (name, same, a[1] == r[1]))
