"""chained per-step launches (ssa_step_params.tile_flags): consecutive launches on two alternating streams, each wavefront waiting for ITS
tile of the previous step instead of the whole previous launch.  Whole episode at 20 000 objects: time per step against the plain
back-to-back launches, final state compared bit for bit.
Needs the experimental kernel of build_ablate/chain_experiment.patch (git apply; NOT part of the library: the experiment was negative,
profiles/r04_chained_launch_experiment.txt)."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
sys.argv = ['bench.py']
import bench
from ssa_gym_amd import host, engine, _lib, _build
if os.environ.get("LIB"):     # a diagnostic build instead of the shipped library
    _build.LIB = os.path.join(ROOT, os.environ["LIB"])
m = int(os.environ.get("M", "20000"))
prop = os.environ.get("PROP", "hybrid")
N = int(os.environ.get("STEPS", "479"))
MODE = os.environ.get("MODE", "two")      # 'one': the flags and fences on ONE stream (their cost alone: every wait is satisfied at once)
pb = bench.build_problem(m, seed=100)
consts = host.make_consts(pb["Q"], pb["R"], 1e-4, 2.0, -3, 20.0, -np.pi / 2, pb["obs_lla"], obs_type='aer', propagator=prop)
gen = torch.Generator(device="cuda").manual_seed(1)
z = torch.randn((1, 480, m, 3), dtype=torch.float64, device='cuda', generator=gen) * torch.as_tensor(pb["z_sigma"], device="cuda")
eng = engine.HotPathEngine(consts, m, 1, pb["trans"], z, history=2)
sched = torch.as_tensor((np.arange(480) % m).astype(np.int32)).cuda()
ntiles = (m + 3) // 4
S = torch.zeros((4, 1, _lib.STAT_SHARDS, _lib.STAT_SHARD_WORDS), dtype=torch.int64, device="cuda")
flags = torch.zeros(ntiles + 1, dtype=torch.int32, device="cuda")
streams = [torch.cuda.Stream(), torch.cuda.Stream()]


def episode(chained):
    eng.load_state(0, pb["x_true"], pb["x"], np.broadcast_to(pb["P0"], (m, 6, 6)))
    S.zero_(); flags.zero_()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(1, N + 1):
        st = streams[k % 2] if (chained and MODE == 'two') else streams[0]
        eng.launch_step((k - 1) % 2, k % 2, k, actions_ptr=sched.data_ptr() + 4 * (k - 1), stream=st.cuda_stream,
                        shards_out=S[k % 4].data_ptr(), shards_clear=S[(k + 2) % 4].data_ptr(),
                        chain=(flags.data_ptr(), k - 1, k) if chained else None)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / N
    s = N % 2
    out = (eng.x_true[s].cpu().numpy(), eng.x_filter[s].cpu().numpy(), eng.P_filter[s].cpu().numpy(), eng.status.cpu().numpy(),
           S[N % 4].cpu().numpy().copy())
    return dt, out, int(flags[ntiles].item())


episode(False)
for rep in range(3):
    dt0, ref, _ = episode(False)
    dt1, got, gave_up = episode(True)
    same = all(np.array_equal(a, b, equal_nan=True) for a, b in zip(ref, got))
    print("%s m=%d streams=%s lib=%s: plain %.2f us per step, chained %.2f us per step; bit-identical %s; gave up %d; failed %d" %
          (prop, m, MODE, os.environ.get('LIB', 'shipped'), dt0 * 1e6, dt1 * 1e6, same, gave_up, int((ref[3] != 0).sum())), flush=True)
