"""host cost of the sharded step at 1 rank (RCCL, torch.distributed.run): enqueue vs completion per mode, and per call (diagnostic)."""
import os, sys, time
import numpy as np, torch, torch.distributed as dist
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
sys.argv = ['bench.py']
import bench
from ssa_gym_amd import host, engine, parallel
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=int(os.environ.get("RANK", 0)), world_size=int(os.environ.get("WORLD_SIZE", 1)))
m = 20000
pb = bench.build_problem(m, seed=100)
consts = host.make_consts(pb["Q"], pb["R"], 1e-4, 2.0, -3, 20.0, -np.pi / 2, pb["obs_lla"], obs_type='aer', propagator='fg')
z = torch.zeros((1, 480, m, 3), dtype=torch.float64, device='cuda')
eng = engine.HotPathEngine(consts, m, 1, pb["trans"], z, history=2)
eng.load_state(0, pb["x_true"], pb["x"], np.broadcast_to(pb["P0"], (m, 6, 6)))
local = parallel.HipLocalStepper(eng, consts, fast_stats=True, defer_fold=True)
local.load_schedule(np.arange(4000) % m)
plan = parallel.ShardPlan(m, 1, 0)
sh = parallel.ShardedStepper(plan, local, obs_cols=1)
snap = eng.snapshot(0)
for mode in (False, True):
    for rep in range(3):
        local.reset_episode(snap, 480); sh.wait(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(400):
            sh.step(k % m, overlap=mode)
        t1 = time.perf_counter()
        local.flush(); sh.wait(); torch.cuda.synchronize()
        t2 = time.perf_counter()
        print("overlap", mode, "enqueue us/step %.2f  completion us/step %.2f" % ((t1 - t0) / 400 * 1e6, (t2 - t0) / 400 * 1e6), flush=True)
# where the overlapped step's host time goes
import collections
acc = collections.defaultdict(float)
def T(name, fn):
    t0 = time.perf_counter(); r = fn(); acc[name] += time.perf_counter() - t0; return r
local.reset_episode(snap, 480); sh.wait(); torch.cuda.synchronize()
N = 400
for k in range(N):
    b = sh.k & 1
    cur = T("current_stream", lambda: torch.cuda.current_stream())
    if sh._pending[b]:
        T("wait done", lambda: cur.wait_event(sh._done[b])); sh._pending[b] = False
    T("local.step (folded)", lambda: local.step(-1, sh._v_obs[b], sh._v_stats[b], obs_cols=1, stream=cur.cuda_stream))
    T("ready.record", lambda: sh._ready[b].record(cur))
    T("comm.wait_event", lambda: sh.comm.wait_event(sh._ready[b]))
    T("allgather(comm)", lambda: sh._all_gather(sh.recv[b], sh.send[b], sh.comm))
    T("done.record", lambda: sh._done[b].record(sh.comm))
    sh._pending[b] = True; sh.k += 1
sh.wait(); torch.cuda.synchronize()
print("overlapped step, host us per call:", {k: round(v / N * 1e6, 2) for k, v in acc.items()}, "sum %.2f" % (sum(acc.values()) / N * 1e6))
# per-call host cost
send, recv = sh.send[0], sh.recv[0]
cur = torch.cuda.current_stream()
def cost(fn, n=2000):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    t1 = time.perf_counter(); torch.cuda.synchronize()
    return (t1 - t0) / n * 1e6
print("ncclAllGather enqueue us: %.2f" % cost(lambda: sh._rccl.all_gather_f64(send.data_ptr(), recv.data_ptr(), sh.width, cur.cuda_stream)))
ev = torch.cuda.Event()
print("event.record us: %.2f" % cost(lambda: ev.record(cur)))
print("stream.wait_event us: %.2f" % cost(lambda: sh.comm.wait_event(ev)))
def ctx():
    with torch.cuda.stream(sh.comm):
        pass
print("with torch.cuda.stream us: %.2f" % cost(ctx))
print("torch.cuda.current_stream() us: %.2f" % cost(lambda: torch.cuda.current_stream()))
print("local.step us: %.2f" % cost(lambda: local.step(-1), 400))
sh.close(); dist.destroy_process_group()
