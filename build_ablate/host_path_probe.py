"""what the host side of env.step() costs at 20 000 objects: launch, synchronisation (stream sync vs spinning on a host-mapped word
the GPU writes), the observation's way to the host (copy engine into pinned memory vs the kernel writing host-mapped memory)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
sys.argv = ['bench.py']
import bench
from ssa_gym_amd import _lib, engine, host
m = 20000
pb = bench.build_problem(m, seed=100)
consts = host.make_consts(pb["Q"], pb["R"], 1e-4, 2.0, -3, 20.0, -np.pi / 2, pb["obs_lla"], obs_type='aer', propagator='fg')
zn = torch.zeros((1, 480, m, 3), dtype=torch.float64, device='cuda')
eng = engine.HotPathEngine(consts, m, 1, pb["trans"], zn, history=2)
eng.load_state(0, pb["x_true"], pb["x"], np.broadcast_to(pb["P0"], (m, 6, 6)))
act = torch.zeros(1, dtype=torch.int32).pin_memory()
stats_h = torch.zeros(8, dtype=torch.float64).pin_memory()
upd_h = torch.zeros(64, dtype=torch.float64).pin_memory()
aer_d = torch.zeros(m * 4, dtype=torch.float64, device='cuda')
aer_h = torch.zeros(m * 4, dtype=torch.float64).pin_memory()
obs_h = torch.zeros(m * 12, dtype=torch.float64).pin_memory()
seq_h = torch.zeros(16, dtype=torch.int64).pin_memory()
seq_np = seq_h.numpy()
one = torch.ones(1, dtype=torch.int64, device='cuda')
cur = torch.cuda.current_stream()
N = 300
tick = [0]


def launch(aer_ptr=0, stats=True):
    tick[0] += 1
    t = tick[0]
    eng.launch_step((t - 1) % 2, t % 2, (t % 470) + 1, actions_ptr=act.data_ptr(), aer_out=aer_ptr, stats_out=stats_h.data_ptr() if stats else 0,
                    upd_out=upd_h.data_ptr(), stream=cur.cuda_stream, fast_stats=True)
    return t


def timeit(name, fn):
    for _ in range(30):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(N):
        fn()
    torch.cuda.synchronize()
    print("%-72s %7.1f us" % (name, (time.perf_counter() - t0) / N * 1e6), flush=True)


def f_launch_only():
    launch()
timeit("launch only (no per-step sync)", f_launch_only)


def f_launch_sync():
    launch()
    cur.synchronize()
timeit("launch + stream.synchronize()", f_launch_sync)

seqdev = torch.zeros(1, dtype=torch.int64, device='cuda')


def f_launch_spin():
    t = launch()
    seqdev.fill_(t)
    seq_h[:1].copy_(seqdev, non_blocking=True)
    while seq_np[0] != t:
        pass
timeit("launch + fill + 8-byte D2H + spin on the pinned word", f_launch_spin)


def f_aer_copy_sync():
    launch(aer_d.data_ptr())
    aer_h.copy_(aer_d, non_blocking=True)
    cur.synchronize()
timeit("'aer': launch + 0.64 MB D2H (copy engine, pinned) + synchronize", f_aer_copy_sync)


def f_aer_mapped_sync():
    launch(aer_h.data_ptr())
    cur.synchronize()
timeit("'aer': kernel writes the 0.64 MB block into host-mapped memory + synchronize", f_aer_mapped_sync)


def f_flat_copy_sync():
    t = launch()
    obs_h.copy_(eng.obs[t % 2].reshape(-1), non_blocking=True)
    cur.synchronize()
timeit("'flatten': launch + 1.92 MB D2H (copy engine, pinned) + synchronize", f_flat_copy_sync)


def f_flat_pageable():
    t = launch()
    eng.obs[t % 2].reshape(-1).cpu()
timeit("'flatten': launch + 1.92 MB .cpu() (pageable, fresh tensor)", f_flat_pageable)

side = torch.cuda.Stream()
ev = torch.cuda.Event()


def f_flat_two_halves():
    t = launch()
    src = eng.obs[t % 2].reshape(-1)
    h = m * 6
    ev.record(cur)
    side.wait_event(ev)
    with torch.cuda.stream(side):
        obs_h[h:].copy_(src[h:], non_blocking=True)
    obs_h[:h].copy_(src[:h], non_blocking=True)
    cur.synchronize()
    side.synchronize()
timeit("'flatten': the copy split over two streams (two copy engines)", f_flat_two_halves)
