"""Feasibility probe: does splitting the 20 000 objects into independent parts, each stepped on its own HIP
stream, hide the kernel-boundary bubbles (one part computes while the other's launch drains)?  Raw ctypes
launches with pre-built parameter blocks (the host must not be the bottleneck)."""
import copy, ctypes as C, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.argv = ['bench.py']
import bench
from ssa_gym_amd import host, engine, _lib

def make(m, seed):
    pb = bench.build_problem(m, seed=seed)
    consts = host.make_consts(pb["Q"], pb["R"], 1e-4, 2.0, -3, 20.0, -np.pi / 2, pb["obs_lla"], obs_type='aer', propagator='fg')
    z = torch.zeros((1, 480, m, 3), dtype=torch.float64, device='cuda')
    eng = engine.HotPathEngine(consts, m, 1, pb["trans"], z, history=2)
    eng.load_state(0, pb["x_true"], pb["x"], np.broadcast_to(pb["P0"], (m, 6, 6)))
    eng.set_actions([3])
    blocks = []
    for par in (0, 1):          # even / odd step: slots and shard sets swap
        p = _lib.ssa_step_params()
        C.memmove(C.byref(p), C.byref(eng._p), C.sizeof(p))
        sin, sout = par, 1 - par
        p.x_true_in, p.x_true_out = eng._bx_t + sin * eng._sx, eng._bx_t + sout * eng._sx
        p.x_in, p.x_out = eng._bx + sin * eng._sx, eng._bx + sout * eng._sx
        p.P_in, p.P_out = eng._bP + sin * eng._sP, eng._bP + sout * eng._sP
        p.obs, p.metrics = eng._bo + sout * eng._so, eng._bm + sout * eng._sm
        p.upd, p.stats = eng._bu + sout * eng._su, eng._bs + sout * eng._ss
        p.time_offset = 1
        p.actions = eng.actions.data_ptr()
        p.launch_mask = _lib.LAUNCH_DEFER_FOLD
        p.stat_shards = eng._shard_sets[par].data_ptr()
        p.stat_shards_prev = eng._shard_sets[1 - par].data_ptr()
        p.stats_prev = eng._bs + sin * eng._ss
        p.aer_out = 0
        blocks.append(p)
    return eng, blocks

lib = _lib.load()
for parts in (1, 2, 4):
    m = 20000 // parts
    made = [make(m, 100 + i) for i in range(parts)]
    streams = [torch.cuda.Stream() for _ in range(parts)]
    handles = [s.cuda_stream for s in streams]
    refs = [(e._cref, [C.byref(b) for b in bl]) for e, bl in made]
    K = 240     # predict-only filters stay healthy for ~250 steps
    def run(k0, n):
        for k in range(k0, k0 + n):
            for (cref, prefs), h in zip(refs, handles):
                rc = lib.ssa_env_step_f64(cref, prefs[k & 1], h)
                assert rc == 0, rc
    run(0, 10)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(10, K)
    th = time.perf_counter() - t0
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    failed = sum(int((e.status != 0).sum().item()) for e, _ in made)
    print("parts=%d (x%d objects, one stream each): %.2f us per 20 000-object step = %.0f env-steps/s  (host enqueue %.2f us/step, failed filters %d)"
          % (parts, m, el / K * 1e6, K / el, th / K * 1e6, failed), flush=True)
