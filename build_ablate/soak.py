"""soak + determinism: N whole episodes of the 20 000-object env in closed loop (device agent) and with a round-robin schedule,
twice from the same seed: the two runs must agree bit for bit (state, covariance, statistics); reports failures and rate."""
import hashlib, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
sys.argv = ['bench.py']
import bench
from ssa_gym_amd import host, engine, _lib, _build
if os.environ.get("LIB"):     # a diagnostic build instead of the shipped library
    _build.LIB = os.path.join(ROOT, os.environ["LIB"])
m, EP = 20000, int(os.environ.get("EPISODES", "20"))
AGENT = getattr(_lib, "AGENT_" + os.environ.get("AGENT", "VISIBLE_GREEDY"))      # the closed loops' device agent
pb = bench.build_problem(m, seed=100)
consts = host.make_consts(pb["Q"], pb["R"], 1e-4, 2.0, -3, 20.0, np.radians(10.0), pb["obs_lla"], obs_type='aer', propagator=os.environ.get('PROP', 'fg'))
def run_persistent():
    """the closed loop through ssa_env_closed_loop_f64: one launch per 479-step episode; must hash like the per-step closed loop"""
    gen = torch.Generator(device="cuda").manual_seed(7)
    zn = torch.randn((1, 480, m, 3), dtype=torch.float64, device="cuda", generator=gen) * torch.as_tensor(pb["z_sigma"], device="cuda")
    eng = engine.HotPathEngine(consts, m, 1, pb["trans"], zn, history=2)
    if os.environ.get("LAYOUT") == "1":      # the engine's storage layout (the hashes are taken in the caller's order all the same)
        from ssa_gym_amd.catalogue import regime_order
        eng.set_layout(regime_order(pb["x_true"]))
    eng.load_state(0, pb["x_true"], pb["x"], np.broadcast_to(pb["P0"], (m, 6, 6)))
    snap = eng.snapshot(0)
    fb = torch.zeros(481, dtype=torch.int32, device="cuda")
    log = torch.zeros(480, dtype=torch.int32, device="cuda")
    stats = torch.zeros((479, 8), dtype=torch.float64, device="cuda")
    h = hashlib.sha256(); tick = 0; fails = []
    t0 = time.perf_counter()
    for ep in range(EP):
        tick += (-tick) % 480
        eng.restore(tick % 2, snap)
        eng.launch_agent_select(tick, tick, AGENT, log.data_ptr(), fallback_ptr=fb.data_ptr())
        assert eng.launch_closed_loop(tick % 2, tick + 1, AGENT, log, stats, fallback=fb[:480])
        tick += 479
        torch.cuda.synchronize()
        assert int(eng.loop_error[0]) == 0
        s = tick % 2
        eng.stats[s, 0].copy_(stats[478])
        for tns in (eng.caller_rows(eng.x_true[s]), eng.caller_rows(eng.x_filter[s]), eng.caller_rows(eng.P_filter[s]), eng.stats[s], eng.caller_rows(eng.status)):
            h.update(tns.cpu().numpy().tobytes())
        fails.append(int((eng.status != 0).sum().item()))
    return h.hexdigest(), fails, time.perf_counter() - t0


def run(closed):
    gen = torch.Generator(device="cuda").manual_seed(7)
    zn = torch.randn((1, 480, m, 3), dtype=torch.float64, device="cuda", generator=gen) * torch.as_tensor(pb["z_sigma"], device="cuda")
    eng = engine.HotPathEngine(consts, m, 1, pb["trans"], zn, history=2)
    if os.environ.get("LAYOUT") == "1":      # the engine's storage layout (the hashes are taken in the caller's order all the same)
        from ssa_gym_amd.catalogue import regime_order
        eng.set_layout(regime_order(pb["x_true"]))
    eng.load_state(0, pb["x_true"], pb["x"], np.broadcast_to(pb["P0"], (m, 6, 6)))
    snap = eng.snapshot(0)
    word = torch.zeros(1, dtype=torch.int32, device="cuda"); fb = torch.zeros(1, dtype=torch.int32, device="cuda")
    sched = (torch.arange(480 * EP, dtype=torch.int32, device="cuda") * 7919) % m
    h = hashlib.sha256(); tick = 0; fails = []; stats_sum = np.zeros(8)
    t0 = time.perf_counter()
    for ep in range(EP):
        tick += (-tick) % 480
        eng.flush_stats(); eng.restore(tick % 2, snap)
        if closed:
            eng.launch_agent_select(tick, tick, AGENT, word.data_ptr(), fallback_ptr=fb.data_ptr())
        for i in range(1, 480):
            tick += 1
            ap = word.data_ptr() if closed else sched.data_ptr() + 4 * (ep * 480 + i)
            eng.launch_step((tick - 1) % 2, tick % 2, tick, actions_ptr=ap, fast_stats=True, defer_fold=True)
            if closed:
                eng.launch_agent_select(tick, tick, AGENT, word.data_ptr(), fallback_ptr=fb.data_ptr())
        eng.flush_stats(); torch.cuda.synchronize()
        s = tick % 2
        for tns in (eng.caller_rows(eng.x_true[s]), eng.caller_rows(eng.x_filter[s]), eng.caller_rows(eng.P_filter[s]), eng.stats[s], eng.caller_rows(eng.status)):
            h.update(tns.cpu().numpy().tobytes())
        fails.append(int((eng.status != 0).sum().item()))
    dt = time.perf_counter() - t0
    return h.hexdigest(), fails, dt
for closed in (False, True):
    a = run(closed); b = run(closed)
    print("%s: %d episodes x 479 steps, %.1f s (%.0f env-steps/s incl. resets and per-episode read-back); failed filters at episode ends: min %d max %d; "
          "run 1 == run 2 bit for bit: %s  (sha256 %s)" % ("closed loop (%s, 10 deg mask)" % os.environ.get("AGENT", "VISIBLE_GREEDY").lower() if closed else "round-robin schedule",
          EP, a[2], EP * 479 / a[2], min(a[1]), max(a[1]), a[0] == b[0], a[0][:16]), flush=True)
    assert a[0] == b[0]
a = run_persistent(); b = run_persistent(); c = run(True)
print("closed loop in one persistent launch per episode: %d episodes x 479 steps, %.1f s (%.0f env-steps/s incl. resets and read-back); failed filters at "
      "episode ends: min %d max %d; run 1 == run 2 bit for bit: %s; == the per-step closed loop bit for bit: %s  (sha256 %s)"
      % (EP, a[2], EP * 479 / a[2], min(a[1]), max(a[1]), a[0] == b[0], a[0] == c[0], a[0][:16]), flush=True)
assert a[0] == b[0] == c[0]
