"""PMC workload for the truncation builds (build_ablate/build_trunc.sh): 50 launches of the one-tile FG step kernel, every one from the
SAME healthy input state (slot 0 -> slot 1; a truncated wave writes nothing, so the state must not ping-pong)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.argv = ['bench.py']
import bench
from ssa_gym_amd import _lib, _build, host, engine
if os.environ.get('LIB'):
    _build.LIB = os.environ['LIB']
m = int(os.environ.get('M', '20000'))
pb = bench.build_problem(m, seed=100)
consts = host.make_consts(pb["Q"], pb["R"], 1e-4, 2.0, -3, 20.0, -np.pi / 2, pb["obs_lla"], obs_type='aer', propagator=os.environ.get('PROP', 'fg'))
z = torch.zeros((1, 480, m, 3), dtype=torch.float64, device='cuda')
eng = engine.HotPathEngine(consts, m, 1, pb["trans"], z, history=2)
STATE = os.environ.get('STATE')          # npz written by GEN=<steps> (a full-library run): a late-episode state
GEN = int(os.environ.get('GEN', '0'))
if STATE and not GEN:
    st_ = np.load(STATE)
    xt0, x0, P0 = st_["x_true"], st_["x"], st_["P"]
else:
    xt0, x0, P0 = pb["x_true"], pb["x"], np.broadcast_to(pb["P0"], (m, 6, 6))
eng.load_state(0, xt0, x0, P0)
if GEN:
    for k in range(GEN):
        eng.set_actions([-1])
        eng.launch_step(k % 2, (k + 1) % 2, k + 1)
    torch.cuda.synchronize()
    sl = GEN % 2
    np.savez(STATE, x_true=eng.x_true[sl].cpu().numpy(), x=eng.x_filter[sl].cpu().numpy(), P=eng.P_filter[sl].cpu().numpy())
    print("failed after", GEN, "steps:", int((eng.status != 0).sum().item()))
    sys.exit(0)
act = int(os.environ.get('ACTION', '-1'))
TICK0 = GEN0 = int(os.environ.get('TICK0', '0'))
if os.environ.get('FAST', '1') == '1':     # the bench's path: statistics by sharded atomics + the aer block in the kernel's epilogue
    from ssa_gym_amd import parallel
    eng.load_state(1, xt0, x0, P0)   # (a truncated wave writes nothing: both slots stay healthy)
    local = parallel.HipLocalStepper(eng, consts, fast_stats=True)
    for k in range(50):
        local.tick = TICK0 + (k & 1)       # the same time index (and slot pair) for every launch
        local.step(act)
else:
    for k in range(50):
        eng.set_actions([act])
        eng.launch_step(0, 1, 1)
torch.cuda.synchronize()
