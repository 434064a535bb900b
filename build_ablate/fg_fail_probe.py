"""which fg filters fail in the soak's closed-loop episode, when, with what status -- and the state one step earlier (diagnostic)"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
sys.argv = ['bench.py']
import bench
from ssa_gym_amd import host, engine, _lib, _build
if os.environ.get("LIB"):
    _build.LIB = os.path.join(ROOT, os.environ["LIB"])
m = 20000
AGENT = _lib.AGENT_VISIBLE_GREEDY
pb = bench.build_problem(m, seed=100)
consts = host.make_consts(pb["Q"], pb["R"], 1e-4, 2.0, -3, 20.0, np.radians(10.0), pb["obs_lla"], obs_type='aer', propagator='fg')
gen = torch.Generator(device="cuda").manual_seed(7)
zn = torch.randn((1, 480, m, 3), dtype=torch.float64, device="cuda", generator=gen) * torch.as_tensor(pb["z_sigma"], device="cuda")
eng = engine.HotPathEngine(consts, m, 1, pb["trans"], zn, history=2)
eng.load_state(0, pb["x_true"], pb["x"], np.broadcast_to(pb["P0"], (m, 6, 6)))
word = torch.zeros(1, dtype=torch.int32, device="cuda"); fb = torch.zeros(1, dtype=torch.int32, device="cuda")
eng.launch_agent_select(0, 0, AGENT, word.data_ptr(), fallback_ptr=fb.data_ptr())
acts = []
prev = None
for t in range(1, 480):
    eng.launch_step((t - 1) % 2, t % 2, t, actions_ptr=word.data_ptr(), fast_stats=True, defer_fold=True)
    torch.cuda.synchronize()
    nf = int(eng.fail_count.item())
    if prev is not None and nf > prev[0]:
        for r in eng.fail_log[prev[0]:nf]:
            j = int(r[_lib.FAIL_OBJ])
            s_in = (t - 1) % 2
            x = eng.x_filter[s_in, j].cpu().numpy(); P = eng.P_filter[s_in, j].cpu().numpy(); xt = eng.x_true[s_in, j].cpu().numpy()
            rr = np.linalg.norm(x[:3]); vv = x[3:] @ x[3:]
            alpha = 2 / rr - vv / 398600441800000.0
            w = np.linalg.eigvalsh(0.5 * (P + P.T))
            print("step %d: object %d fails with status %d; selected %s; state before: r %.3e m, alpha %.3e (a = %.3e m), sd pos %.3e, eig(P) min %.3e max %.3e, "
                  "err %s" % (t, j, int(r[_lib.FAIL_STATUS]), acts[-3:], rr, alpha, 1 / alpha, np.sqrt(np.trace(P[:3, :3])), w.min(), w.max(), r[_lib.FAIL_ERR:_lib.FAIL_ERR + 4]))
    prev = (nf,)
    acts.append(int(word.item()))
    eng.launch_agent_select(t, t, AGENT, word.data_ptr(), fallback_ptr=fb.data_ptr())
torch.cuda.synchronize()
print("lib %s: failed at the end %d; distinct actions %d" % (os.environ.get("LIB", "shipped"), int((eng.status != 0).sum().item()), len(set(acts))))
