"""Episode-level workload shared by tests/test_episode_failures.py (TEST INFRASTRUCTURE).

The reference's filters FAIL late in a predict-mostly episode (ssa_tasker_simple_2.py:271-285: predict() raising
LinAlgError from an exhausted robust_cholesky ladder, dynamics.py:402-417, or returning NaN, farnocchia.py:337-353);
the first failure puts 1e20 into delta_pos and ends a 'jones' / 'shaped' episode (:325-343).  This module builds ONE
deterministic episode -- 2 000 objects drawn from the golden catalogue subset, env defaults (alpha 1e-4, dt 20 s,
obs_limit -90 deg: an update every step), 479 round-robin steps -- and runs it on the CPU oracle; the GPU test runs the
same inputs through ssa_env_step_f64 and compares failure statistics.
"""
import numpy as np

import oracle as orc  # noqa: F401
from conftest import golden

WINDOWS = (60, 120, 180, 240, 300, 360, 420, 479)
N_STEPS = 479
SEEDS = (7, 8, 9, 10, 11)      # five workloads: ~270 failures pooled -> a three-sigma band of +-6 % on the count


def workload(m=2000, seed=7, n=480):
    rs = np.random.RandomState(seed)
    cat = golden("catalogue_subset.npy")
    g = golden("ukf_step_golden.npz")
    xt = cat[rs.randint(0, len(cat), m)]
    x = xt + rs.normal(size=(m, 6)) * np.array([1e5] * 3 + [1e2] * 3)      # envs/__init__.py:25 x_sigma
    P = np.tile(g["P0"], (m, 1, 1))
    zn = np.random.RandomState(1).normal(size=(n, m, 3)) * np.array([np.pi / 648000] * 2 + [1e3])
    return dict(x_true=xt, x=x, P=P, g=g, z_noise=zn, c2t=golden("c2t_2020-05-04_dt20_n480.npy"), m=m)


MU = 398600441800000.0
DIVERGED_STEP = 300


def diverged_ids(x):
    """objects whose filter mean has left the strong-elliptic regime (ecc >= 0.99 or unbound): the population the reference's failures
    come from (a diverged prior + the cancellation in the covariance sum)"""
    r = np.linalg.norm(x[:, :3], axis=1)
    v2 = np.sum(x[:, 3:] ** 2, axis=1)
    alpha = 2.0 / r - v2 / MU
    h = np.cross(x[:, :3], x[:, 3:])
    ecc = np.sqrt(np.maximum(0.0, 1.0 - np.sum(h * h, axis=1) * alpha / MU))
    bad = (alpha <= 0) | (ecc >= 0.99) | ~np.isfinite(ecc)
    return [int(j) for j in np.where(bad)[0]]


def jaccard(a, b):
    """overlap of two failed-filter sets: |a & b| / |a | b| (1 for two empty sets)"""
    a, b = set(int(v) for v in a), set(int(v) for v in b)
    return 1.0 if not (a or b) else len(a & b) / float(len(a | b))


def first_failure_steps(status_hist):
    """status_hist[k][j] = status of object j after step k + 1 -> {object: first step with status != 0}"""
    st = np.asarray(status_hist) != 0
    failed = np.where(st[-1])[0]
    return {int(j): int(np.argmax(st[:, j])) + 1 for j in failed}


def summarise(n_failed, status_end, max_dpos, first_fail=None):
    """n_failed[k], max_dpos[k] for step k + 1 (k = 0 .. 478); status_end int32[m]; first_fail {object: step} (WHICH filters failed, and when)"""
    n_failed, max_dpos = np.asarray(n_failed), np.asarray(max_dpos)
    jones = None
    for k in range(len(max_dpos)):       # 'jones' termination (ssa_tasker_simple_2.py:325-335); NaN compares false as in numpy
        if max_dpos[k] > 5e6 or max_dpos[k] < 3e4:
            jones = k + 1
            break
    first_any = int(np.argmax(n_failed > 0)) + 1 if (n_failed > 0).any() else None
    out = dict(failed_at={int(w): int(n_failed[w - 1]) for w in WINDOWS},
               status_mix=np.bincount(status_end, minlength=5)[:5].tolist(),
               jones_done_step=jones, first_failure_step=first_any)
    if first_fail is not None:
        ids = sorted(first_fail)
        out["failed_ids"] = ids
        out["failed_first_step"] = [first_fail[j] for j in ids]
    return out


def run_oracle(w, centred=False, resample=False, threads=8):
    """the reference-order CPU restatement (oracle/ssa_oracle.c, OpenMP build: bit-identical to the serial one)"""
    o = orc.Oracle(omp=True)
    o.lib.orc_omp_threads(int(threads))
    g, m = w["g"], w["m"]
    Wm, Wc, scale = orc.merwe_weights(1e-4, 2.0, -3)
    xt, x, P = w["x_true"], w["x"], w["P"]
    st = np.zeros(m, dtype=np.int32)
    nf, mx, hist = [], [], []
    for i in range(1, N_STEPS + 1):
        a = (i - 1) % m
        r = o.env_step(xt, x, P, st, 20.0, g["Q"], g["R"], Wm, Wc, scale, a, w["c2t"][i], g["obs_lla"], g["obs_itrs"],
                       -np.pi / 2, w["z_noise"][i, a], centred=centred, resample=resample)
        xt, x, P = r["x_true"], r["x"], r["P"]
        nf.append(int((st != 0).sum()))
        hist.append(st.copy())
        if i == DIVERGED_STEP:
            div = diverged_ids(np.where((st != 0)[:, None], np.nan, x))      # (a failed filter's sentinel state counts as diverged)
        d = r["metrics"][0]
        mx.append(np.nan if np.isnan(d).any() else float(d.max()))
    out = summarise(nf, st, mx, first_failure_steps(hist))
    out["diverged_at_%d" % DIVERGED_STEP] = div
    return out


def run_hip(hip, w, propagator, resample=False, covariance=None):
    """the same episode through ssa_env_step_f64 (one launch per step, every step's statistics kept: full history)"""
    torch = hip.torch
    g, m = w["g"], w["m"]
    consts = hip.host.make_consts(g["Q"], g["R"], 1e-4, 2.0, -3, 20.0, -np.pi / 2, g["obs_lla"], obs_type='aer',
                                  propagator=propagator, resample=resample, covariance=covariance)
    eng = hip.engine.HotPathEngine(consts, m, 1, w["c2t"], w["z_noise"][None], history=480)
    eng.load_state(0, w["x_true"], w["x"], w["P"])
    sched = torch.as_tensor((np.arange(N_STEPS) % m).astype(np.int32)).cuda()
    hist = torch.zeros((N_STEPS, m), dtype=torch.int32, device="cuda")
    for i in range(1, N_STEPS + 1):
        eng.launch_step(i - 1, i, i, actions_ptr=sched.data_ptr() + 4 * (i - 1), fast_stats=True)
        hist[i - 1].copy_(eng.status)          # (stream-ordered device copy: which filters have failed by step i)
    torch.cuda.synchronize()
    stats = eng.stats[1:N_STEPS + 1, 0].cpu().numpy()
    return summarise(stats[:, hip.lib.STAT_N_FAILED].astype(int), eng.status.cpu().numpy(), stats[:, hip.lib.STAT_MAX_DPOS],
                     first_failure_steps(hist.cpu().numpy()))
