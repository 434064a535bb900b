#!/usr/bin/env python3
"""bench.py -- env-steps/s of the ssa-gym hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--objects M] [--propagator fg|elements]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one call of the reference's SSA_Tasker_Env.step() (ssa_tasker_simple_2.py:243-367)
for M objects per GPU: M true-state propagations + M UKF predicts (13 Kepler solves, 6x6
Cholesky, unscented transform each) + one az-el-range UKF update + observations / error
metrics / reward statistics.  Inputs are synthetic (catalogue.synthetic_catalogue: drawn by the
reference's own recipe, envs/orbit_gen.py:30-70 -- regime probabilities AND the visibility
acceptance rule; the reference's own catalogue file does not travel) and resident in
HBM before the timed region.  Workload at N=1: BASELINE config 3 (20 000 objects, two-body
Farnocchia; the "J2 on" of that config has no counterpart in the reference -- SURVEY section 0).
N>1: BASELINE config 4 -- one env of N x 20 000 objects sharded 20k per GPU, with one RCCL
all-gather per step reassembling the global (az, el, range, trace P) observation vector and the
reward statistics (parallel.py).  `value` = 20 000-object env-steps per second summed over GPUs.

Prints ONE JSON line (rank 0).  `roofline` prices the dominant kernel (the fused step kernel)
against HBM; `cpu_baseline` times the single-threaded C oracle (a port of the reference's
per-object loop; the reference's numba/filterpy stack is not installable here) on the host.
"""
import argparse
import json
import gc
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALG_BYTES_PER_OBJECT_STEP = 896   # r+w x_true 48, x 48, P 288 each way; obs 96; metrics 32
HBM_PEAK_GBS = 8000.0             # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
FP64_FLOP_PER_OBJECT_STEP = 6400  # profiles/r03_counters.json (fg kernel): (ADD 18 + MUL 89 + 2 FMA 106 + TRANS 10) x 64 lanes + 9 MFMA x 512, per 4 objects
FP64_PEAK_TFLOPS = 78.6           # MI355X datasheet fp64 vector peak (an FMA micro-benchmark reaches 63.2 on these boxes)


def build_problem(m, seed, n_time=480):
    from ssa_gym_amd import host
    from ssa_gym_amd.catalogue import synthetic_catalogue
    from ssa_gym_amd.envs.transformations import trans_matrix_table
    from datetime import datetime
    cat = synthetic_catalogue(20000, seed=0)
    rs = np.random.RandomState(seed)
    x_true = cat[np.arange(m) % len(cat)].copy()
    x_sigma = np.array([1e5] * 3 + [1e2] * 3)
    x = x_true + rs.normal(size=(m, 6)) * x_sigma                       # envs/__init__.py:25 x_sigma
    P0 = np.diag(x_sigma ** 2)
    Q = host.Q_discrete_white_noise(dim=2, dt=20.0, var=0.000025 ** 2, block_size=3, order_by_dim=False)
    R = np.diag([host.arcsec2rad ** 2] * 2 + [1e3 ** 2])
    obs_lla = np.array((38.828198, -77.305352, 20.0)) * [host.deg2rad, host.deg2rad, 1]
    trans = trans_matrix_table(datetime(2020, 5, 4, 0, 0, 0), 20.0, n_time)
    z_sigma = np.array([host.arcsec2rad, host.arcsec2rad, 1e3])
    return dict(x_true=x_true, x=x, P0=P0, Q=Q, R=R, obs_lla=obs_lla, trans=trans, z_sigma=z_sigma)


def host_cpu_share():
    """CPUs this process may actually use: the affinity mask capped by the cgroup CPU quota (a container that sees
    256 CPUs with a 16-CPU quota collapses when 256 threads share it: 3.9 env-steps/s against 144 with 16,
    build_ablate/cpu_threads_sweep.py)."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]          # cgroup v2
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())             # cgroup v1
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // per))
        except (OSError, ValueError):
            pass
    return n


MIN_TIMED_S = 0.05      # every timed block is repeated until the blocks add up to >= 50 ms (timed_repeats)
MIN_REPEATS, MAX_REPEATS = 3, 400


def timed_repeats(block, sync, agree=None, prepare=None, group=1, min_groups=3):
    """`block()` enqueues EXACTLY the K steps of one timed block; `sync()` is the fence on both sides of it.  The block is
    repeated until the timed blocks add up to MIN_TIMED_S (a 20-step block of 14 us steps is 0.3 ms: one sample of it says
    little); returns (typical, min, max, repeats) of the per-block elapsed seconds.
    `group`: how many CONSECUTIVE blocks make one sample -- callers whose blocks walk through an episode pass the number of blocks
    that span one (ceil(479 / K)).  With the behaviour-faithful default a step late in an episode costs twice a step early in it;
    the MEDIAN BLOCK (rounds 1-3) then reads the healthy majority (56.9 k with the driver's 20-step blocks where whole episodes run
    at 51.7 k, below the kernel's own whole-episode mean), the plain MEAN is exact but takes every hiccup of a shared box at face
    value.  typical = the median over the groups of the mean block time inside each group: every sample covers a whole episode,
    and the median over samples keeps the robustness.  group = 1: the median block, as before.
    `agree(x)` makes a per-rank number rank-uniform (MAX over ranks), so that every rank runs the same number of repeats.
    `prepare()` runs untimed before every block (e.g. an env reset when the episode would end inside the block)."""
    agree = agree or (lambda v: v)
    prepare = prepare or (lambda: None)
    group = max(1, int(group))
    el = []
    # (the interpreter's cyclic garbage collector: a full collection over everything the set-up allocated is 40-50 ms, and its allocation
    # counter tripped inside the FIRST timed block of `value`, reproducibly -- one 20-step block of 48 ms in `value_spread`.  Collected now,
    # and what survives is frozen: later collections look at the few objects the timed loop itself creates.)
    gc.collect()
    gc.freeze()
    prepare()
    sync()
    t0 = time.perf_counter()
    block()
    sync()
    el.append(agree(time.perf_counter() - t0))
    reps = int(min(MAX_REPEATS, max(MIN_REPEATS, np.ceil(MIN_TIMED_S / max(el[0], 1e-9)))))
    if group > 1:                        # whole groups, at least three of them
        reps = group * max(min_groups, -(-reps // group))
    for _ in range(reps - 1):
        prepare()
        sync()
        t0 = time.perf_counter()
        block()
        sync()
        el.append(agree(time.perf_counter() - t0))
    samples = np.asarray(el).reshape(-1, group).mean(axis=1)
    if os.environ.get("SSA_BENCH_BLOCKS"):       # (diagnostic: every block's elapsed time, one line per timed_repeats call)
        with open(os.environ["SSA_BENCH_BLOCKS"], "a") as fh:
            fh.write(" ".join("%.6f" % v for v in el) + "\n")
    return float(np.median(samples)), float(min(el)), float(max(el)), len(el)


def _drain():
    import torch
    torch.cuda.synchronize()


def episode_groups(steps_per_block, episode=479):
    """blocks that span one episode (timed_repeats' `group`)"""
    return max(1, -(-episode // max(1, int(steps_per_block))))


def spread(K, scale, med, lo, hi, reps):
    """fields every leg reports next to its rate: repeats and the min / max rate over them"""
    return {"repeats": reps, "value_spread": [round(K / hi * scale, 2), round(K / lo * scale, 2)]}


def cpu_baseline(m, budget_s=15.0, all_cores=False):
    """C oracle (oracle/ssa_oracle.c), same step definition, bounded sample: one host core, or (all_cores) its
    OpenMP build over the cores this process may run on."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as orc
    orc.build()
    o = orc.Oracle(omp=all_cores)
    cores = o.lib.orc_omp_threads(host_cpu_share()) if all_cores else 1
    pb = build_problem(m, seed=0)
    Wm, Wc, scale = orc.merwe_weights(1e-4, 2.0, -3)
    xt, x, P = pb["x_true"], pb["x"], np.tile(pb["P0"], (m, 1, 1))
    status = np.zeros(m, dtype=np.int32)
    obs_itrs = o.lla2ecef(pb["obs_lla"])
    rs = np.random.RandomState(1)
    steps, t0 = 0, time.perf_counter()
    while True:
        i = steps + 1
        r = o.env_step(xt, x, P, status, 20.0, pb["Q"], pb["R"], Wm, Wc, scale, (i - 1) % m, pb["trans"][i % 480],
                       pb["obs_lla"], obs_itrs, -np.pi / 2, rs.normal(size=3) * pb["z_sigma"])
        xt, x, P = r["x_true"], r["x"], r["P"]
        steps += 1
        el = time.perf_counter() - t0
        if el > budget_s or steps >= 2000:
            break
    return {"value": steps / el, "unit": "env-steps/s", "cores": cores, "kind": "port",
            "sample": "%d steps of the %d-object env on %d host core(s) (oracle/ssa_oracle.c, gcc -O2%s), %.1f s"
                      % (steps, m, cores, " -fopenmp" if all_cores else "", el)}


EPISODE_M, EPISODE_WINDOWS = 2000, (240, 300, 360, 420, 479)


def episode_failures_hip(propagator):
    """failed filters over ONE 479-step round-robin episode of 2 000 objects (env defaults, alpha = 1e-4, an update every step) on
    the HIP path: counts at steps 240 .. 479 and the status-code mix.  The reference loses 2-3 % of its filters this way
    (LinAlgError from an exhausted robust_cholesky ladder on a diverged prior: ssa_tasker_simple_2.py:271-285, dynamics.py:402-417)."""
    import torch
    from ssa_gym_amd import _lib, engine, host
    m = EPISODE_M
    pb = build_problem(m, seed=7)
    consts = host.make_consts(pb["Q"], pb["R"], 1e-4, 2.0, -3, 20.0, -np.pi / 2, pb["obs_lla"], obs_type='aer', propagator=propagator)
    zn = np.random.RandomState(1).normal(size=(480, m, 3)) * pb["z_sigma"]
    eng = engine.HotPathEngine(consts, m, 1, pb["trans"], zn[None], history=480)
    eng.load_state(0, pb["x_true"], pb["x"], np.broadcast_to(pb["P0"], (m, 6, 6)))
    sched = torch.as_tensor((np.arange(479) % m).astype(np.int32)).cuda()
    for i in range(1, 480):
        eng.launch_step(i - 1, i, i, actions_ptr=sched.data_ptr() + 4 * (i - 1), fast_stats=True)
    torch.cuda.synchronize()
    nf = eng.stats[1:480, 0, _lib.STAT_N_FAILED].cpu().numpy().astype(int)
    st = eng.status.cpu().numpy()
    return {"failed_at_step": {str(w): int(nf[w - 1]) for w in EPISODE_WINDOWS}, "linalg": int((st == _lib.ST_PREDICT_LINALG).sum()),
            "nan": int((st == _lib.ST_PREDICT_NAN).sum())}


def episode_failures_oracle():
    """the same episode on the CPU oracle (reference order of operations; part of the cpu_baseline leg)"""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as orc
    orc.build()
    o = orc.Oracle(omp=True)
    o.lib.orc_omp_threads(host_cpu_share())
    m = EPISODE_M
    pb = build_problem(m, seed=7)
    Wm, Wc, scale = orc.merwe_weights(1e-4, 2.0, -3)
    zn = np.random.RandomState(1).normal(size=(480, m, 3)) * pb["z_sigma"]
    xt, x, P = pb["x_true"], pb["x"], np.tile(pb["P0"], (m, 1, 1))
    st = np.zeros(m, dtype=np.int32)
    obs_itrs = o.lla2ecef(pb["obs_lla"])
    nf = []
    for i in range(1, 480):
        a = (i - 1) % m
        r = o.env_step(xt, x, P, st, 20.0, pb["Q"], pb["R"], Wm, Wc, scale, a, pb["trans"][i], pb["obs_lla"], obs_itrs, -np.pi / 2, zn[i, a])
        xt, x, P = r["x_true"], r["x"], r["P"]
        nf.append(int((st != 0).sum()))
    return {"failed_at_step": {str(w): nf[w - 1] for w in EPISODE_WINDOWS}, "linalg": int((st == orc.ST_PREDICT_LINALG).sum()),
            "nan": int((st == orc.ST_PREDICT_NAN).sum())}


def local_variant_rate(m, K, W, propagator, resample=False, seed=100, layout=True):
    """the N=1 measurement of `value` for another kernel variant on the same workload: K timed per-step launches (one
    launch per step, deferred statistics fold, episodes of 480 steps with device-side resets) after W warm-up steps"""
    import torch
    from ssa_gym_amd import engine, host, parallel
    pb = build_problem(m, seed=seed)
    consts = host.make_consts(pb["Q"], pb["R"], 1e-4, 2.0, -3, 20.0, -np.pi / 2, pb["obs_lla"], obs_type='aer',
                              propagator=propagator, resample=resample)
    gen = torch.Generator(device="cuda").manual_seed(1)
    zn = torch.randn((1, 480, m, 3), dtype=torch.float64, device="cuda", generator=gen) * torch.as_tensor(pb["z_sigma"], device="cuda")
    eng = engine.HotPathEngine(consts, m, 1, pb["trans"], zn, history=2)
    if layout:       # the engine's storage layout (objects of one orbit regime share wavefronts; invisible to the caller: see `caller_order`)
        from ssa_gym_amd.catalogue import regime_order
        eng.set_layout(regime_order(pb["x_true"]))
    eng.load_state(0, pb["x_true"], pb["x"], np.broadcast_to(pb["P0"], (m, 6, 6)))
    local = parallel.HipLocalStepper(eng, consts, fast_stats=True, defer_fold=True)
    local.load_schedule(list(np.arange(W + K) % m))
    snap = eng.snapshot(0)
    i = 0

    def run(n):
        nonlocal i
        for _ in range(n):
            if i == 479:
                local.reset_episode(snap, 480)
                i = 0
            i += 1
            local.step(-1)
    run(W)

    def sync():
        local.flush()
        torch.cuda.synchronize()
    el, lo, hi, reps = timed_repeats(lambda: run(K), sync, group=episode_groups(K))
    out = {"value": round(K / el * (m / 20000.0), 2), "ms_per_step": round(1e3 * el / K, 5),
           "failed_filters": int((eng.status != 0).sum().item())}
    out.update(spread(K, m / 20000.0, el, lo, hi, reps))
    # this variant's kernel against the same roofline as `roofline` prices the headline's: one whole episode of per-step launches, the
    # dominant kernel timed by the event pair bound to each dispatch
    from ssa_gym_amd import _lib
    nl = min(479, _lib.PROFILE_SLOTS)
    local.reset_episode(snap, 480)
    torch.cuda.synchronize()
    for k in range(nl):
        local.step(-1, profile_slot=k)
    sync()
    kern_ms = sum(eng.profile_ms(k) for k in range(nl)) / nl
    out["kernel_ms"] = round(kern_ms, 5)
    out["roofline_frac"] = round(ALG_BYTES_PER_OBJECT_STEP * m / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)
    del eng, local, zn
    return out


def closed_loop_rate(m, K, W, seed=100, persistent=True, chunk=120, agent=None, propagator='hybrid', layout=True):
    """closed loop WITHOUT the host: the reference's agent_visible_greedy (agents.py:36: arg-max of trace(P) over the visible
    objects) chooses every step's action on the device.  persistent: ssa_env_closed_loop_f64 -- `chunk` steps and their decisions
    per launch, the wavefronts agree on the next action among themselves while the next predicts already run; else the
    multi-launch form: per step one launch of the step kernel, then ssa_agent_select_f64 (two small launches) writes the action
    word the next step reads, all in one stream.  Same workload, episodes of 480 steps."""
    import torch
    from ssa_gym_amd import _lib, engine, host
    pb = build_problem(m, seed=seed)
    consts = host.make_consts(pb["Q"], pb["R"], 1e-4, 2.0, -3, 20.0, -np.pi / 2, pb["obs_lla"], obs_type='aer', propagator=propagator)
    gen = torch.Generator(device="cuda").manual_seed(1)
    zn = torch.randn((1, 480, m, 3), dtype=torch.float64, device="cuda", generator=gen) * torch.as_tensor(pb["z_sigma"], device="cuda")
    eng = engine.HotPathEngine(consts, m, 1, pb["trans"], zn, history=2)
    if layout:      # the engine's storage layout (objects of one orbit regime share wavefronts; the agent kernels and the persistent launch take the table)
        from ssa_gym_amd.catalogue import regime_order
        eng.set_layout(regime_order(pb["x_true"]))
    eng.load_state(0, pb["x_true"], pb["x"], np.broadcast_to(pb["P0"], (m, 6, 6)))
    snap = eng.snapshot(0)
    word = torch.zeros(1, dtype=torch.int32, device="cuda")
    fb = torch.zeros(max(K, W, chunk) + 2, dtype=torch.int32, device="cuda")
    picks = torch.zeros((K + W + 2 + chunk, 2), dtype=torch.int64, device="cuda")
    log = torch.zeros(chunk + 1, dtype=torch.int32, device="cuda")
    stats = torch.zeros((chunk, _lib.STAT_STRIDE), dtype=torch.float64, device="cuda")
    st = {"i": 0, "tick": 0, "n": 0}
    AG = _lib.AGENT_VISIBLE_GREEDY if agent is None else int(agent)

    def new_episode():
        eng.flush_stats()
        st["tick"] += (-st["tick"]) % 480
        eng.restore(st["tick"] % 2, snap)
        st["i"] = 0
        eng.launch_agent_select(st["tick"], st["tick"], AG, (log if persistent else word).data_ptr(), fallback_ptr=fb.data_ptr())

    def run(n):
        done = 0
        while done < n:
            if st["i"] == 479:
                new_episode()
            if persistent:
                kk = min(chunk, n - done, 479 - st["i"])
                t = st["tick"]
                ok = eng.launch_closed_loop(t % 2, t + 1, AG, log[:kk + 1], stats, fallback=fb[:kk + 1],
                                            picks=picks[st["n"]:st["n"] + kk + 1])
                if not ok:
                    raise RuntimeError("ssa_env_closed_loop_f64 declined %d objects" % m)
                log[:1].copy_(log[kk:kk + 1])        # the decision after the chunk's last step opens the next chunk
                st["i"] += kk
                st["tick"] += kk
                st["n"] += kk
                done += kk
                continue
            st["i"] += 1
            st["tick"] += 1
            t = st["tick"]
            eng.launch_step((t - 1) % 2, t % 2, t, actions_ptr=word.data_ptr(), fast_stats=True, defer_fold=True)
            eng.launch_agent_select(t, t, AG, word.data_ptr(), fallback_ptr=fb.data_ptr(),
                                    pick_ptr=picks.data_ptr() + 16 * (st["n"] + 1))
            st["n"] += 1
            done += 1
    eng.launch_agent_select(0, 0, AG, (log if persistent else word).data_ptr(), fallback_ptr=fb.data_ptr())
    run(W)

    def sync():
        eng.flush_stats()
        torch.cuda.synchronize()

    def block():
        st["n"] = W
        run(K)
    el, lo, hi, reps = timed_repeats(block, sync, group=episode_groups(K))
    if persistent and int(eng.loop_error[0]) != 0:
        raise RuntimeError("ssa_env_closed_loop_f64 gave up on a timeout")
    chosen = picks[W + 1:W + K + 1, 0].cpu().numpy()
    return {"value": round(K / el * (m / 20000.0), 2), "ms_per_step": round(1e3 * el / K, 5), **spread(K, m / 20000.0, el, lo, hi, reps),
            "agent": "agent_visible_greedy (device)", "propagator": propagator, "distinct_objects_selected": int(len(set(chosen.tolist()))),
            "failed_filters": int((eng.status != 0).sum().item()),
            "note": ("closed loop in ONE persistent launch per %d steps (ssa_env_closed_loop_f64): state resident in LDS, the decision made "
                     "between the wavefronts while the next predicts run; no host round trip" % chunk) if persistent else
                    "closed loop on the device: step launch + ssa_agent_select_f64 (2 launches) per step, no host round trip"}


def gym_api_rate(m, mode, n=200, obs_device=False, zero_copy=False, obs_pool=64, f32=False):
    """env.step() through the gym API (host in the loop: action in, launch, one sync, statistics + observation out over
    PCIe): the closed-loop rate an unmodified agents.py / RLlib worker sees.  Never `value`.  zero_copy: config['obs_zero_copy'] --
    step() hands out a view of the host-mapped ring instead of a fresh copy (the default, as the reference)."""
    from ssa_gym_amd.envs import env_config, make
    cfg = dict(env_config)
    cfg.update(rso_count=m, steps=480, reward_type='trinary', obs_returned=mode, seed=0, history=2, device_rng=True, obs_device=obs_device,
               obs_zero_copy=zero_copy, obs_pool=obs_pool, obs_dtype=np.float32 if f32 else np.float64)
    env = make(config=cfg)
    for k in range(20):
        env.step(k % m)
    cnt = {"k": 20}

    def prepare():
        if env.i + n >= env.n - 1:
            env.reset()
            _drain()            # (a reset is untimed: none of its uploads / kernels may still be queued when the next timed block starts)

    def block():
        for _ in range(n):
            env.step(cnt["k"] % m)
            cnt["k"] += 1
    el, lo, hi, reps = timed_repeats(block, lambda: None, prepare=prepare)
    dt = el / n
    return {"value": round(1.0 / dt * (m / 20000.0), 2), "ms_per_step": round(1e3 * dt, 5), **spread(n, m / 20000.0, el, lo, hi, reps),
            "obs_bytes_per_step": 0 if obs_device else m * (12 if mode == 'flatten' else 4) * (4 if f32 else 8)}


def torch_policy_rate(m, n=192):
    """env.run_policy(): the closed loop with a policy written in torch (here the visible-greedy rule as tensor expressions on the
    device scores; any torch module fits) -- the policy's own kernels + the step launch per step, captured ONCE per 32 steps into a
    hipGraph and replayed (`graph`), or enqueued kernel by kernel from the host (`eager`); no host round trip per step either way, one
    synchronisation per chunk.  Never `value`."""
    import torch
    from ssa_gym_amd.envs import env_config, make
    cfg = dict(env_config)
    cfg.update(rso_count=m, steps=480, reward_type='trinary', obs_returned='flatten', seed=0, history=2, device_rng=True, obs_limit=10.0)
    env = make(config=cfg)

    def policy(view):
        sc, mask = view.scores()
        return torch.argmax(torch.where(mask.view(torch.bool), sc[0], float("-inf"))).to(torch.int32).reshape(1)
    fixed = torch.zeros(1, dtype=torch.int32, device="cuda")

    def fixed_policy(view):
        return fixed
    picked = []

    def prepare():
        if env.i + n >= env.n - 1:
            env.reset()
            _drain()            # (a reset is untimed: none of its uploads / kernels may still be queued when the next timed block starts)

    def measure(pol, graph):
        for _ in range(3):          # (untimed: torch's first launches of each expression, the caching allocator, the graph captures)
            prepare()
            env.run_policy(pol, n, graph=graph)

        def block():
            a, _, _ = env.run_policy(pol, n, graph=graph)
            picked.extend(a.tolist())
        return timed_repeats(block, lambda: None, prepare=prepare)
    neg_inf = torch.full((), float("-inf"), dtype=torch.float64, device="cuda")

    def policy_lean(view):          # the same rule written for the GPU: the -inf operand preallocated, torch.argmax's int64 handed back as it is
        sc, mask = view.scores()
        return torch.argmax(torch.where(mask.view(torch.bool), sc[0], neg_inf))

    def policy_head(view):          # the same rule with the env's one-launch arg-max head instead of torch.where + torch.argmax + a cast
        sc, mask = view.scores()
        return view.argmax(sc[0], mask)
    el0, _, _, _ = measure(fixed_policy, False)     # the env's side alone: a policy that returns a preallocated tensor
    elg0, _, _, _ = measure(fixed_policy, True)
    ele, _, _, _ = measure(policy, False)
    elh, _, _, _ = measure(policy_head, True)
    ell, _, _, _ = measure(policy_lean, True)
    picked.clear()
    el, lo, hi, reps = measure(policy, True)
    dt = el / n
    return {"value": round(1.0 / dt * (m / 20000.0), 2), "ms_per_step": round(1e3 * dt, 5), **spread(n, m / 20000.0, el, lo, hi, reps),
            "distinct_objects_selected": len(set(picked)), "graph_error": env.policy_graph_error,
            "eager": {"value": round(n / ele * (m / 20000.0), 2), "ms_per_step": round(1e3 * ele / n, 5),
                      "note": "the same policy with every kernel enqueued from the host (run_policy(graph=False): round 3's form)"},
            "env_side_only": {"value": round(n / el0 * (m / 20000.0), 2), "ms_per_step": round(1e3 * el0 / n, 5),
                              "note": "eager, with a policy that returns a preallocated tensor: what run_policy itself costs (step launch + bookkeeping)"},
            "env_side_only_graph": {"value": round(n / elg0 * (m / 20000.0), 2), "ms_per_step": round(1e3 * elg0 / n, 5),
                                    "note": "the same from replayed graphs: the step launches alone, bookkeeping of chunk c hidden behind chunk c + 1"},
            "lean_torch": {"value": round(n / ell * (m / 20000.0), 2), "ms_per_step": round(1e3 * ell / n, 5),
                           "note": "graph replay; the same rule in two torch kernels less: the -inf operand preallocated (the scalar form launches a fill), "
                                   "torch.argmax's int64 handed back as it is (run_policy reads its low word: no cast kernel)"},
            "argmax_head": {"value": round(n / elh * (m / 20000.0), 2), "ms_per_step": round(1e3 * elh / n, 5),
                            "note": "graph replay; the policy = view.scores() + view.argmax(score, mask) (PolicyView.argmax: the arg-max head as ONE launch of the "
                                    "library, np.argmax semantics) instead of torch.where + torch.argmax + .to(int32), which are 4 launches and 25 us of the 45 a step "
                                    "of the torch expression keeps the GPU busy (profiles/r04_run_policy_timeline.txt)"},
            "note": "SSA_Tasker_Env.run_policy(): %d steps per call = %d replays of a 32-step hipGraph holding, per step, the policy's torch kernels "
                    "(scores kernel + where + argmax + cast) and the step launch that reads their action word from device memory; the time index "
                    "advances on the device; the host books chunk c while the GPU runs chunk c + 1" % (n, n // 32)}


def vec_env_rate(m, E=8, n=68, obs_device=False, zero_copy=False, f32=False, layout=False):
    """BASELINE config 5's per-GPU load through the vector-env API: E envs of m objects advanced by ONE launch per
    SSA_Tasker_VecEnv.step() (per-env actions and time indices, auto-reset), host in the loop, the E 'aer' observation
    vectors returned over PCIe.  Reported in 20 000-object env-steps per second (E per call).  Never `value`."""
    from ssa_gym_amd.envs import env_config
    from ssa_gym_amd.envs.vector_env import SSA_Tasker_VecEnv
    cfg = dict(env_config)
    cfg.update(rso_count=m, steps=480, reward_type='trinary', obs_returned='aer', seed=0, device_rng=True, obs_device=obs_device,
               obs_zero_copy=zero_copy, obs_dtype=np.float32 if f32 else np.float64, storage_layout='regime' if layout else None)
    env = SSA_Tasker_VecEnv(cfg, num_envs=E, seed=0)
    acts = lambda k: [(k * 7 + 13 * e) % m for e in range(E)]    # noqa: E731
    for k in range(10):
        env.step(acts(k))
    env.reset()                 # (the timed blocks start with an episode: seven 68-step blocks span one, steps 1-476 of 479)
    _drain()
    cnt = {"k": 10}

    def block():
        for _ in range(n):
            env.step(acts(cnt["k"]))
            cnt["k"] += 1

    def prepare():
        # (round 3's line showed one repeat in six 3.6 x slower than the others, reproducibly: the block in which the eight envs reach step
        # 479 and auto-reset -- eight host-side catalogue draws, 24 host-to-device copies of 1 MB and eight noise tables, ~25 ms inside a
        # 7 ms block.  A reset is not a step: the envs are reset UNTIMED when the next block would cross the episode's end, as the
        # gym-API legs do.)
        if int(env.i.max()) + n >= env.n - 1:
            env.reset()
            _drain()            # (a reset is untimed: none of its uploads / kernels -- the layout's gathers among them -- may still be queued
            #                      when the next timed block starts)
    # (whole episodes, as `value`: with the behaviour-faithful default a late block is 1.5 x slower than an early one where the kernel
    # dominates -- obs_device -- and the median BLOCK, the statistic of these legs until round 4's last build, was an early one)
    el, lo, hi, reps = timed_repeats(block, lambda: None, prepare=prepare, group=(env.n - 1) // n, min_groups=5)
    dt = el / n
    return {"value": round(E / dt * (m / 20000.0), 2), "ms_per_vector_step": round(1e3 * dt, 5), "envs": E,
            "timing": "median over whole episodes (%d blocks of %d vector steps each) of the mean block" % ((env.n - 1) // n, n),
            **spread(n, E * m / 20000.0, el, lo, hi, reps),
            "obs_bytes_per_step": 0 if obs_device else E * m * 4 * (4 if f32 else 8),
            "note": ("SSA_Tasker_VecEnv.step() with config['obs_device']: %d envs x %d objects per launch, host in the loop; the observations stay "
                     "on the GPU (CUDA tensor returned, for policies that live there), rewards / dones cross PCIe" % (E, m)) if obs_device else
                    "SSA_Tasker_VecEnv.step(): %d envs x %d objects per launch, host in the loop, PCIe inclusive" % (E, m)}


def self_launch(argv, n):
    """`python bench.py --gpus N` outside torch.distributed.run: start the N ranks as CHILD processes of a parent
    that never touches the GPU (no torch import here, no HIP call), relay rank 0's JSON line and exit with the
    launcher's code.  One process per GPU, rendezvous on 127.0.0.1."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in proc.stdout:
        if ln.lstrip().startswith("{") and '"metric"' in ln:
            line = ln.strip()
        else:
            sys.stderr.write(ln)
    rc = proc.wait()
    if line is not None:
        print(line, flush=True)
    raise SystemExit(rc if rc else (0 if line is not None else 1))


def dry_run(args):
    """launcher / rendezvous / collective plumbing of the N-rank path without a GPU (CPU test of `--gpus N`): gloo
    process group, the ShardedStepper host logic over a payload-only local stepper (no step arithmetic: nothing here
    computes the hot path), rank 0 prints the JSON line."""
    import torch
    import torch.distributed as dist
    from ssa_gym_amd import parallel
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    m = args.objects
    plan = parallel.ShardPlan(m * world, world, rank)

    class PayloadOnly:
        device = "cpu"

        def step(self, local_action, obs_out=None, stats_out=None, profile_slot=None):
            obs_out.fill_(float(rank))
            stats_out.zero_()
            stats_out[parallel.STAT_CNT_LT_1E7] = plan.m_local
            stats_out[parallel.STAT_ARGMAX_SPOS] = -1.0
    sh = parallel.ShardedStepper(plan, PayloadOnly())
    t0 = time.perf_counter()
    for k in range(args.steps):
        sh.step(k % plan.m_total)
    dist.barrier()
    el = time.perf_counter() - t0
    obs, st = sh.global_obs(), sh.global_stats()
    ok = bool(obs.numel() == 4 * plan.m_total and st[parallel.STAT_CNT_LT_1E7] == plan.m_total and
              all(float(obs[4 * plan.offsets[r]]) == float(r) for r in range(world)))
    if rank == 0:
        print(json.dumps({"metric": "env_steps_per_sec_at_20k_objects", "value": None, "dry_run": True, "n_gpus": world,
                          "steps": args.steps, "ms_per_step": round(1e3 * el / max(1, args.steps), 5),
                          "config": {"workload": "DRY RUN (gloo, no GPU, no step arithmetic): launcher + all-gather plumbing",
                                     "ranks": dist.get_world_size(), "payload_ok": ok}}), flush=True)
    dist.barrier()
    dist.destroy_process_group()
    if not ok:
        raise SystemExit(3)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--objects", type=int, default=20000, help="objects per GPU")
    ap.add_argument("--propagator", default="hybrid", choices=["hybrid", "fg", "elements", "j2"],
                    help="hybrid (the env default: behaviour-faithful) / fg / elements: two-body Farnocchia (parity-checked); "
                         "j2: J2+RK4 extension (no reference counterpart)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--rollout", type=int, default=60,
                    help="steps per launch of the additional open-loop rollout measurement (0 = skip); never `value`")
    ap.add_argument("--dry-run", action="store_true",
                    help="CPU rehearsal of the N-rank launch path (gloo, no GPU, no step arithmetic); never a measurement")
    ap.add_argument("--payload", default="trace", choices=["trace", "aer"],
                    help="sharded runs: what the per-step all-gather carries per object -- trace P (BASELINE config 4: \"all-gather of per-object "
                         "covariance-trace obs\") or the four-column (az, el, range, trace P) block of the 'aer' observation mode")
    ap.add_argument("--eager-sharded", action="store_true",
                    help="sharded runs: enqueue every step from the host (default: hipGraph replay of whole units of steps)")
    ap.add_argument("--no-legs", action="store_true", help="skip the additional N=1 legs (j2, elements, resample, gym_api, closed_loop)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and "RANK" not in os.environ:
        self_launch(sys.argv[1:], args.gpus)      # does not return
    if args.dry_run:
        return dry_run(args)

    # stdout carries exactly ONE line -- the JSON record.  Native libraries write there too (RCCL prints a version banner at
    # communicator creation): from here on file descriptor 1 is stderr, and the record goes to the saved descriptor.
    sys.stdout.flush()
    real_stdout = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist
    import ssa_gym_amd
    from ssa_gym_amd import _lib, engine, host, parallel

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but the launcher started %d ranks" % (args.gpus, world))
    # SSA_BENCH_REHEARSAL=1: functional rehearsal of the N-rank path on a box with ONE card -- every rank on cuda:0, the
    # collectives carried by gloo (RCCL refuses two ranks on one device).  Exercises everything but RCCL itself; the ranks share
    # the GPU, so the line it prints is marked and is never a measurement.
    rehearsal = os.environ.get("SSA_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    use_dist = world > 1 or "RANK" in os.environ   # under torch.distributed.run even 1 rank goes through RCCL
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    if rank == 0:
        ssa_gym_amd.build()          # no-op when the in-tree libssa_hip.so is current
    if use_dist:
        dist.barrier()               # nobody loads the library before rank 0 has (re)built it
    _lib.load()

    m, K, W = args.objects, args.steps, args.warmup
    n_time = 480
    pb = build_problem(m, seed=100 + rank)
    consts = host.make_consts(pb["Q"], pb["R"], 1e-4, 2.0, -3, 20.0, -np.pi / 2, pb["obs_lla"], obs_type='aer',
                              propagator=args.propagator)
    gen = torch.Generator(device="cuda").manual_seed(1 + rank)
    z_noise = torch.randn((1, n_time, m, 3), dtype=torch.float64, device="cuda", generator=gen) * \
        torch.as_tensor(pb["z_sigma"], device="cuda")
    eng = engine.HotPathEngine(consts, m, 1, pb["trans"], z_noise, history=2)
    # the engine's storage layout (HotPathEngine.set_layout / ssa_step_params.obj_ids: objects of one orbit regime share wavefronts; the
    # kernels speak the caller's indices at every boundary, whole episodes are bit-identical to the caller's order); a rank of a sharded run
    # lays out its own shard (the all-gather payload is written at the caller's rows).  An ENGINE-level option: the gym-API legs do not use it (step() writes the
    # observation for the host, whose rows would then leave the kernel one by one: 73 -> 84 us per 'flatten' step)
    storage_layout = None
    # (every rank lays out its own shard; SSA_BENCH_LAYOUT=0: the caller's order.  Launches of more than 20 480 objects -- a wavefront then walks
    # several tiles, stride = the number of wavefronts -- take the plain sort: every wavefront's walk then runs through the same mix of regimes
    # (160 000 objects: 93 us per step against 113 in the caller's order; the one-tile dealing, where some wavefronts walk ONLY slow tiles, is slower than either))
    if m >= 64 and os.environ.get("SSA_BENCH_LAYOUT", "1") == "1":
        from ssa_gym_amd.catalogue import regime_order
        eng.set_layout(regime_order(pb["x_true"]))
        storage_layout = ("regime: objects stored by ascending semi-major axis, dealt tile by tile over the XCDs (catalogue.regime_order; HotPathEngine.set_layout -- an engine-level option, results bit-identical to the caller's order; `caller_order` is the same run without it)")
    eng.load_state(0, pb["x_true"], pb["x"], np.broadcast_to(pb["P0"], (m, 6, 6)))

    plan = parallel.ShardPlan(m * world, world, rank)
    # statistics by the two-launch path (max delta_pos, trinary counts, failures: everything the 'jones' and
    # 'trinary' rewards read); the sharded multi-GPU step needs the post kernel for its payload anyway
    # One launch per step on a single GPU: the statistics of step k are folded by extra wavefronts riding in step
    # k+1's launch (the last one by flush() before the closing fence), so every step's statistics exist when the
    # timed region ends.  The sharded multi-GPU step needs them in its all-gather payload and folds at once.
    local = parallel.HipLocalStepper(eng, consts, fast_stats=True, defer_fold=True, fold_inside=os.environ.get("SSA_BENCH_FOLD") == "inside")
    # round-robin actions over the GLOBAL catalogue (BASELINE.md protocol): a_i = i mod m_total
    total_steps = W + K
    glob_actions = np.arange(total_steps) % plan.m_total
    local.load_schedule([plan.local_action(int(a)) for a in glob_actions])
    obs_cols = 1 if args.payload == "trace" else 4
    sharded = parallel.ShardedStepper(plan, local, obs_cols=obs_cols) if use_dist else None

    # episodes of the reference's default length (env_config['steps'] = 480: step indices 1..479), then a
    # reset from the device-resident initial state -- a predict-only UKF at alpha = 1e-4 is numerically
    # unstable beyond a few hundred steps in the reference's own arithmetic, so an endless
    # episode would benchmark diverged filters
    snap = eng.snapshot(0)
    ep_len = n_time
    state = {"i": 0, "overlap": False}

    def one_step(k, profile_slot=None, local_only=False):
        if state["i"] == ep_len - 1:
            local.reset_episode(snap, ep_len)
            state["i"] = 0
        state["i"] += 1
        if sharded is not None and not local_only:
            sharded.step(int(glob_actions[k]), overlap=state["overlap"])
        else:
            local.step(-1, profile_slot=profile_slot)   # action comes from the pre-staged schedule

    def max_over_ranks_early(v):
        t = torch.tensor([v], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def fence():
        local.flush()
        if sharded is not None:
            sharded.wait()
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()

    # Where the all-gather runs is chosen during the warm-up, by measurement, identically on every rank: in the
    # compute stream (the global observation of step k is complete before step k+1 starts: what a closed-loop
    # agent needs), or on a communication stream so that RCCL moves step k's payload over xGMI while step k+1
    # computes (legitimate for this protocol's pre-staged round-robin schedule; the global observation is then
    # complete one step later).  Either way every step's payload is gathered inside the timed region.
    # ---- sharded runs: the steps of a timed block are replayed from a captured hipGraph (parallel.GraphedShardedSteps): the
    # host enqueues one graph per `unit` steps instead of 13-34 us of launches and event bookkeeping per step.  Episodes are
    # whole units (the env is reset when the next unit would cross step 479).  Which stream the all-gather runs in is again
    # decided by measurement (a few units of each form), identically on every rank.
    graphed = None
    divs = [K] if K <= 120 else [d for d in range(2, 121) if K % d == 0]
    graph_note = None
    # (more than one rank: opt-in with SSA_BENCH_GRAPH_SHARDED=1 -- the capture of RCCL's collective has only ever run on ONE rank here,
    # the per-step enqueue is the form the multi-rank logic was built and tested around)
    # (the peer-store exchange is plain kernels: its units are captured at any world size)
    want_graph = not args.eager_sharded and (world == 1 or os.environ.get("SSA_BENCH_GRAPH_SHARDED") == "1"
                                             or (sharded is not None and sharded._peer is not None))
    if sharded is not None and want_graph and divs:
        unit = max(divs)
        cyc = np.arange(plan.m_total)
        probe, best, err = {}, None, None
        try:
            if os.environ.get("SSA_BENCH_FORCE_GRAPH_FAIL"):      # (rehearsal of the fallback)
                raise RuntimeError("forced")
            for ov in ((False, True) if ((world > 1 or os.environ.get("SSA_BENCH_PROBE_OVERLAP")) and sharded._peer is None) else (False,)):
                local.reset_episode(snap, ep_len)
                gs = parallel.GraphedShardedSteps(sharded, unit, cyc, overlap=ov)
                gs.rewind()
                done_steps = 0
                for _ in range(3):          # (captures: one per phase seen)
                    if done_steps + unit > ep_len - 1:
                        local.reset_episode(snap, ep_len)
                        gs.rewind()
                        done_steps = 0
                    gs.run_unit()
                    done_steps += unit
                fence()
                reps = max(3, 60 // unit)
                t0 = time.perf_counter()
                for _ in range(reps):
                    if done_steps + unit > ep_len - 1:
                        local.reset_episode(snap, ep_len)
                        gs.rewind()
                        done_steps = 0
                    gs.run_unit()
                    done_steps += unit
                fence()
                probe[ov] = max_over_ranks_early((time.perf_counter() - t0) / (reps * unit))
                if best is None or probe[ov] < 0.97 * probe[best[0]]:
                    best = (ov, gs)
        except Exception as exc:  # noqa: BLE001   (e.g. a runtime that cannot capture the collective: every rank falls back together)
            err = exc
        agreed = max_over_ranks_early(0.0 if (err is None and best is not None) else 1.0) == 0.0
        if agreed:
            graphed = best[1]
            state["overlap"] = best[0]
            allgather_probe = {"graph_unit": unit, "in_stream_ms": round(1e3 * probe[False], 5),
                               "comm_stream_ms": round(1e3 * probe[True], 5) if True in probe else None}
            local.reset_episode(snap, ep_len)
            graphed.rewind()
            state["i"] = 0

            def timed_block_graph():
                for _ in range(K // unit):
                    if state["i"] + unit > ep_len - 1:
                        local.reset_episode(snap, ep_len)
                        graphed.rewind()
                        state["i"] = 0
                    graphed.run_unit()
                    state["i"] += unit
            for _ in range(max(1, W // K)):     # (untimed: every phase's graph exists before the timed blocks)
                timed_block_graph()
        else:
            graph_note = "hipGraph capture of the sharded unit failed on at least one rank (%r): per-step enqueue" % (err,)
            sys.stderr.write("bench: %s\n" % graph_note)
            eng.env_time0.zero_()
            local.reset_episode(snap, ep_len)
            state["i"] = 0
    if graphed is not None:
        pass
    elif sharded is not None and W >= 40:
        h = W // 2
        times = []
        for mode in (False, True):       # (untimed: the collective's first use sets up RCCL's channels -- 0.4 ms once -- and would
            state["overlap"] = mode      # be billed to whichever mode is probed first)
            for k in range(3):
                one_step(k)
        for mode, lo, hi in ((False, 0, h), (True, h, W)):
            state["overlap"] = mode
            fence()
            t0 = time.perf_counter()
            for k in range(lo, hi):
                one_step(k)
            fence()
            times.append((time.perf_counter() - t0) / (hi - lo))
        tt = torch.tensor(times, dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        state["overlap"] = bool(tt[1].item() < 0.97 * tt[0].item())   # overlap only when clearly faster
        allgather_probe = {"in_stream_ms": round(1e3 * tt[0].item(), 5), "comm_stream_ms": round(1e3 * tt[1].item(), 5)}
    else:
        allgather_probe = None
        for k in range(W):
            one_step(k)
        # (an episode's end lies inside some timed block: the restore of the snapshot runs once here, untimed, so that its copy kernels are
        # loaded -- the first use of a kernel in the process costs tens of milliseconds, which showed as one 45-70 ms block in `value_spread`)
        fence()
        local.reset_episode(snap, ep_len)
        state["i"] = 0
        fence()
    # The timed block: EXACTLY K steps between two fences (barrier + synchronize), MAX over ranks.  The driver's K = 20 makes
    # that block 0.3 ms, so it is repeated (same K, same fences) until the blocks add up to 50 ms and `value` is the MEDIAN
    # block; min / max go into `value_spread`.
    def max_over_ranks(v):
        if not use_dist:
            return v
        t = torch.tensor([v], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def timed_block():
        if os.environ.get("SSA_BENCH_STEPTIMES") and not state.get("dumped"):      # (diagnostic: host time of every enqueue of the FIRST timed block)
            ts = []
            for k in range(W, W + K):
                t1 = time.perf_counter()
                one_step(k)
                ts.append(round(1e3 * (time.perf_counter() - t1), 3))
            state["dumped"] = True
            sys.stderr.write("steptimes [ms] %s\n" % ts)
            return
        for k in range(W, W + K):
            one_step(k)
    elapsed, el_min, el_max, repeats = timed_repeats(timed_block_graph if graphed is not None else timed_block, fence, agree=max_over_ranks,
                                                         group=episode_groups(K))

    # sanity: nothing diverged during the run
    n_failed = int((eng.status != 0).sum().item())
    if graphed is not None:
        eng.env_time0.zero_()      # (the graphs advanced the time origin on the device; what follows enqueues steps one by one)

    # ---- dominant kernel, timed live with events on the launch stream: K back-to-back launches
    # of the fused step kernel alone (no statistics kernel, no collective)
    roof = None
    if rank == 0:
        # whole steps exactly as in the timed region, back to back; launch k of the dominant kernel is timed by the
        # HIP event pair bound to that dispatch (on the launch stream)
        nl = min(ep_len - 1, _lib.PROFILE_SLOTS)   # one whole 479-step episode, whatever --steps was
        local.reset_episode(snap, ep_len)
        state["i"] = 0
        torch.cuda.synchronize()
        for k in range(nl):   # (sharded runs: the same kernel, timed on the local stepper without the collective)
            one_step(0, profile_slot=k, local_only=True)
        local.flush()
        torch.cuda.synchronize()
        kern_ms = sum(eng.profile_ms(k) for k in range(nl)) / nl
        alg_bytes = ALG_BYTES_PER_OBJECT_STEP * m
        achieved = alg_bytes / (kern_ms * 1e-3) / 1e9
        traffic, traffic_src = None, None
        tj = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tj):      # (PMC passes cannot run inside this process: the figure is the committed profile's, labelled as such)
            try:
                tjd = json.load(open(tj))
                tkey = "%s_%d" % (args.propagator, m)
                traffic = tjd.get(tkey)
                rnd = tjd.get("collected", {})
                rnd = rnd.get(tkey, "committed profile") if isinstance(rnd, dict) else rnd
                traffic_src = ("profiles/traffic.json[%s] (collected in %s): rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this kernel, NOT measured "
                               "in this run" % (tkey, rnd)) if traffic is not None else None
            except Exception:  # noqa: BLE001
                traffic = None
        fp64_tflops = FP64_FLOP_PER_OBJECT_STEP * m / (kern_ms * 1e-3) / 1e12
        roof = {"bound": "hbm", "kernel": "ssa::step_fast_kernel<%d>" % {"elements": 0, "fg": 1, "j2": 2, "hybrid": 3}[args.propagator],
                "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_source": traffic_src,
                "kernel_ms": round(kern_ms, 5), "algorithmic_bytes_per_launch": alg_bytes, "launches_timed": nl,
                "fp64_frac": round(fp64_tflops / FP64_PEAK_TFLOPS, 4),
                "fp64": {"flop_per_object_step": FP64_FLOP_PER_OBJECT_STEP, "achieved_tflops": round(fp64_tflops, 2),
                         "peak_tflops": FP64_PEAK_TFLOPS,
                         "note": "flop count from the SQ_INSTS_VALU_*_F64 / MFMA counters of the fg kernel (profiles/r03_counters.json); datasheet vector peak"},
                "limiter": "per-wavefront dependency chain + VALU / LDS issue (HBM idles between the load and store bursts); "
                           "`bound` names the roofline north_star declares for the path, the kernel is not memory-bound",
                "note": "whole 479-step episode of back-to-back per-step launches, HIP event pair bound to each dispatch"}

    # ---- additional line (never `value`): the same K steps through ssa_env_rollout_f64, `--rollout` steps per
    # launch -- what an open-loop schedule such as this protocol's round-robin allows (state resident in LDS across
    # the steps of a launch; every step's outputs still written)
    roll = None
    if rank == 0 and world == 1 and not use_dist and args.rollout > 0:
        R = args.rollout

        def roll_steps(n):      # (the rollout launches keep the engine's storage layout)
            done = 0
            while done < n:
                if state["i"] == ep_len - 1:
                    local.reset_episode(snap, ep_len)
                    state["i"] = 0
                kk = min(R, n - done, ep_len - 1 - state["i"])
                local.rollout(kk)
                state["i"] += kk
                done += kk
        Kr = max(K, 4 * R)        # (at least four full launches per timed block, whatever --steps was)
        roll_steps(max(W, R))
        el, lo, hi, reps = timed_repeats(lambda: roll_steps(Kr), torch.cuda.synchronize, group=episode_groups(Kr))
        roll = {"steps_per_launch": R, "steps": Kr, "value": round(Kr / el * (m / 20000.0), 2), "ms_per_step": round(1e3 * el / Kr, 5),
                **spread(Kr, m / 20000.0, el, lo, hi, reps),
                "failed_filters": int((eng.status != 0).sum().item()),
                "note": "open-loop schedule only (actions of a launch known up front); bit-identical to per-step launches"}

    # ---- additional N=1 legs on the same 20 000-object workload, each measured exactly like `value` (never `value`):
    # the J2 extension (BASELINE config 3's "J2 on"), the reference-operation-order propagator, the redraw variant of
    # predict() (the UKF default is parity-unpinned: filterpy is absent), and the gym API with the host in the loop
    legs = {}
    if rank == 0 and world == 1 and not use_dist and not args.no_legs:
        Kl, Wl = min(max(K, 200), 1000), min(max(W, 50), 100)   # (legs: at least 200 steps per timed block)
        for name, kw in (("fg", dict(propagator="fg")), ("hybrid", dict(propagator="hybrid")), ("j2", dict(propagator="j2")),
                         ("elements", dict(propagator="elements")), ("resample", dict(propagator=args.propagator, resample=True))):
            if name == args.propagator:
                continue
            legs[name] = local_variant_rate(m, Kl, Wl, **kw)
        legs["caller_order"] = local_variant_rate(m, Kl, Wl, propagator=args.propagator, layout=False)
        legs["caller_order"].update(note="the same launches with the objects STORED as the caller numbers them (rounds 1-3; what the gym-API legs run). "
                                         "`value` and the other engine-level per-step legs run with the engine's storage layout: ascending semi-major axis, dealt "
                                         "tile by tile over the XCDs (catalogue.regime_order; HotPathEngine.set_layout) -- late in an episode the diverged "
                                         "filters are the LEO objects, and in catalogue order 76 % of the wavefronts hold at least one.  The kernels speak "
                                         "the caller's indices at every boundary and an object's arithmetic does not depend on its position: whole "
                                         "episodes are bit-identical either way (build_ablate/layout_episode_ab.py)")
        if "j2" in legs:
            legs["j2"].update(note="EXTENSION without reference counterpart (SURVEY section 0): two-body + J2, RK4, 4 sub-steps")
        legs["resample"].update(note="predict() redraws the sigma points from the prior (SSA_FLAG_RESAMPLE); `value` keeps the "
                                     "propagated points; which of the two the reference's unpinned filterpy does is unverifiable offline")
        if "fg" in legs:
            legs["fg"].update(note="SSA_PROP_FG: every conic through one universal-variable equation -- more accurate than the reference on diverged "
                                   "states, so its filters survive where the reference's fail (`episode_failures`): the explicitly named accuracy / "
                                   "speed option (fx_xyz_farnocchia_fg); `value` of rounds 1-3 was measured on it")
        legs["closed_loop_per_step_launches"] = closed_loop_rate(m, Kl, Wl, persistent=False, propagator=args.propagator)
        if m <= 20160:      # (one wavefront per four objects + the service wavefronts must all be resident: ssa_env_closed_loop_f64)
            legs["closed_loop"] = closed_loop_rate(m, max(Kl, 480), Wl, propagator=args.propagator)
        else:
            legs["closed_loop"] = dict(legs["closed_loop_per_step_launches"], note="more than 20 160 objects: ssa_env_closed_loop_f64 declines "
                                       "(SSA_E_UNSUPPORTED), the closed loop runs as step + ssa_agent_select_f64 launches")
        legs["gym_api"] = {"flatten": gym_api_rate(m, 'flatten'), "aer": gym_api_rate(m, 'aer'),
                           "flatten_zero_copy": gym_api_rate(m, 'flatten', zero_copy=True),
                           "flatten_copy": gym_api_rate(m, 'flatten', obs_pool=0),
                           "flatten_f32": gym_api_rate(m, 'flatten', f32=True), "aer_f32": gym_api_rate(m, 'aer', f32=True),
                           "flatten_device_obs": gym_api_rate(m, 'flatten', obs_device=True),
                           "note": "SSA_Tasker_Env.step() per call, host in the loop, PCIe + one sync inclusive (20 000 objects).  flatten: the default "
                                   "-- a FRESH observation array per step, as the reference, WITHOUT a copy: a pinned buffer nobody holds, written by the "
                                   "kernel, taken back when the consumer drops the array (envs/_obspool.py); flatten_copy: what a consumer that keeps more "
                                   "than config['obs_pool'] = 64 observations alive gets (a 1.92 MB host copy per step; obs_pool = 0 here); "
                                   "flatten_f32 / aer_f32: config['obs_dtype'] = float32 (EXTENSION: the reference's observations are float64) -- the kernel "
                                   "writes the host-facing copy in single precision, half the bytes over PCIe; "
                                   "flatten_zero_copy: config['obs_zero_copy'] "
                                   "-- a view of the two-deep host-mapped ring the kernel writes; aer: the reference's one persistent array; "
                                   "flatten_device_obs: config['obs_device'] -- the observation stays on the GPU as a CUDA tensor (a policy that "
                                   "lives there), reward / done still cross PCIe"}
        legs["closed_loop_torch_policy"] = torch_policy_rate(m)
        if m == 20000:
            legs["vec_env"] = vec_env_rate(m)
            legs["vec_env_zero_copy"] = vec_env_rate(m, zero_copy=True)
            legs["vec_env_f32"] = vec_env_rate(m, f32=True)
            legs["vec_env_f32"]["note"] += "; config['obs_dtype'] = float32 (EXTENSION): 2.56 MB of observations per vector step instead of 5.12"
            legs["vec_env_device_obs"] = vec_env_rate(m, obs_device=True, layout=True)
            legs["vec_env_device_obs"]["note"] += ("; config['storage_layout'] = 'regime' (every env's objects stored by orbit regime, one permutation per env -- "
                                                   "HotPathEngine.set_layout([n_env][n_obj]); observations, actions and rewards in the env's own numbering, "
                                                   "bit-identical); `caller_order`: without it")
            legs["vec_env_device_obs"]["caller_order"] = vec_env_rate(m, obs_device=True)["value"]

    cpu, cpu_all, ep_fail = None, None, None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:   # reported at N=1 only
        cpu = cpu_baseline(m)
        cpu_all = cpu_baseline(m, budget_s=8.0, all_cores=True)
        if not args.no_legs:
            # behaviour over a whole episode, next to the rates: which variants lose filters the way the reference does
            ep_fail = {"workload": "%d objects, 479 round-robin steps, env defaults (alpha 1e-4, dt 20 s, an update every step), same inputs" % EPISODE_M,
                       "oracle": episode_failures_oracle(),
                       "elements": episode_failures_hip("elements"), "hybrid": episode_failures_hip("hybrid"), "fg": episode_failures_hip("fg"),
                       "note": "failed filters (status != 0) at the given steps.  oracle = CPU restatement in the reference's order of operations; "
                               "hybrid (the env default, `value`) and elements (with the reference's covariance arithmetic) are the BEHAVIOUR-FAITHFUL "
                               "variants; fg is more accurate on diverged states and its filters survive (tests/test_episode_failures.py: five workloads, "
                               "pooled counts, failed-set overlap, population, first-failure distribution)"}

    if rank == 0:
        steps_per_s = K / elapsed
        value = steps_per_s * world * (m / 20000.0)
        out = {
            "metric": "env_steps_per_sec_at_20k_objects", "value": round(value, 2),
            "unit": "env-steps/s (20 000-object UKF+propagate steps, summed over GPUs)",
            "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": round(1e3 * elapsed / K, 5),
            "repeats": repeats, "value_spread": [round(K / el_max * world * (m / 20000.0), 2), round(K / el_min * world * (m / 20000.0), 2)],
            "timing": "`repeats` timed blocks of exactly `steps` steps, each between two fences (repeated until >= 50 ms in total and at least three whole episodes); value = the median over episode-long groups of consecutive blocks of the mean block inside each group",
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "ssa_tasker_simple-v2 hot path: %d objects/GPU x %d GPU, %s + "
                                   "UKF predict (13 sigma points) + 1 az-el-range update/step%s" %
                                   (m, world, {"fg": "two-body Farnocchia (fg)", "elements": "two-body Farnocchia (elements)", "hybrid": "two-body Farnocchia (hybrid: behaviour-faithful)",
                                               "j2": "two-body + J2 RK4 propagator (EXTENSION, no reference counterpart; 4 sub-steps)"}[args.propagator],
                                    (", sharded env with one RCCL all-gather per step of %s + reward statistics" % ("per-object trace P (BASELINE config 4)" if obs_cols == 1 else "the (az,el,range,trP) observation block")) if use_dist else ""),
                       "objects_per_gpu": m, "objects_total": m * world, "alpha": 1e-4, "dt_s": 20.0,
                       "catalogue": "synthetic, drawn by the reference's recipe (envs/orbit_gen.py:30-70: regime probabilities and the visibility "
                                    "acceptance rule; ssa-gym_amd/catalogue.py): 6 806 LEO / 2 162 equatorial / 1 149 circular rows of 20 000 "
                                    "(the reference's file: 6 755 / 2 231 / 1 135), ecc <= 0.737",
                       "storage_layout": storage_layout,
                       "propagator": args.propagator, "parallelism": "object-shard x%d" % world,
                       "allgather": (("comm-stream (overlapped with the next step)" if state["overlap"] else "in-stream")
                                     if use_dist else None),
                       "allgather_api": (("direct peer stores through hipIpc-mapped pointers (ssa_peer_push_f64 / ssa_peer_wait: plain kernels, no collective; "
                                           "SSA_ALLGATHER=peer)" if sharded._peer is not None else
                                           "RCCL ncclAllGather enqueued directly in the compute/communication stream"
                                           if sharded._rccl is not None else "torch.distributed.all_gather_into_tensor")
                                         if sharded is not None else None),
                       "allgather_warmup_probe": allgather_probe,
                       "sharded_enqueue": (("hipGraph replay, %d steps per graph (episodes of whole units)" % graphed.U) if graphed is not None
                                           else ((graph_note or "per step from the host") if use_dist else None)),
                       "allgather_bytes_per_rank": (sharded.width * 8 if sharded is not None else None),
                       "rccl_ranks": (sharded._rccl.count() if (sharded is not None and sharded._rccl is not None)
                                      else (dist.get_world_size() if use_dist else None)),
                       "ukf_variant": "keep propagated sigma points for update() (default; PARITY-UNPINNED, see `resample`)",
                       "behaviour": ("%s: per-step parity within the north_star tolerance; does NOT reproduce the reference's episode-level filter "
                                     "failures (see `episode_failures`; the behaviour-faithful variants are `hybrid` -- the env default -- and `elements`)" % args.propagator
                                     if args.propagator in ("fg", "j2") else
                                     "%s: the BEHAVIOUR-FAITHFUL variant%s -- per-step parity within the north_star tolerance AND the reference's episode-level "
                                     "filter failures (`episode_failures`, tests/test_episode_failures.py over five workloads)%s"
                                     % (args.propagator, " and the env default (what `fx_xyz_farnocchia` resolves to)" if args.propagator == "hybrid" else "",
                                        ": the universal-variable series solver on strong-elliptic states, the reference's own strong-hyperbolic chain on "
                                        "diverged sigma points (the branch whose propagation error shapes the failures), universal variables on the "
                                        "near-parabolic bands in between, the prior covariance in the reference's arithmetic" if args.propagator == "hybrid" else ""))},
            "object_steps_per_sec": round(steps_per_s * m * world, 1),
            "failed_filters": n_failed,
            "roofline": roof, "cpu_baseline": cpu, "cpu_baseline_all_cores": cpu_all, "episode_failures": ep_fail, "rollout": roll,
        }
        out.update(legs)
        if rehearsal:
            out["rehearsal"] = "SSA_BENCH_REHEARSAL=1: all ranks share cuda:0, collectives over gloo -- functional check only, NOT a measurement"
        if cpu:
            out["speedup_vs_cpu_baseline"] = round(steps_per_s * world / cpu["value"], 1)
            out["speedup_vs_cpu_baseline_all_cores"] = round(steps_per_s * world / cpu_all["value"], 1)
        real_stdout.write(json.dumps(out) + "\n")
        real_stdout.flush()
    if use_dist:
        dist.barrier()
        if sharded is not None:
            sharded.close()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
