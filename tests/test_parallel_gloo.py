"""world_size-2 gloo test (CPU) of the object-sharding host logic (parallel.py): partition,
action ownership, the single all-gather per step, and the statistics reduction must reproduce
the unsharded env exactly.  The local stepper injected here is oracle-backed (tests may use the
oracle); the product injects the HIP engine."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, golden


class OracleLocalStepper:
    """LocalStepper protocol on CPU tensors: step(local_action), pack_into(obs, stats)."""

    def __init__(self, xt, x, P, ep, c2t, lo):
        import oracle as orc
        self.o, self.orc = orc.Oracle(), orc
        self.xt, self.x, self.P, self.ep, self.c2t, self.lo = xt.copy(), x.copy(), P.copy(), ep, c2t, lo
        self.status = np.zeros(len(x), dtype=np.int32)
        self.device = torch.device("cpu")
        self.tick = 0
        self.Wm, self.Wc, self.scale = orc.merwe_weights(1e-4, 2.0, -3)

    raw_shards = False     # True: hand the statistics over as raw shard words, the way the HIP step kernel's atomics leave them

    def step(self, a, obs_out=None, stats_out=None, shards_out=None, shards_clear=None, obs_cols=4):
        self.tick += 1
        self.cols = obs_cols
        ep = self.ep
        zn = ep["z_noise"][self.tick, self.lo + a] if a >= 0 else np.zeros(3)
        r = self.o.env_step(self.xt, self.x, self.P, self.status, 20.0, ep["Q"], ep["R"], self.Wm, self.Wc, self.scale, a,
                            self.c2t[self.tick], ep["obs_lla"], ep["obs_itrs"], -np.pi / 2, zn)
        self.xt, self.x, self.P, self.met = r["x_true"], r["x"], r["P"], r["metrics"]
        if obs_out is not None and shards_out is not None:   # raw form: max delta_pos as ordered bits | packed counts | failures, spread
            obs_out.copy_(torch.as_tensor(self._obs()))
            w = np.zeros((128, 16), dtype=np.uint64)             # over a few shards (one 128-byte line each) like the kernel's per-tile atomics
            d = self.met[0]
            for t0 in range(0, len(d), 4):
                sh_ = (t0 // 4) % 128
                tile = d[t0:t0 + 4]
                w[sh_, 0] = max(w[sh_, 0], tile.view(np.uint64).max())
                w[sh_, 1] += np.uint64((tile < 1e4).sum()) + (np.uint64((tile < 1e7).sum()) << np.uint64(32))
                w[sh_, 2] += np.uint64((self.status[t0:t0 + 4] != 0).sum())
            shards_out.copy_(torch.as_tensor(w.reshape(-1).view(np.float64)))
            shards_clear.zero_()
        elif obs_out is not None:
            self.pack_into(obs_out, stats_out)

    def _obs(self):
        a = self.o.aer_obs(self.x, self.P, self.c2t[self.tick], self.ep["obs_lla"], self.ep["obs_itrs"])
        return a if getattr(self, "cols", 4) == 4 else np.ascontiguousarray(a.reshape(-1, 4)[:, 3])   # (trace P alone)

    def pack_into(self, obs_out, stats_out):
        ep = self.ep
        obs_out.copy_(torch.as_tensor(self._obs()))
        d, s = self.met[0], self.met[2]
        stats_out.copy_(torch.tensor([d.max(), (d < 1e4).sum(), (d < 1e7).sum(), np.argmax(s), (self.status != 0).sum(),
                                      s.max(), 0, 0], dtype=torch.float64))


def _worker(rank, world, port, q, raw=False, cols=4):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from ssa_gym_amd import parallel
    ep = golden("episode_aer_m20_n480.npz")
    c2t = golden("c2t_2020-05-04_dt20_n480.npy")
    m = 19                                   # uneven split: 10 + 9
    plan = parallel.ShardPlan(m, world, rank)
    xt, x = ep["x_true0"][:m], ep["x0"][:m]
    P = np.tile(ep["P0"], (m, 1, 1))
    sl = slice(plan.lo, plan.hi)
    local = OracleLocalStepper(xt[sl], x[sl], P[sl], ep, c2t, plan.lo)
    local.raw_shards = raw
    sh = parallel.ShardedStepper(plan, local, obs_cols=cols)
    outs = []
    for i in range(1, 6):
        a = [3, 12, 18, 0, 9][i - 1]
        sh.step(a)
        sh.wait()
        outs.append((sh.global_obs().numpy().copy(), sh.global_stats().copy()))
        if raw:     # the fold of the raw shard words in torch (what GPU consumers use: no host round trip) equals the numpy one
            dev = sh.global_stats_device().numpy()
            ref = outs[-1][1]
            assert np.array_equal(dev[[0, 1, 2, 4]], ref[[0, 1, 2, 4]], equal_nan=True) and dev[3] == -1.0 and np.isnan(dev[5])
    if rank == 0:
        q.put(outs)
    dist.barrier()
    dist.destroy_process_group()


def test_fold_raw_statistics_semantics():
    """parallel.fold_raw_statistics: max by ordered bits with NaN on top (np.max), packed trinary counts, failures"""
    from ssa_gym_amd.parallel import fold_raw_statistics, STAT_SHARD_WORDS
    w = np.zeros((5, STAT_SHARD_WORDS), dtype=np.uint64)
    w[0, 0] = np.array([3.5e6]).view(np.uint64)[0]
    w[3, 0] = np.array([7.25e9]).view(np.uint64)[0]
    w[1, 1] = np.uint64(3) | (np.uint64(11) << np.uint64(32))
    w[4, 1] = np.uint64(2) | (np.uint64(5) << np.uint64(32))
    w[2, 2] = 4
    out = fold_raw_statistics(torch.as_tensor(w.view(np.int64))).numpy()
    assert out[0] == 7.25e9 and out[1] == 5 and out[2] == 16 and out[4] == 4 and out[3] == -1.0 and np.isnan(out[5])
    w[2, 0] = np.array([np.nan]).view(np.uint64)[0] & np.uint64(0x7fffffffffffffff)
    assert np.isnan(fold_raw_statistics(torch.as_tensor(w.view(np.int64))).numpy()[0])


def test_shard_plan():
    from ssa_gym_amd.parallel import ShardPlan
    for m, w in ((20000 * 8, 8), (19, 2), (7, 4), (3, 4)):
        plans = [ShardPlan(m, w, r) for r in range(w)]
        assert sum(p.m_local for p in plans) == m and plans[0].lo == 0 and plans[-1].hi == m
        assert max(p.m_local for p in plans) - min(p.m_local for p in plans) <= 1
        for a in (0, m // 2, m - 1):
            owners = [p.local_action(a) for p in plans]
            assert sum(o >= 0 for o in owners) == 1
            r, j = plans[0].owner(a)
            assert plans[r].lo + j == a and owners[r] == j
        assert all(p.local_action(-1) == -1 for p in plans)


@pytest.mark.parametrize("raw,cols", [(False, 4), (True, 4), (True, 1)])
def test_sharded_env_matches_unsharded_world2(raw, cols):
    """raw = True: the statistics cross the all-gather as raw shard words (what the HIP step kernel's atomics leave in the send
    buffer: no fold launch) and every rank folds all ranks' words on arrival."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from ssa_gym_amd import parallel
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + (7 if raw else 0) + (13 if cols == 1 else 0)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, raw, cols)) for r in range(2)]
    for p in procs:
        p.start()
    outs = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # unsharded reference run of the same 19-object env
    ep = golden("episode_aer_m20_n480.npz")
    c2t = golden("c2t_2020-05-04_dt20_n480.npy")
    m = 19
    plan1 = parallel.ShardPlan(m, 1, 0)
    local = OracleLocalStepper(ep["x_true0"][:m], ep["x0"][:m], np.tile(ep["P0"], (m, 1, 1)), ep, c2t, 0)
    sh = parallel.ShardedStepper(plan1, local, obs_cols=cols)
    for i, a in enumerate([3, 12, 18, 0, 9]):
        sh.step(a)
        obs, st = sh.global_obs().numpy(), sh.global_stats()
        assert np.array_equal(obs, outs[i][0]) and obs.shape == (cols * m,)   # bit-identical observation vector (cols = 1: trace P per object)
        if raw:                                         # (the raw form carries no arg-max of sigma_pos, as on the GPU)
            k = [parallel.STAT_MAX_DPOS, parallel.STAT_CNT_LT_1E4, parallel.STAT_CNT_LT_1E7, parallel.STAT_N_FAILED]
            assert np.array_equal(st[k], outs[i][1][k]) and outs[i][1][parallel.STAT_ARGMAX_SPOS] == -1
        else:
            assert np.array_equal(st, outs[i][1])       # identical reward statistics (incl. global arg-max)


def test_bench_self_launches_its_ranks_dry_run():
    """`python bench.py --gpus 2` must start its two ranks itself (parent stays GPU-free and relays rank 0's JSON line):
    rehearsed on CPU with --dry-run (gloo, payload-only stepper: launcher, rendezvous on 127.0.0.1, the sharded
    all-gather host logic; no step arithmetic, never a measurement)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry-run", "--steps", "4", "--objects", "37"],
                         env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["dry_run"] is True and rec["config"]["ranks"] == 2 and rec["config"]["payload_ok"] is True


# ---------------------------------------------------------------------------------------------------------------------------
# The PRODUCT's sharded path at world size 2 on one card: two processes, each with its own HIP engine for its shard of the
# objects on cuda:0, the payload all-gather carried by gloo (RCCL refuses two ranks on one device; the collective is the
# only piece replaced -- partition, owner-only update, raw statistics words in the payload, the step kernel writing its
# observation block straight into the send buffer are the code bench.py --gpus N runs).
def _hip_worker(rank, world, port, q, m, actions, cols, exchange="rccl", unit=0):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    try:
        import ssa_gym_amd
        from ssa_gym_amd import _lib, engine, host, parallel
        ssa_gym_amd.build()
        _lib.load()
        torch.cuda.set_device(0)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        xt, x, P, g, zn, c2t = _hip_problem(m)
        plan = parallel.ShardPlan(m, world, rank)
        sl = slice(plan.lo, plan.hi)
        consts = host.make_consts(g["Q"], g["R"], 1e-4, 2.0, -3, 20.0, -np.pi / 2, g["obs_lla"])
        eng = engine.HotPathEngine(consts, plan.m_local, 1, c2t, np.ascontiguousarray(zn[:, :, sl]), history=2)
        if exchange == "peer":      # (these cases also run with a storage layout of the rank's shard: HotPathEngine.set_layout -- the all-gather
            eng.set_layout(np.random.RandomState(40 + rank).permutation(plan.m_local))    # payload must come out in the caller's order all the same)
        eng.load_state(0, xt[sl], x[sl], P[sl])
        local = parallel.HipLocalStepper(eng, consts, fast_stats=True)
        sh = parallel.ShardedStepper(plan, local, obs_cols=cols, exchange=exchange)
        assert sh._rccl is None and sh._local_raw and (sh._peer is not None) == (exchange == "peer")
        outs = []

        def record():
            sh.wait()
            torch.cuda.synchronize()
            if sh._peer is not None:
                sh._peer.check()
            dev = sh.global_stats_device().cpu().numpy()
            st = sh.global_stats()
            assert np.array_equal(dev[[0, 1, 2, 4]], st[[0, 1, 2, 4]], equal_nan=True)
            outs.append((sh.global_obs().cpu().numpy(), st))
        if unit:          # whole units replayed from a captured hipGraph (the peer-store exchange is plain kernels: capturable at world 2)
            gs = parallel.GraphedShardedSteps(sh, unit, actions)
            gs.rewind()
            for _ in range(len(actions) // unit):
                gs.run_unit()
                record()
            assert gs.capture_failed is None, gs.capture_failed
        else:
            for k, a in enumerate(actions):
                sh.step(a, overlap=(k % 3 == 2))
                record()
        eng.to_caller_order()
        t = local.tick % 2
        state = (eng.x_true[t].cpu().numpy(), eng.x_filter[t].cpu().numpy(), eng.P_filter[t].cpu().numpy(), eng.status.cpu().numpy())
        q.put((rank, outs if rank == 0 else None, state))
        dist.barrier()
        sh.close()
        dist.destroy_process_group()
    except Exception as exc:   # (the parent must not wait for a queue item that never comes)
        import traceback
        q.put((rank, "error: %s\n%s" % (exc, traceback.format_exc()), None))
        raise


def _hip_problem(m):
    rs = np.random.RandomState(11)
    cat = golden("catalogue_subset.npy")
    g = golden("ukf_step_golden.npz")
    xt = cat[rs.randint(0, len(cat), m)]
    x = xt + rs.normal(size=(m, 6)) * np.array([1e5] * 3 + [1e2] * 3)
    P = np.tile(g["P0"], (m, 1, 1))
    zn = rs.normal(size=(1, 480, m, 3)) * np.array([4.8e-6, 4.8e-6, 1e3])
    return xt, x, P, g, zn, golden("c2t_2020-05-04_dt20_n480.npy")


@pytest.mark.gpu
@pytest.mark.parametrize("cols,exchange,unit", [(4, "rccl", 0), (1, "rccl", 0), (4, "peer", 0), (1, "peer", 0), (1, "peer", 4)])
def test_sharded_hip_env_world2_shares_one_gpu(cols, exchange, unit):
    """1 003 objects split 502 + 501 over two ranks (two processes on cuda:0), 8 steps whose actions land in both shards:
    the gathered observation vector, the folded statistics and every rank's slice of the state equal the unsharded HIP
    engine's BIT FOR BIT (an object's arithmetic does not depend on which wavefront or rank holds it).
    exchange: 'rccl' = the collective (carried by gloo here: RCCL refuses two ranks on one device); 'peer' = the all-gather by DIRECT
    PEER STORES (peer.py, ssa_peer_push_f64 / ssa_peer_wait): each rank writes its payload into the other's arena through a pointer
    mapped over hipIpc and raises a flag, nothing of it is a collective -- per step from the host, and (unit = 4) as whole units
    replayed from a captured hipGraph at world size 2."""
    import ssa_gym_amd
    from ssa_gym_amd import _lib, engine, host, parallel
    ssa_gym_amd.build()
    _lib.load()
    assert torch.cuda.is_available()
    m, world = 1003, 2
    actions = [3, 700, 501, 502, 17, 1002, 0, 640]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000) + (5 if cols == 1 else 0) + (11 if exchange == "peer" else 0) + unit
    procs = [ctx.Process(target=_hip_worker, args=(r, world, port, q, m, actions, cols, exchange, unit)) for r in range(world)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(world):
        rank, outs, state = q.get(timeout=300)
        assert state is not None, outs
        got[rank] = (outs, state)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    # the unsharded run of the same env on this process's engine
    xt, x, P, g, zn, c2t = _hip_problem(m)
    consts = host.make_consts(g["Q"], g["R"], 1e-4, 2.0, -3, 20.0, -np.pi / 2, g["obs_lla"])
    eng = engine.HotPathEngine(consts, m, 1, c2t, zn, history=2)
    eng.load_state(0, xt, x, P)
    local = parallel.HipLocalStepper(eng, consts, fast_stats=True)
    sh = parallel.ShardedStepper(parallel.ShardPlan(m, 1, 0), local, obs_cols=cols)
    k_st = [parallel.STAT_MAX_DPOS, parallel.STAT_CNT_LT_1E4, parallel.STAT_CNT_LT_1E7, parallel.STAT_N_FAILED]
    for k, a in enumerate(actions):
        sh.step(a)
        torch.cuda.synchronize()
        if unit and (k + 1) % unit:
            continue              # (the graphed run recorded the last step of every unit)
        j = (k + 1) // unit - 1 if unit else k
        obs, st = sh.global_obs().cpu().numpy(), sh.global_stats()
        assert obs.shape == (cols * m,) and np.array_equal(obs, got[0][0][j][0], equal_nan=True), k
        assert np.array_equal(st[k_st], got[0][0][j][1][k_st], equal_nan=True), k
    t = local.tick % 2
    full = (eng.x_true[t].cpu().numpy(), eng.x_filter[t].cpu().numpy(), eng.P_filter[t].cpu().numpy(), eng.status.cpu().numpy())
    for r in range(world):
        plan = parallel.ShardPlan(m, world, r)
        for a, b in zip(full, got[r][1]):
            assert np.array_equal(a[plan.lo:plan.hi], b, equal_nan=True), r
    # the updates happened (both shards): the selected objects' covariances shrank below the propagated prior's
    trP = np.trace(full[2], axis1=1, axis2=2)
    assert np.all(trP[[3, 700, 501, 502]] < 0.5 * np.median(trP))


@pytest.mark.gpu
@pytest.mark.parametrize("extra", [["--steps", "20", "--warmup", "5"], ["--steps", "40", "--warmup", "40", "--payload", "aer"],
                                   ["--steps", "20", "--warmup", "5", "peer"]])
def test_bench_two_ranks_rehearsal_on_one_gpu(extra):
    """`python bench.py --gpus 2` end to end with the HIP engine in both ranks (SSA_BENCH_REHEARSAL=1: the ranks share cuda:0 and
    gloo carries the collectives -- the launcher, the warm-up probe of where the all-gather runs, the timed blocks with MAX over
    ranks, the roofline leg on rank 0 while rank 1 waits, the closing barrier).  The rate it prints is NOT a measurement."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env["SSA_BENCH_REHEARSAL"] = "1"
    peer = extra[-1] == "peer"          # the all-gather by direct peer stores, its units replayed from a hipGraph at world size 2
    if peer:
        extra = extra[:-1]
        env["SSA_ALLGATHER"] = "peer"
    # (`fg`: the variant whose filters survive whole episodes -- how many timed blocks a run takes, and so how far into an episode it gets,
    # depends on the box; the behaviour-faithful default loses filters late in an episode by design and would make `failed_filters` a coin)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--objects", "4000", "--propagator", "fg"] + extra,
                         env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and "rehearsal" in rec and rec["failed_filters"] == 0 and rec["value"] > 0
    assert rec["config"]["objects_total"] == 8000 and rec["config"]["rccl_ranks"] == 2
    assert rec["roofline"]["kernel_ms"] > 0
    if peer:
        assert rec["config"]["allgather_api"].startswith("direct peer stores") and rec["config"]["sharded_enqueue"].startswith("hipGraph replay")
    else:
        assert rec["config"]["allgather_api"] == "torch.distributed.all_gather_into_tensor"
