"""Object-axis sharding of one environment across the GPUs of a node.

Propagation and predict are independent per object and the single update touches only the
object the action names, so the catalogue splits into contiguous blocks, one per rank; the only
exchange is the per-step reassembly of the global observation vector plus the reward
reductions.  Both travel in ONE all-gather per step (RCCL over xGMI through
torch.distributed's "nccl" backend; "gloo" in the CPU tests): each rank contributes

    [ aer_obs (4 doubles per local object: az, el, range, trace P -- ssa_tasker_simple_2.py:834; or, with
      obs_cols = 1, trace P alone: the "per-object covariance-trace observation" of BASELINE config 4) |
      8 reward statistics (include/ssa_hip.h SSA_STAT_*) ]

and every rank then holds the whole (4 * m_total) observation vector and reduces the statistics
locally (max / sum / arg-max with global indices).  Messages are small (0.64 MB per rank at
20 000 objects), so a single fully-connected all-gather over the 7 point-to-point xGMI links is
the right collective; a separate all-reduce would only add a second launch latency.

The local compute engine is injected (`LocalStepper` protocol), so that this host logic is
exercised by world_size-2 gloo tests on CPU with an oracle-backed stepper; the product always
injects the HIP engine (there is no CPU fallback in the product).
"""
import numpy as np
import torch
import torch.distributed as dist

STAT_STRIDE = 8
STAT_SHARDS, STAT_SHARD_WORDS = 128, 16   # include/ssa_hip.h: raw statistics shards [128][16] uint64 per env (one 128-byte line each, words 0..2 used)
RAW_WORDS = STAT_SHARDS * STAT_SHARD_WORDS
STAT_MAX_DPOS, STAT_CNT_LT_1E4, STAT_CNT_LT_1E7, STAT_ARGMAX_SPOS, STAT_N_FAILED, STAT_MAX_SPOS = range(6)


class ShardPlan:
    """contiguous block partition of m_total objects over `world` ranks (sizes differ by <= 1)."""

    def __init__(self, m_total, world, rank):
        self.m_total, self.world, self.rank = int(m_total), int(world), int(rank)
        base, rem = divmod(self.m_total, self.world)
        self.sizes = [base + (1 if r < rem else 0) for r in range(self.world)]
        self.offsets = np.concatenate([[0], np.cumsum(self.sizes)]).astype(np.int64)
        self.lo, self.hi = int(self.offsets[rank]), int(self.offsets[rank + 1])
        self.m_local = self.hi - self.lo
        self.m_pad = max(self.sizes)   # all_gather_into_tensor needs equal contributions

    def owner(self, a):
        """global object index -> (rank, local index)"""
        r = int(np.searchsorted(self.offsets, a, side="right") - 1)
        return r, int(a - self.offsets[r])

    def local_action(self, a):
        """the action as seen by this rank: local index if it owns object a, else -1 (no update)."""
        if a is None or a < 0:
            return -1
        r, j = self.owner(a)
        return j if r == self.rank else -1


def _direct_communicator(dev, group):
    """rccl.Communicator for `group`, or None -- decided COLLECTIVELY: the ranks agree (all-reduce MIN of a success
    flag over the process group) before any of them uses it, so that a rank on which librccl cannot be loaded or
    ncclCommInitRank fails never leaves the others issuing ncclAllGather on a private communicator while it calls
    all_gather_into_tensor on the process group (mismatched collectives hang).  Every rank takes part in the unique-id
    broadcast inside rccl.Communicator whether or not its own library load succeeded."""
    import warnings
    from . import rccl
    comm, err = None, None
    try:
        comm = rccl.Communicator(dev, group)
    except Exception as exc:  # noqa: BLE001
        err = exc
    ok = torch.tensor([1 if comm is not None else 0], dtype=torch.int32, device=dev)
    dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)
    if int(ok.item()) == 1:
        return comm
    if comm is not None:
        comm.close()
    warnings.warn("direct RCCL communicator unavailable on at least one rank (%s); every rank uses "
                  "torch.distributed.all_gather_into_tensor" % (err if err is not None else "another rank failed"))
    return None


def _peer_exchange(plan, width, dev, group, use_dist):
    """the peer-store exchange, or None on EVERY rank when any rank could not map its peers (the caller then takes the collective);
    every rank takes part in the handle exchange inside peer.PeerExchange whatever happens to it afterwards."""
    import warnings
    from . import peer
    px, err = None, None
    try:
        px = peer.PeerExchange(plan.world if use_dist else 1, plan.rank if use_dist else 0, width, dev, group)
    except Exception as exc:  # noqa: BLE001
        err = exc
    if not use_dist:
        if px is None:
            raise err
        return px
    ok = torch.tensor([1 if px is not None else 0], dtype=torch.int32, device=dev if dist.get_backend(group) == "nccl" else "cpu")
    dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)
    if int(ok.item()) == 1:
        return px
    if px is not None:
        px.close()
    warnings.warn("peer-store all-gather unavailable on at least one rank (%s); every rank uses the collective"
                  % (err if err is not None else "another rank failed"))
    return None


class ShardedStepper:
    """one env of m_total objects, one rank per shard.

    Three send/receive buffer pairs rotate.  With `overlap=True` (GPU) the all-gather of step k is
    issued on a dedicated communication stream behind an event, so RCCL moves step k's payload over
    xGMI while the compute stream already runs the kernels of step k+1; the compute stream only waits
    (on an event, not on the host) before it touches a payload buffer again: step k accumulates into
    buffer k % 3 and zeroes the statistics words of buffer (k + 1) % 3, whose all-gather was issued at
    step k - 2 -- one whole step of slack, so the raw-statistics form (no fold launch) works in both modes."""

    NB = 3

    def __init__(self, plan, local, group=None, direct_rccl=True, obs_cols=4, exchange=None):
        assert obs_cols in (1, 4)
        self.plan, self.local, self.group, self.cols = plan, local, group, int(obs_cols)
        self._rccl = None
        # exchange: 'rccl' (the all-gather collective: default) | 'peer' (direct peer stores over hipIpc-mapped pointers, peer.py: plain
        # kernels, no collective -- GPU only; also the environment variable SSA_ALLGATHER).  Every rank must ask for the same one.
        import os
        exchange = exchange or os.environ.get("SSA_ALLGATHER") or "rccl"
        if exchange not in ("rccl", "peer"):
            raise ValueError("ShardedStepper: exchange must be 'rccl' or 'peer', not %r" % (exchange,))
        self._peer, self._waited = None, 0
        dev = local.device
        # payload of one rank: [ aer block 4 * m_pad | 8 folded statistics | 256 raw statistics words (uint64 bit patterns) ]
        # A local stepper with `raw_shards` (the HIP engine) lets the step kernel accumulate straight into the raw words of
        # the send buffer -- no fold launch -- and every rank folds all ranks' words on arrival; other steppers fill the 8
        # folded statistics.  (The step kernel of step k also zeroes the raw words of the NEXT send buffer, whose last
        # all-gather -- two steps ago -- has completed: in stream order, or behind the event the compute stream waits on.)
        self.width = self.cols * plan.m_pad + STAT_STRIDE + RAW_WORDS
        self.send = [torch.zeros(self.width, dtype=torch.float64, device=dev) for _ in range(self.NB)]
        self.recv = [torch.zeros(plan.world * self.width, dtype=torch.float64, device=dev) for _ in range(self.NB)]
        self._raw = [False] * self.NB
        self.k = 0
        # the slices of the payload buffers the local stepper writes into, built once (a tensor slice costs the host 1-2 us:
        # at 13 us of GPU work per step the per-step host path decides whether the collective can hide behind the compute)
        o_st = self.cols * plan.m_pad
        self._v_obs = [s[:self.cols * plan.m_local] for s in self.send]
        self._v_stats = [s[o_st:o_st + STAT_STRIDE] for s in self.send]
        self._v_shards = [s[o_st + STAT_STRIDE:] for s in self.send]
        self._kw = {} if self.cols == 4 else {"obs_cols": self.cols}
        self._use_dist = dist.is_available() and dist.is_initialized()
        self._local_raw = bool(getattr(local, "raw_shards", False))
        self._gpu = torch.device(dev).type == "cuda"
        if self._gpu:
            self.comm = torch.cuda.Stream(device=dev)
            self._ready = [torch.cuda.Event() for _ in range(self.NB)]
            self._done = [torch.cuda.Event() for _ in range(self.NB)]
            self._pending = [False] * self.NB
            # the all-gather enqueued by RCCL itself in our stream (no ProcessGroup stream hops, rccl.py); the
            # torch.distributed collective remains the fallback
            if exchange == "peer":
                self._peer = _peer_exchange(plan, self.width, torch.device(dev), group, self._use_dist)
            if self._peer is not None:
                self.recv = self._peer.recv          # (the receive buffers ARE the rank's arena the peers store into)
            elif direct_rccl and dist.is_available() and dist.is_initialized() and dist.get_backend(group) == "nccl":
                self._rccl = _direct_communicator(torch.device(dev), group)
        elif exchange == "peer":
            raise ValueError("ShardedStepper: the peer-store exchange moves device memory (HIP stepper on a GPU)")

    def _all_gather(self, recv, send, stream):
        if self._rccl is not None:
            self._rccl.all_gather_f64(send.data_ptr(), recv.data_ptr(), self.width, stream.cuda_stream)
        else:
            dist.all_gather_into_tensor(recv, send, group=self.group)

    def step(self, global_action, overlap=False):
        p = self.plan
        b, bn = self.k % self.NB, (self.k + 1) % self.NB
        send, recv = self.send[b], self.recv[b]
        use_dist = self._use_dist
        overlap = overlap and self._gpu and use_dist
        cur = torch.cuda.current_stream() if self._gpu else None
        if self._gpu:
            for q in (b, bn):               # buffers this step writes (payload) or zeroes (next buffer's statistics words):
                if self._pending[q]:        # their last all-gather (three / two steps ago) must have left
                    cur.wait_event(self._done[q])
                    self._pending[q] = False
        # kernels of this step write the observation block and the statistics straight into `send`
        raw = self._local_raw
        kw = dict(self._kw, stream=cur.cuda_stream) if self._gpu else self._kw   # (the stream handle, looked up once per step)
        if raw:
            self.local.step(p.local_action(global_action), self._v_obs[b], None,
                            shards_out=self._v_shards[b], shards_clear=self._v_shards[bn], **kw)
        else:
            self.local.step(p.local_action(global_action), self._v_obs[b], self._v_stats[b], **kw)
        self._raw[b] = raw
        if self._peer is not None:
            # in the step's own stream, ONE launch: per peer, wait until its payload of the PREVIOUS step has arrived (normally there already;
            # it orders this push behind that peer's readers of the slot it overwrites, peer.py), then store this step's payload into its arena
            self._peer.push(send, b, self.k + 1, cur.cuda_stream, after=((self.k - 1) % self.NB) if self.k >= 1 else None)
        elif not use_dist:   # single process without a process group
            recv.copy_(send)
        elif overlap:
            self._ready[b].record(cur)
            self.comm.wait_event(self._ready[b])
            if self._rccl is not None:     # (enqueued into the communication stream by handle: no current-stream switch on the host)
                self._all_gather(recv, send, self.comm)
                self._done[b].record(self.comm)
            else:
                with torch.cuda.stream(self.comm):
                    self._all_gather(recv, send, self.comm)
                    self._done[b].record(self.comm)
            self._pending[b] = True
        elif self._gpu:
            self._all_gather(recv, send, cur)
        else:
            dist.all_gather_into_tensor(recv, send, group=self.group)
        self.k += 1

    def close(self):
        """release the direct RCCL communicator (after the last collective has completed)."""
        if self._rccl is not None:
            torch.cuda.synchronize()
            self._rccl.close()
            self._rccl = None
        if self._peer is not None:
            self._peer.close()
            self._peer = None

    def wait(self):
        """make the current stream wait for every all-gather issued so far."""
        if self._peer is not None:
            if self.k >= 1 and self._waited < self.k:
                self._peer.wait((self.k - 1) % self.NB, self.k, torch.cuda.current_stream().cuda_stream)
                self._waited = self.k
            return
        if self._gpu:
            for b in range(self.NB):
                if self._pending[b]:
                    torch.cuda.current_stream().wait_event(self._done[b])
                    self._pending[b] = False

    # ---- views of the reassembled global state of the latest step (call after wait())
    def _latest(self):
        return self.recv[(self.k - 1) % self.NB]

    def global_obs(self):
        p = self.plan
        rows = self._latest().view(p.world, self.width)
        return torch.cat([rows[r, :self.cols * p.sizes[r]] for r in range(p.world)])

    def global_stats_device(self):
        """the global statistics of the latest step as a DEVICE tensor [8], folded on the device (no host synchronisation: what a
        consumer in the stream -- a device-side agent, the next step's bookkeeping -- reads).  Raw-shard payloads only."""
        p = self.plan
        if not self._raw[(self.k - 1) % self.NB]:
            raise RuntimeError("global_stats_device: the latest payload carries folded statistics (CPU stepper); use global_stats()")
        rows = self._latest().view(p.world, self.width)[:, self.cols * p.m_pad + STAT_STRIDE:]
        w = rows.contiguous().view(torch.int64).view(p.world * STAT_SHARDS, STAT_SHARD_WORDS)
        return fold_raw_statistics(w)

    def global_stats(self):
        """reduce the per-rank statistics exactly as the single-GPU kernel would have produced them."""
        p = self.plan
        rows = self._latest().view(p.world, self.width)[:, self.cols * p.m_pad:].cpu().numpy()
        st = rows[:, :STAT_STRIDE]
        out = np.zeros(STAT_STRIDE)
        if self._raw[(self.k - 1) % self.NB]:      # raw shard words of every rank: fold them as reward_fold_kernel would
            w = np.ascontiguousarray(rows[:, STAT_STRIDE:]).view(np.uint64).reshape(p.world * STAT_SHARDS, STAT_SHARD_WORDS)[:, :4]
            out[STAT_MAX_DPOS] = np.array([w[:, 0].max()], dtype=np.uint64).view(np.float64)[0]   # ordered bit patterns, NaN on top
            out[STAT_CNT_LT_1E4] = float((w[:, 1] & np.uint64(0xffffffff)).sum())
            out[STAT_CNT_LT_1E7] = float((w[:, 1] >> np.uint64(32)).sum())
            out[STAT_N_FAILED] = float(w[:, 2].sum())
            out[STAT_MAX_SPOS], out[STAT_ARGMAX_SPOS] = np.nan, -1.0
            return out
        out[STAT_MAX_DPOS] = np.nan if np.isnan(st[:, STAT_MAX_DPOS]).any() else st[:, STAT_MAX_DPOS].max()
        out[STAT_CNT_LT_1E4] = st[:, STAT_CNT_LT_1E4].sum()
        out[STAT_CNT_LT_1E7] = st[:, STAT_CNT_LT_1E7].sum()
        out[STAT_N_FAILED] = st[:, STAT_N_FAILED].sum()
        sp = st[:, STAT_MAX_SPOS]
        if (st[:, STAT_ARGMAX_SPOS] < 0).any():      # atomics statistics path: arg-max of sigma_pos not computed
            out[STAT_MAX_SPOS], out[STAT_ARGMAX_SPOS] = np.nan, -1.0
            return out
        if np.isnan(sp).any():                       # np.argmax: first NaN wins
            r = int(np.where(np.isnan(sp))[0][0])
        else:
            r = int(np.argmax(sp))                   # first rank holding the maximum = lowest global index
        out[STAT_MAX_SPOS] = sp[r]
        out[STAT_ARGMAX_SPOS] = p.offsets[r] + st[r, STAT_ARGMAX_SPOS]
        return out


class GraphedShardedSteps:
    """U consecutive sharded steps -- U x [step kernel, all-gather] -- captured ONCE into a hipGraph and replayed.

    Enqueueing a sharded step costs the host 13-14 us (parameter block, two ctypes calls, event bookkeeping): as much as the
    step kernel runs, so the eager loop is host-bound, and with the all-gather overlapped on a communication stream (two
    cross-stream event waits per step) the host needs 34 us per step.  Replaying a captured unit costs one launch per U steps.
    Everything a step reads that changes from replay to replay lives in device memory the graph itself advances:
      * time: the kernels read env_time0 (device) + their position in the unit; the unit's last node adds U to it;
      * actions: a window of U action words, gathered at the unit's start from the rank's (cyclic) schedule at a device-side
        cursor that the unit then advances.
    History slots and payload buffers are addressed by position (slot = (phase + j) % 2, buffer = j % 3), so a unit is captured
    per starting phase (at most 6 graphs, captured on first use).  overlap: the all-gather of step j runs on the communication
    stream while the step kernel of step j + 1 runs -- a fork / join inside the captured graph.
    The caller keeps the episode bookkeeping: reset the env between units (`rewind()` sets the time index back to 0)."""

    def __init__(self, sharded, unit, schedule_global, overlap=False):
        self.sh, self.U, self.overlap = sharded, int(unit), bool(overlap)
        self.local = sharded.local
        self.eng = self.local.engine
        if self.eng.E != 1 or self.eng.H != 2:
            raise ValueError("graphed sharded steps: one env, history depth 2")
        if not self.sh._gpu or not getattr(self.local, "raw_shards", False):
            raise ValueError("graphed sharded steps need the HIP stepper")
        if self.sh._use_dist and self.sh._rccl is None and self.sh._peer is None:
            # (a ProcessGroup collective is not captured here: gloo's runs on the host, and a capture that fails half way leaves
            # the backend's internal streams in capture mode -- refuse BEFORE anything is captured, the caller enqueues per step)
            raise ValueError("graphed sharded steps need the direct RCCL all-gather (rccl.py); this group has none")
        plan = sharded.plan
        sched = np.asarray([plan.local_action(int(a)) for a in schedule_global], dtype=np.int32)
        dev = self.eng.dev
        self.sched = torch.as_tensor(sched).to(dev)                      # the rank's local action per global step (cyclic)
        self.cursor = torch.zeros(1, dtype=torch.int64, device=dev)
        self._ar = torch.arange(self.U, dtype=torch.int64, device=dev)
        self._idx = torch.zeros(self.U, dtype=torch.int64, device=dev)
        self.window = torch.full((self.U,), -1, dtype=torch.int32, device=dev)
        self._graphs = {}
        self.capture_failed = None          # repr of the exception that made a phase fall back to the eager enqueue
        self._force_capture_failure = False
        self.eng.env_time0.zero_()
        self._stream = torch.cuda.Stream(device=dev)                     # capture / replay stream

    def rewind(self):
        """new episode: the next unit starts at time index 1 (the caller has restored the state)"""
        self.eng.flush_stats()
        self.eng.env_time0.zero_()

    def _enqueue_unit(self, tick0, k0):
        sh, local, eng, U = self.sh, self.local, self.eng, self.U
        cur = torch.cuda.current_stream()
        n = self.sched.numel()
        torch.remainder(self.cursor + self._ar, n, out=self._idx)
        torch.index_select(self.sched, 0, self._idx, out=self.window)
        wptr = self.window.data_ptr()
        done = []
        for j in range(U):
            b, bn = (k0 + j) % sh.NB, (k0 + j + 1) % sh.NB
            if self.overlap and j >= 2:
                cur.wait_event(done[j - 2])          # buffer bn's all-gather (two steps ago) has left before this step zeroes its words
            t = tick0 + j + 1
            eng.launch_step((t - 1) % 2, t % 2, j + 1, actions_ptr=wptr + 4 * j, aer_out=sh._v_obs[b].data_ptr(), fast_stats=True,
                            shards_out=sh._v_shards[b].data_ptr(), shards_clear=sh._v_shards[bn].data_ptr(), aer_cols=sh.cols,
                            stream=cur.cuda_stream)
            sh._raw[b] = True
            if sh._peer is not None:      # plain kernels in the unit's stream; step numbers relative to the device-side base
                sh._peer.push(sh.send[b], b, k0 + j + 1, cur.cuda_stream, after=((k0 + j - 1) % sh.NB) if k0 + j >= 1 else None)
            elif self.overlap:
                ev = torch.cuda.Event()
                ev.record(cur)
                sh.comm.wait_event(ev)
                sh._all_gather(sh.recv[b], sh.send[b], sh.comm)
                dn = torch.cuda.Event()
                dn.record(sh.comm)
                done.append(dn)
            else:
                sh._all_gather(sh.recv[b], sh.send[b], cur)
        if self.overlap and sh._peer is None:
            for dn in done[-2:]:
                cur.wait_event(dn)                   # join: the unit ends when its last all-gathers have
        eng.env_time0.add_(U)
        self.cursor.add_(U)
        if sh._peer is not None:
            sh._peer.advance_on_device(U)

    def _end_stray_capture(self):
        """after a failed capture: no stream of ours may be left in capture mode (hipStreamIsCapturing / hipStreamEndCapture)"""
        import ctypes
        try:
            hip = ctypes.CDLL("libamdhip64.so")
        except OSError:
            return
        for st in (self._stream, self.sh.comm):
            status = ctypes.c_int(0)
            if hip.hipStreamIsCapturing(ctypes.c_void_p(st.cuda_stream), ctypes.byref(status)) == 0 and status.value != 0:
                graph = ctypes.c_void_p()
                hip.hipStreamEndCapture(ctypes.c_void_p(st.cuda_stream), ctypes.byref(graph))
                if graph.value:
                    hip.hipGraphDestroy(graph)
        hip.hipGetLastError()     # (the sticky error of the failed capture is consumed here, not by the next launch)

    def run_unit(self):
        """replay (capture on first use) the unit for the current phase; returns nothing, synchronises nothing.  A phase whose capture
        fails (a runtime that cannot capture the collective) is remembered and enqueued eagerly from then on -- the same kernels and the
        same collectives in the same order, so ranks that captured and ranks that did not stay in step; `capture_failed` says why."""
        sh, local = self.sh, self.local
        tick0, k0 = local.tick, sh.k
        key = (tick0 % 2, k0 % sh.NB)
        g = self._graphs.get(key)
        cur = torch.cuda.current_stream()
        # (host bookkeeping first: whatever happens below, U steps of this rank's state and U all-gathers have been / will be enqueued)
        local.tick += self.U
        sh.k += self.U
        try:
            self._run_unit(g, key, cur, tick0, k0)
        finally:
            if sh._peer is not None:       # (the device-side base of the step numbers moved by U: the host's copy follows)
                sh._peer.advanced(self.U)

    def _run_unit(self, g, key, cur, tick0, k0):
        sh = self.sh
        if g is False:                 # this phase could not be captured: per-step enqueue
            self._enqueue_unit(tick0, k0)
            return
        if g is not None:
            g.replay()
            return
        # first use of this phase: the unit runs ONCE eagerly on the capture stream (RCCL sets up its channels on first use,
        # and this run IS the unit the caller asked for), then the same enqueue sequence is captured for the replays to
        # come -- the capture itself executes nothing
        self._stream.wait_stream(cur)
        with torch.cuda.stream(self._stream):
            self._enqueue_unit(tick0, k0)
            self._stream.synchronize()
        # (begin / end by hand instead of `with torch.cuda.graph(...)`: when the capture fails -- a collective that cannot be
        # captured -- the context manager's exit raises from capture_end() before it restores the current stream, and a stream
        # left capturing makes the next allocation or copy of the process fail.  Here a failed capture is ended, every stream
        # is checked, the current stream is restored, and the phase falls back to the eager enqueue.)
        g = torch.cuda.CUDAGraph()
        import gc
        gc.collect()               # (no collection inside the capture: a CUDAGraph finalised there is destroyed while a stream captures -- not
        torch.cuda.synchronize()   # permitted, and an error thrown from that destructor ends the process)
        gc_was_on = gc.isenabled()
        gc.disable()
        try:
            with torch.cuda.stream(self._stream):
                g.capture_begin(capture_error_mode="thread_local")
                try:
                    if self._force_capture_failure:
                        raise RuntimeError("forced capture failure (test)")
                    self._enqueue_unit(tick0, k0)
                    g.capture_end()
                except BaseException:
                    try:
                        g.capture_end()
                    except Exception:  # noqa: BLE001  (an invalidated capture reports its error again here)
                        pass
                    self._end_stray_capture()
                    raise
            self._graphs[key] = g
        except Exception as exc:  # noqa: BLE001
            self._graphs[key] = False
            self.capture_failed = repr(exc)
            import ctypes
            ctypes.pythonapi.Py_IncRef(ctypes.py_object(g))     # (a graph whose capture failed is never finalised: its destructor aborts the
            #                                                      process in this torch build -- envs/ssa_tasker_simple_2.py::_never_destroy)
            # (what the aborted capture recorded of the device-side advances never ran; the eager unit above did advance them once: consistent)
        finally:
            if gc_was_on:
                gc.enable()
        cur.wait_stream(self._stream)


def fold_raw_statistics(rows_words):
    """device-side fold of raw statistics shard words [n_shards][>= 3] (int64 bit patterns) -> float64[8] on the same device,
    as reward_fold_kernel / ShardedStepper.global_stats() do: max delta_pos by its ordered bits (NaN on top), the two trinary
    counts, the failures; arg-max of sigma_pos is not part of the raw form (-1 / NaN)."""
    w = rows_words
    out = torch.zeros(STAT_STRIDE, dtype=torch.float64, device=w.device)
    out[STAT_MAX_DPOS] = w[:, 0].max().view(torch.float64)          # (non-negative doubles and NaN order like integers)
    out[STAT_CNT_LT_1E4] = (w[:, 1] & 0xffffffff).sum().to(torch.float64)
    out[STAT_CNT_LT_1E7] = (w[:, 1] >> 32).sum().to(torch.float64)
    out[STAT_N_FAILED] = w[:, 2].sum().to(torch.float64)
    out[STAT_ARGMAX_SPOS] = -1.0
    out[STAT_MAX_SPOS] = float("nan")
    return out


class HipLocalStepper:
    """LocalStepper over the HIP engine (the only one the product uses)."""

    def __init__(self, engine, consts, fast_stats=False, defer_fold=False, fold_inside=False):
        self.engine, self.consts = engine, consts
        # fold_inside (with fast_stats, instead of defer_fold): every step's statistics are folded by that step's own last
        # wavefronts (SSA_LAUNCH_FOLD_INSIDE) -- one launch per step and nothing left to flush
        self.fold_inside = bool(fold_inside) and bool(fast_stats)
        self.fast_stats = fast_stats   # statistics by the common-path kernel's atomics: two launches, no arg-max of sigma_pos
        # one launch per step: the statistics of step k are folded by extra wavefronts of step k+1 (flush() folds the
        # last one).  For consumers that read the statistics in bulk; a closed loop reads them every step (no deferral).
        self.defer_fold = defer_fold and not self.fold_inside
        self.device = engine.dev
        self.raw_shards = bool(fast_stats)   # the step kernel can accumulate into caller-provided shard words (no fold launch)
        self.tick = 0
        self._act = torch.zeros(1, dtype=torch.int32)
        self._sched = None

    def load_schedule(self, local_actions):
        """pre-stage the local action of every coming step in HBM (no per-step host->device copy)."""
        self._sched = torch.as_tensor(np.asarray(local_actions, dtype=np.int32)).to(self.device)
        self._sched_k0 = self.tick

    def flush(self):
        """fold the statistics of the last deferred step."""
        self.engine.flush_stats()

    def step(self, local_action, obs_out=None, stats_out=None, profile_slot=None, shards_out=None, shards_clear=None, obs_cols=4,
             stream=None):
        """enqueue one env step; when given, the post kernel writes the shard's aer observation
        block and its reward statistics directly into `obs_out` / `stats_out` (the all-gather payload)."""
        e = self.engine
        self.tick += 1
        aer = obs_out.data_ptr() if obs_out is not None else 0
        st = stats_out.data_ptr() if stats_out is not None else 0
        so = shards_out.data_ptr() if shards_out is not None else 0
        sc = shards_clear.data_ptr() if shards_clear is not None else 0
        if self._sched is not None:
            k = (self.tick - 1 - self._sched_k0) % self._sched.numel()
            e.launch_step((self.tick - 1) % e.H, self.tick % e.H, self.tick, actions_ptr=self._sched.data_ptr() + 4 * k,
                          aer_out=aer, stats_out=st, fast_stats=self.fast_stats, defer_fold=self.defer_fold and stats_out is None,
                          profile_slot=profile_slot, shards_out=so, shards_clear=sc, aer_cols=obs_cols, stream=stream,
                          fold_inside=self.fold_inside)
            return
        self._act[0] = int(local_action)
        e.actions.copy_(self._act)
        e.launch_step((self.tick - 1) % e.H, self.tick % e.H, self.tick, aer_out=aer, stats_out=st, fast_stats=self.fast_stats,
                      defer_fold=self.defer_fold and stats_out is None, profile_slot=profile_slot, shards_out=so, shards_clear=sc,
                      aer_cols=obs_cols, stream=stream, fold_inside=self.fold_inside)

    def rollout(self, n_steps):
        """advance n_steps of the pre-staged schedule in ONE launch (open-loop actions; HotPathEngine.launch_rollout)."""
        e = self.engine
        if self._sched is None:
            raise RuntimeError("rollout needs load_schedule()")
        while n_steps > 0:   # (the staged schedule is cyclic: split a launch that would run past its end)
            k0 = (self.tick - self._sched_k0) % self._sched.numel()
            n = min(n_steps, self._sched.numel() - k0)
            e.launch_rollout(self.tick % e.H, self.tick + 1, self._sched[k0:k0 + n].view(n, 1))
            self.tick += n
            n_steps -= n

    def reset_episode(self, snap, episode_len):
        """start a new episode from a device-resident snapshot: the next step gets time index 1."""
        e = self.engine
        e.flush_stats()                                  # the last step of the finished episode
        self.tick += (-self.tick) % episode_len          # advance to the next multiple of the episode length
        e.restore(self.tick % e.H, snap)
