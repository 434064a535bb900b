"""All-gather of the sharded step by DIRECT PEER STORES (SURVEY section 8e "Collective").

One process per GPU.  Every rank owns an arena in its HBM -- NB receive buffers of `world` slots of `width` doubles each, followed by NB x world
64-bit flag words -- and maps the arenas of all its peers once, over hipIpc (torch's CUDA-IPC reductions carry the handles through the
process group; nothing else is ever sent through it).  After its step kernel a rank

    push(b, k): waits (per peer, inside the launch) until that peer's payload of step k - 1 has arrived, writes its payload of step k into slot
                [b][rank] of EVERY arena (its own included) and then stores k into flag [b][rank] there
    wait(b, k): lets the stream wait until the `world` flags of buffer b in the OWN arena have reached k

(`ssa_peer_push_f64`, `ssa_peer_wait`, include/ssa_hip.h).  No collective library, no communicator: plain
kernels, capturable into a hipGraph at any world size; over xGMI a push is `world - 1` point-to-point writes of 160 KB (20 000 objects,
covariance-trace payload), each over its own link.  The step number of a launch is seq0 (device memory) + an offset, so that a replayed graph
advances it on the device.

The reference has no counterpart (one env per process, no exchange step: SURVEY section 5 "distributed backend"); the oracle of this module is
the all-gather it replaces: `tests/test_parallel_gloo.py::test_sharded_hip_env_world2_shares_one_gpu[*-peer-*]` runs two ranks on the one card of
the test box -- per step from the host and as replayed hipGraph units -- and compares every step's reassembled observations, statistics and
states, bit for bit, with the unsharded engine and with the collective's path; `tests/test_hip_step.py::test_graphed_sharded_steps_equal_eager_steps`
covers one rank, the forced capture failure and the bounded wait.
"""
import torch
import torch.distributed as dist

from . import _lib


class PeerExchange:
    NB = 3

    def __init__(self, world, rank, width, device, group=None, timeout_s=2.0):
        dev = torch.device(device)
        if dev.type != "cuda":
            raise _lib.SsaHipError("peer-store all-gather: device memory only (the gloo / CPU steppers use torch.distributed)")
        self.world, self.rank, self.width, self.dev, self.group = int(world), int(rank), int(width), dev, group
        nb, w = self.NB, self.world
        self._n_data = nb * w * self.width
        # one allocation per rank = one IPC handle per rank; never freed before close()
        self.arena = torch.zeros(self._n_data + nb * w, dtype=torch.float64, device=dev)
        self.recv = [self.arena[b * w * self.width:(b + 1) * w * self.width] for b in range(nb)]
        self.flags = self.arena[self._n_data:].view(torch.int64)                 # [nb][world], zero
        self.seq0 = torch.zeros(1, dtype=torch.int64, device=dev)               # the step number the offsets count from (device)
        self.seq0_host = 0                                                        # ... and what the host knows it to be
        self.error = torch.zeros(1, dtype=torch.int32, device=dev)
        self._tickets = torch.zeros(self.world, dtype=torch.int32, device=dev)   # (a push is several workgroups per peer: the last raises the flag)
        self.timeout_ticks = int(timeout_s * 1e8)
        self._peers = [None] * w
        self._peers[self.rank] = self.arena
        if w > 1:
            from torch.multiprocessing.reductions import reduce_tensor
            torch.cuda.synchronize(dev)                                           # (the zeroes above are in memory before anybody maps it)
            mine = reduce_tensor(self.arena)
            got = [None] * w
            dist.all_gather_object(got, mine, group=group)
            for r in range(w):
                if r != self.rank:
                    fn, args = got[r]
                    self._peers[r] = fn(*args)                                    # hipIpcOpenMemHandle (cached per allocation by torch)
                    if self._peers[r].numel() != self.arena.numel():
                        raise _lib.SsaHipError("peer-store all-gather: rank %d maps an arena of another size" % r)
        # pointer tables, device resident: row b = where THIS rank's slot / flag of buffer b lives in every peer's arena
        dst = [[self._peers[r].data_ptr() + 8 * ((b * w + self.rank) * self.width) for r in range(w)] for b in range(nb)]
        flg = [[self._peers[r].data_ptr() + 8 * (self._n_data + b * w + self.rank) for r in range(w)] for b in range(nb)]
        self._dst = torch.tensor(dst, dtype=torch.int64, device=dev)
        self._flg = torch.tensor(flg, dtype=torch.int64, device=dev)
        self._dst_ptr = [self._dst[b].data_ptr() for b in range(nb)]
        self._flg_ptr = [self._flg[b].data_ptr() for b in range(nb)]
        self._own_flags = [self.flags[b * w:].data_ptr() for b in range(nb)]
        self._seq_ptr, self._err_ptr = self.seq0.data_ptr(), self.error.data_ptr()
        self._lib = _lib.load()
        torch.cuda.synchronize(dev)
        if w > 1:
            dist.barrier(group=group)                                             # every rank has mapped every arena before the first push

    # seq = the 1-based number of the step; the offset handed to the kernels is relative to the device-side base
    def push(self, send, b, seq, stream, after=None):
        """after = the buffer of step seq - 1: the push to peer r starts when r's payload of that step has arrived here (the reuse rule of the
        rotating buffers, inside the launch: no separate wait in front of it)"""
        _lib.check(self._lib.ssa_peer_push_f64(send.data_ptr(), self.width, self._dst_ptr[b], self._flg_ptr[b], self.world, self._seq_ptr,
                                                seq - self.seq0_host, self._own_flags[after] if after is not None else None,
                                                self.timeout_ticks, self._err_ptr, self._tickets.data_ptr(), stream), "ssa_peer_push_f64")

    def wait(self, b, seq, stream):
        _lib.check(self._lib.ssa_peer_wait(self._own_flags[b], self.world, self._seq_ptr, seq - self.seq0_host, self.timeout_ticks,
                                            self._err_ptr, stream), "ssa_peer_wait")

    def advance_on_device(self, n):
        """inside a captured unit: the base moves by n on the device at this point of the stream (every replay); the host's copy follows
        through `advanced(n)` per replay"""
        self.seq0.add_(n)

    def advanced(self, n):
        self.seq0_host += int(n)

    def check(self):
        """host-side check of the bounded waits (synchronises): raises when a source did not arrive in time"""
        e = int(self.error.item())
        if e:
            self.error.zero_()
            raise _lib.SsaHipError("peer-store all-gather: rank %d's payload did not arrive within %.1f s" % (e - 1, self.timeout_ticks / 1e8))

    def close(self):
        """unmap the peers' arenas (after every rank's last push has landed: a barrier first)"""
        if self._peers is None:
            return
        torch.cuda.synchronize(self.dev)
        if self.world > 1 and dist.is_initialized():
            dist.barrier(group=self.group)
        for r in range(self.world):
            if r != self.rank:
                self._peers[r] = None
        self._peers = None
        if self.world > 1 and dist.is_initialized():
            dist.barrier(group=self.group)
