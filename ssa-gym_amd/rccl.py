"""RCCL called directly (ctypes) for the ONE data-path collective of the sharded step.

torch.distributed stays the launcher / rendezvous layer (`backend="nccl"` is RCCL), but its ProcessGroup runs
every collective on an internal stream and hops to and from the caller's stream with events -- two cross-stream
waits per step, which is most of what the per-step all-gather costs on xGMI-sized messages.  A communicator
created here from the same ranks enqueues `ncclAllGather` straight into the stream the step kernels run in
(or into the caller's communication stream), with no hop.  The unique id travels through the existing
torch.distributed group; there is no second rendezvous."""
import ctypes as C
import os

import torch
import torch.distributed as dist

NCCL_UNIQUE_ID_BYTES = 128
ncclFloat64 = 8


class _UniqueId(C.Structure):
    _fields_ = [("internal", C.c_char * NCCL_UNIQUE_ID_BYTES)]


def _load():
    cands = [os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"), "/opt/rocm/lib/librccl.so", "librccl.so"]
    last = None
    for c in cands:   # torch's own copy first: the process then holds ONE RCCL
        try:
            return C.CDLL(c)
        except OSError as e:
            last = e
    raise OSError("librccl.so not found (%s)" % last)


class Communicator:
    """one RCCL communicator over the ranks of a torch.distributed group (default: WORLD)."""

    def __init__(self, device, group=None):
        self.comm = C.c_void_p()
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        err = None
        try:
            self.lib = _load()
            L = self.lib
            L.ncclGetUniqueId.argtypes = [C.POINTER(_UniqueId)]
            L.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, _UniqueId, C.c_int]
            L.ncclAllGather.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p]
            L.ncclCommDestroy.argtypes = [C.c_void_p]
            L.ncclCommCount.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
            L.ncclGetErrorString.argtypes = [C.c_int]
            L.ncclGetErrorString.restype = C.c_char_p
        except (OSError, AttributeError) as exc:
            self.lib, err = None, exc
        uid = _UniqueId()
        if self.rank == 0 and err is None:
            rc = self.lib.ncclGetUniqueId(C.byref(uid))
            if rc != 0:
                err = RuntimeError("ncclGetUniqueId failed: %s" % self.lib.ncclGetErrorString(rc).decode())
        # every rank takes part in the broadcast and in the agreement below, whether or not its own load succeeded:
        # a one-sided failure must not leave the other ranks blocked in a collective
        dev = torch.device("cuda", torch.cuda.current_device())
        raw = bytes((C.c_ubyte * NCCL_UNIQUE_ID_BYTES).from_buffer_copy(uid))   # (raw memory: the id may contain NULs)
        buf = torch.tensor(list(raw), dtype=torch.uint8, device=dev)
        dist.broadcast(buf, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        ok = torch.tensor([0 if err is not None else 1], dtype=torch.int32, device=dev)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)
        if int(ok.item()) != 1:
            raise RuntimeError("direct RCCL communicator: library load / unique id failed on at least one rank (%s)" % (err,))
        C.memmove(C.byref(uid), bytes(buf.cpu().tolist()), NCCL_UNIQUE_ID_BYTES)
        # (created on the process's current device, as set by the launcher).  ncclCommInitRank is itself a collective: if it fails
        # or never starts on ONE rank the others would sit in it forever.  It runs in a helper thread with a deadline
        # (SSA_RCCL_INIT_TIMEOUT seconds, default 60), and the ranks agree on the outcome (all-reduce MIN over the process group)
        # before anybody uses the communicator: a one-sided failure makes EVERY rank raise here -- parallel._direct_communicator
        # then falls back to torch.distributed's all-gather on all ranks alike.  (A rank whose helper thread is still blocked
        # abandons it: the thread is a daemon and the communicator is never used.)
        import threading
        res = {}

        def _init():
            try:
                torch.cuda.set_device(dev)
                res["rc"] = self.lib.ncclCommInitRank(C.byref(self.comm), self.world, uid, self.rank)
            except Exception as exc:  # noqa: BLE001
                res["exc"] = exc
        th = threading.Thread(target=_init, daemon=True)
        th.start()
        th.join(float(os.environ.get("SSA_RCCL_INIT_TIMEOUT", "60")))
        mine = (not th.is_alive()) and res.get("rc", -1) == 0
        ok = torch.tensor([1 if mine else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)
        if int(ok.item()) != 1:
            why = ("timed out" if th.is_alive() else "raised %r" % (res["exc"],) if "exc" in res else
                   ("failed: %s" % self.lib.ncclGetErrorString(res["rc"]).decode()) if res.get("rc", 0) != 0 else "failed on another rank")
            if mine:
                self.close()
            else:
                self.comm = C.c_void_p()
            raise RuntimeError("direct RCCL communicator: ncclCommInitRank %s" % why)

    def count(self):
        """number of ranks of the communicator as RCCL itself reports it (ncclCommCount)"""
        n = C.c_int(0)
        self._check(self.lib.ncclCommCount(self.comm, C.byref(n)), "ncclCommCount")
        return int(n.value)

    def _check(self, rc, what):
        if rc != 0:
            raise RuntimeError("%s failed: %s" % (what, self.lib.ncclGetErrorString(rc).decode()))

    def all_gather_f64(self, send_ptr, recv_ptr, count, stream):
        """recv[r*count:(r+1)*count] = rank r's send[0:count] (doubles), enqueued in `stream` (a hipStream_t handle)."""
        self._check(self.lib.ncclAllGather(send_ptr, recv_ptr, count, ncclFloat64, self.comm, stream), "ncclAllGather")

    def close(self):
        if self.comm:
            self.lib.ncclCommDestroy(self.comm)
            self.comm = C.c_void_p()
