"""Observation buffers that are handed out WITHOUT a copy and come back by themselves.

The reference's `step()` returns a fresh array per call (`ssa_tasker_simple_2.py:360-366`: `.flatten()` / a history row that no later step
overwrites): a consumer may keep it -- replay buffers, sample collectors, frame stacks.  Copying the kernel-written, host-mapped buffer into a
fresh array costs a 1.9 MB host memcpy per step at 20 000 objects (65 us of a 124 us step; 5.1 MB for the vector env).  Instead the step kernel
writes each step's observation into a buffer NOBODY HOLDS: the array `step()` returns is a new ndarray over that buffer's memory whose base
is a small ctypes object; when the array and every view derived from it are gone, the object's finaliser puts the buffer back on the free
list.  A consumer that keeps more than `cap` observations alive at once gets copies beyond that (pinned memory is not for hoarding)."""
import ctypes
import weakref

import numpy as np


class ObsPool:
    def __init__(self, n_doubles, shape, cap=64, dtype=np.float64):
        self.n, self.shape, self.cap = int(n_doubles), tuple(shape), int(cap)
        self.dtype = np.dtype(dtype)         # float64 (the reference's), or float32 (config['obs_dtype']: the kernel writes the host copy in single)
        self._ctype = ctypes.c_float if self.dtype == np.float32 else ctypes.c_double
        self._bufs, self._np, self.ptrs, self._free = [], [], [], []
        self.handed_out = 0            # (diagnostic: arrays given out / buffers ever allocated)

    def acquire(self):
        """index of a buffer nobody holds (a new pinned one while fewer than `cap` exist), or None: the caller copies this step"""
        if self._free:
            return self._free.pop()
        if len(self._bufs) < self.cap:
            import torch
            t = torch.zeros(self.n, dtype=torch.float32 if self.dtype == np.float32 else torch.float64).pin_memory()
            self._bufs.append(t)
            self._np.append(t.numpy())
            self.ptrs.append(t.data_ptr())
            return len(self._bufs) - 1
        return None

    def release(self, k):
        self._free.append(k)

    def hand_out(self, k):
        """a FRESH ndarray over buffer k (its memory is pinned, host-mapped, just written by the kernel); buffer k is free again when this
        array and every view of it have been dropped"""
        # (from_buffer, not from_address: the ctypes object then OWNS a reference to the pinned tensor's array -- an observation a consumer still
        # holds keeps its memory alive even when the env that handed it out is gone)
        c = (self._ctype * self.n).from_buffer(self._np[k])
        weakref.finalize(c, self._free.append, k)
        self.handed_out += 1
        return np.frombuffer(c, dtype=self.dtype).reshape(self.shape)

    def __len__(self):
        return len(self._bufs)
