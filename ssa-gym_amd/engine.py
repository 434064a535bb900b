"""HotPathEngine: device-resident state of E environments x m objects and the per-step
launch sequence (fused step kernel + reward statistics kernel).

Data layout in HBM (all float64, the reference's own array-of-structures shapes so a
history slot is byte-identical to the reference's `x_true[i]`, `x_filter[i]`, `P_filter[i]`,
`obs[i]` numpy slices -- ssa_tasker_simple_2.py:132-161):

    x_true  [H][E*m][6]        x_filter [H][E*m][6]       P_filter [H][E*m][6][6]
    obs     [H][E*m][12]       metrics  [H][E][4][m]      stats    [H][E][8]
    upd     [H][E][64]         status   [E*m] int32

H is the history depth: the reference keeps the whole episode (H = n steps); with 288 GB of
HBM that is affordable up to hundreds of thousands of objects, and the step kernel then
reads slot i-1 and writes slot i with no copy.  H = 2 is the ping-pong minimum (what
agents.py needs: P_filter[i] and P_filter[i-1]).
"""
import ctypes as C

import numpy as np
import torch

from . import _lib, device

f64 = torch.float64


class _Snapshot(tuple):
    """a snapshot's tensors + the storage layout (`order`) they were taken under"""
    order = None


class HotPathEngine:
    def __init__(self, consts, n_obj, n_env, trans, z_noise, history, device_name="cuda",
                 zn_stride_env=None, zn_stride_time=None, zn_stride_obj=3):
        _lib.load()
        if not torch.cuda.is_available():
            raise _lib.SsaHipError("no GPU visible: the ssa-gym hot path runs on MI355X only (no CPU fallback)")
        self.consts = consts
        self.m, self.E, self.H = int(n_obj), int(n_env), int(history)
        if self.H < 2:
            raise ValueError("history depth must be >= 2")
        self.dev = torch.device(device_name)
        N = self.m * self.E
        d = self.dev
        self.x_true = torch.zeros((self.H, N, 6), dtype=f64, device=d)
        self.x_filter = torch.zeros((self.H, N, 6), dtype=f64, device=d)
        self.P_filter = torch.zeros((self.H, N, 6, 6), dtype=f64, device=d)
        self.obs = torch.zeros((self.H, N, 12), dtype=f64, device=d)
        self.metrics = torch.full((self.H, self.E, 4, self.m), float("nan"), dtype=f64, device=d)
        self.stats = torch.zeros((self.H, self.E, _lib.STAT_STRIDE), dtype=f64, device=d)
        self.upd = torch.zeros((self.H, self.E, _lib.UPD_STRIDE), dtype=f64, device=d)
        self.status = torch.zeros(N, dtype=torch.int32, device=d)
        self.trans = device.as_dev(np.asarray(trans, dtype=np.float64).reshape(-1, 9), d)
        self.n_time = self.trans.shape[0]
        self.z_noise = z_noise if isinstance(z_noise, torch.Tensor) else device.as_dev(z_noise, d)
        # default strides for z_noise[E][n_time][m][3]
        self.zn_stride_time = self.m * 3 if zn_stride_time is None else int(zn_stride_time)
        self.zn_stride_env = self.n_time * self.m * 3 if zn_stride_env is None else int(zn_stride_env)
        # per-env time origin and action words: two views of ONE device buffer, so that a caller that refreshes both every step
        # (the vector env) needs a single host-to-device copy
        self.time_actions = torch.zeros(2 * self.E, dtype=torch.int32, device=d)
        self.time_actions[self.E:] = -1
        self.env_time0 = self.time_actions[:self.E]
        self.actions = self.time_actions[self.E:]
        self._p = _lib.ssa_step_params()
        self._p.n_obj, self._p.n_env = self.m, self.E
        self._p.status = self.status.data_ptr()
        self._p.trans = self.trans.data_ptr()
        self._p.env_time = self.env_time0.data_ptr()
        # the kernel indexes z_noise[e*stride_env + (i % n_time)*stride_time + a*stride_obj + 0..2]: check the extent here
        need = (self.E - 1) * self.zn_stride_env + (self.n_time - 1) * self.zn_stride_time + (self.m - 1) * int(zn_stride_obj) + 3
        if min(self.zn_stride_env, self.zn_stride_time, int(zn_stride_obj)) < 0 or self.z_noise.numel() < need \
                or self.z_noise.dtype != torch.float64 or not self.z_noise.is_contiguous():
            raise _lib.SsaHipError("z_noise: %d contiguous float64 values needed for (n_env=%d, n_time=%d, n_obj=%d) with strides "
                                   "(%d, %d, %d), got %s of %d" % (need, self.E, self.n_time, self.m, self.zn_stride_env,
                                                                   self.zn_stride_time, int(zn_stride_obj), self.z_noise.dtype,
                                                                   self.z_noise.numel()))
        self._p.z_noise = self.z_noise.data_ptr()
        self._p.zn_stride_env, self._p.zn_stride_time = self.zn_stride_env, self.zn_stride_time
        self._p.zn_stride_obj = int(zn_stride_obj)
        self._p.n_time = self.n_time
        self._lib = _lib.load()
        self._stats_ws = device.stats_workspace(self.E, d)
        self.work = torch.zeros(self._lib.ssa_env_step_work_bytes(self.m, self.E) // 4, dtype=torch.int32, device=d)
        self._p.work, self._p.stat_ws, self._p.launch_mask = self.work.data_ptr(), self._stats_ws.data_ptr(), 0
        # filter_error()'s records written by the kernels themselves into host-mapped pinned memory (ssa_step_params.fail_log): one record
        # per filter that fails, capacity = every filter once per episode; the counter is a device word, zeroed with the episode
        self.fail_log_host = torch.zeros((N, _lib.FAIL_STRIDE), dtype=f64).pin_memory()
        self.fail_log = self.fail_log_host.numpy()
        self.fail_count = torch.zeros(1, dtype=torch.int32, device=d)
        self._p.fail_log, self._p.fail_count, self._p.fail_cap = self.fail_log_host.data_ptr(), self.fail_count.data_ptr(), N
        # statistics accumulators of the atomics path; two sets, alternated when the fold is deferred
        self._shard_sets = torch.zeros((2, self.E, _lib.STAT_SHARDS, _lib.STAT_SHARD_WORDS), dtype=torch.int64, device=d)
        self.stat_shards = self._shard_sets[0]
        self._shard_cur = 0
        self._fold_pending = None      # (shard set index, stats destination) of a step whose fold was deferred
        self._shard_ptr = [self._shard_sets[0].data_ptr(), self._shard_sets[1].data_ptr()]
        # arg-max slots of sigma_pos (ssa_step_params.spos_tiles: the 'shaped' reward on the one-launch paths), one set per shard set;
        # the step kernel needs whole tiles per env for them
        self.ntiles = (N + 3) // 4
        self.supports_argmax = self.E == 1 or self.m % 4 == 0
        self._spos_sets = None
        self._actions_ptr = self.actions.data_ptr()
        self._order = None             # storage layout (set_layout): position -> the caller's index, or None
        self._pcache = {}
        self._cref = C.byref(self.consts)
        self._pref = C.byref(self._p)
        # element strides of one history slot
        self._sx, self._sP, self._so = N * 6 * 8, N * 36 * 8, N * 12 * 8
        self._sm, self._ss, self._su = self.E * 4 * self.m * 8, self.E * _lib.STAT_STRIDE * 8, self.E * _lib.UPD_STRIDE * 8
        self._bx_t, self._bx, self._bP = self.x_true.data_ptr(), self.x_filter.data_ptr(), self.P_filter.data_ptr()
        self._bo, self._bm, self._bs, self._bu = (self.obs.data_ptr(), self.metrics.data_ptr(), self.stats.data_ptr(),
                                                  self.upd.data_ptr())

    # ------------------------------------------------------------------ layout (ssa_step_params.obj_ids)
    def set_layout(self, order):
        """Store the objects in another order than the caller numbers them: position i of every state tensor holds the object the
        caller calls order[i]; None: the caller's order.  Objects of one orbit regime then share wavefronts (catalogue.regime_order) --
        late in an episode the diverged filters are the LEO objects, and packed they cost a launch 10 % less.
        The per-step launches speak the caller's indices wherever an index enters or leaves (actions, failure records, arg-max of sigma_pos,
        the host-facing observation rows: include/ssa_hip.h); the state tensors of this engine (x_true, x_filter, P_filter, obs, metrics,
        status) are in STORAGE order while a layout is set -- `to_caller_order()` puts them back (the rollout and closed-loop launches keep the
        layout).  Call before load_state(); the state present is not moved.
        Several envs (the per-step launch only): `order` is [n_env][n_obj], one permutation per env (indices within the env, as the actions
        are), n_obj % 4 == 0; `set_env_layout(e, order)` replaces one env's row (a vector env's auto-reset)."""
        if order is None and self._order is None:
            return
        if order is not None:       # (checked before anything changes: a refused table leaves the layout in force as it was)
            order = np.asarray(order, dtype=np.int64)
            if self.E == 1:
                order = order.reshape(-1)
            if order.shape != ((self.m,) if self.E == 1 else (self.E, self.m)) or \
                    not np.array_equal(np.sort(order.reshape(self.E, self.m), axis=1), np.broadcast_to(np.arange(self.m), (self.E, self.m))):
                raise _lib.SsaHipError("set_layout: `order` must be a permutation of 0 .. n_obj - 1 (one per env)")
            if self.E > 1 and self.m % 4:
                raise _lib.SsaHipError("a storage layout with several envs needs n_obj % 4 == 0 (whole tiles per env)")
        self._order = None
        self._p.obj_ids = 0
        self._pcache.clear()
        if order is None:
            return
        self._order = order.copy()
        self._obj_ids = torch.full((4 * self.ntiles,), -1, dtype=torch.int32, device=self.dev)   # (whole tiles: the kernel reads a tile's four words at once)
        self._p.obj_ids = self._obj_ids.data_ptr()
        self._upload_layout()

    def _upload_layout(self):
        """the device tables of self._order (in place: the launch parameter blocks keep their pointers)"""
        N = self.m * self.E
        rows = (self._order.reshape(self.E, self.m) + (np.arange(self.E, dtype=np.int64) * self.m)[:, None]).reshape(-1)
        self._obj_ids[:N].copy_(torch.as_tensor(self._order.reshape(-1).astype(np.int32)))
        self._order_idx = torch.as_tensor(rows).to(self.dev)                      # storage position -> the caller's row
        self._slot_of = torch.as_tensor(np.argsort(rows)).to(self.dev)            # the caller's row -> storage position
        self._slot_of32 = self._slot_of.to(torch.int32)                           # (the closed loop's inverse table: ssa_closed_loop_params.slot_of)

    def set_env_layout(self, e, order):
        """one env's row of a several-env layout replaced (before load_env_state of that env: the state present is not moved)"""
        if self._order is None or self.E == 1:
            raise _lib.SsaHipError("set_env_layout: set_layout([n_env][n_obj]) first")
        order = np.asarray(order, dtype=np.int64).reshape(-1)
        if order.shape[0] != self.m or not np.array_equal(np.sort(order), np.arange(self.m)):
            raise _lib.SsaHipError("set_env_layout: `order` must be a permutation of 0 .. n_obj - 1")
        self._order[e] = order
        self._upload_layout()

    def _reorder(self, idx, slots):
        """every per-object tensor of the given history slots gathered through `idx` (new[i] = old[idx[i]], rows of all envs), and the status words"""
        N = self.m * self.E
        for sl in slots:
            for tns in (self.x_true, self.x_filter, self.P_filter, self.obs):
                tns[sl].copy_(tns[sl].index_select(0, idx))
            mt = self.metrics[sl].reshape(self.E, 4, self.m).permute(1, 0, 2).reshape(4, N).index_select(1, idx)      # [E][4][m]
            self.metrics[sl].copy_(mt.reshape(4, self.E, self.m).permute(1, 0, 2).reshape(self.metrics[sl].shape))
        self.status.copy_(self.status.index_select(0, idx))

    def caller_rows(self, tensor):
        """a per-object tensor of ONE history slot ([n_obj, ...]) as the caller numbers the objects: a gather while a layout is set, the
        tensor itself otherwise (for the occasional reader that should not cost the layout: the observation a reset returns)"""
        return tensor if self._order is None else tensor.index_select(0, self._slot_of)

    def to_caller_order(self):
        """put the state tensors back into the caller's order and drop the layout (asynchronous, in the current stream): for everything that
        reads them as the caller numbers them -- a policy's views of the state, the operator entry points, inspection"""
        if self._order is None:
            return
        self.flush_stats()
        self._reorder(self._slot_of, range(self.H))
        self.set_layout(None)

    # ------------------------------------------------------------------ state in
    def load_state(self, slot, x_true, x_filter, P_filter):
        """reset(): place the initial truth / estimates / covariances in history slot `slot`
        and compute obs + metrics + stats for it (ssa_tasker_simple_2.py:228-229)."""
        N = self.m * self.E
        self.x_true[slot].copy_(device.as_dev(np.asarray(x_true).reshape(N, 6), self.dev))
        self.x_filter[slot].copy_(device.as_dev(np.asarray(x_filter).reshape(N, 6), self.dev))
        self.P_filter[slot].copy_(device.as_dev(np.asarray(P_filter).reshape(N, 6, 6), self.dev))
        self.status.zero_()
        self.fail_count.zero_()
        for e in range(self.E):
            sl = slice(e * self.m, (e + 1) * self.m)
            device.observe(self.x_true[slot, sl], self.x_filter[slot, sl], self.P_filter[slot, sl],
                           obs=self.obs[slot, sl], metrics=self.metrics[slot, e])
        device.reward_stats(self.metrics[slot], self.status, self.m, self.E, out=self.stats[slot])
        if self._order is not None:      # (statistics first, in the caller's order -- np.argmax's first maximum -- then into storage order)
            self._reorder(self._order_idx, [slot])

    def load_env_state(self, slot, e, x_true, x_filter, P_filter):
        """reset() of ONE environment of a vectorised batch: overwrite its slice of `slot`."""
        sl = slice(e * self.m, (e + 1) * self.m)
        self.x_true[slot, sl].copy_(device.as_dev(np.asarray(x_true).reshape(self.m, 6), self.dev))
        self.x_filter[slot, sl].copy_(device.as_dev(np.asarray(x_filter).reshape(self.m, 6), self.dev))
        self.P_filter[slot, sl].copy_(device.as_dev(np.ascontiguousarray(np.broadcast_to(P_filter, (self.m, 6, 6))), self.dev))
        self.status[sl].zero_()
        device.observe(self.x_true[slot, sl], self.x_filter[slot, sl], self.P_filter[slot, sl],
                       obs=self.obs[slot, sl], metrics=self.metrics[slot, e])
        device.reward_stats(self.metrics[slot, e:e + 1], self.status[sl], self.m, 1, out=self.stats[slot, e:e + 1])
        if self._order is not None:      # (statistics first, in the caller's order; then the env's rows into storage order)
            idx = torch.as_tensor(self._order.reshape(self.E, self.m)[e]).to(self.dev)
            for tns in (self.x_true, self.x_filter, self.P_filter, self.obs):
                tns[slot, sl].copy_(tns[slot, sl].index_select(0, idx))
            self.metrics[slot, e].copy_(self.metrics[slot, e].reshape(4, self.m).index_select(1, idx).reshape(self.metrics[slot, e].shape))

    def env_caller_rows(self, e, tensor):
        """rows of ONE env ([n_obj, ...], storage order) as the caller numbers that env's objects"""
        if self._order is None:
            return tensor
        idx = self._slot_of[e * self.m:(e + 1) * self.m] - e * self.m
        return tensor.index_select(0, idx)

    def snapshot(self, slot):
        """device-side copy of a history slot (initial state of an episode); remembers the storage layout it was taken under."""
        snap = _Snapshot((self.x_true[slot].clone(), self.x_filter[slot].clone(), self.P_filter[slot].clone(),
                          self.obs[slot].clone(), self.metrics[slot].clone(), self.stats[slot].clone()))
        snap.order = None if self._order is None else self._order.copy()
        return snap

    def restore(self, slot, snap):
        """reset(): device-to-device restore of an episode's initial state, asynchronous (and of the layout the snapshot was taken under)."""
        order = getattr(snap, "order", None)
        if (order is None) != (self._order is None) or (order is not None and not np.array_equal(order, self._order)):
            self.flush_stats()
            self.set_layout(order)
        xt, x, P, obs, met, st = snap[:6]
        self.x_true[slot].copy_(xt)
        self.x_filter[slot].copy_(x)
        self.P_filter[slot].copy_(P)
        self.obs[slot].copy_(obs)
        self.metrics[slot].copy_(met)
        self.stats[slot].copy_(st)
        self.status.zero_()
        self.fail_count.zero_()

    def snapshot_state(self, slot):
        """a history slot AND the per-object status words: what a launch that may have to be undone (the persistent closed loop
        when it gives up) restores"""
        base = self.snapshot(slot)
        snap = _Snapshot(tuple(base) + (self.status.clone(), self.fail_count.clone()))
        snap.order = base.order
        return snap

    def restore_state(self, slot, snap):
        self.restore(slot, snap)
        self.status.copy_(snap[6])
        self.fail_count.copy_(snap[7])

    # ------------------------------------------------------------------ one step
    def _spos_ptr(self, k):
        if self._spos_sets is None:
            self._spos_sets = torch.zeros((2, self.ntiles, 2), dtype=torch.int64, device=self.dev)
        return self._spos_sets[k].data_ptr()

    def _step_params(self, slot_in, slot_out, aer_out, stats_out, upd_out, shard_set, shards_out=0, shards_clear=0, aer_cols=4, obs_mirror=0,
                     argmax=False):
        """parameter block of a step between two history slots: everything but the time index, the action pointer and the
        deferred-fold hand-over is fixed per (slot pair, outputs, shard set), so the blocks are built once and cached -- a
        step then costs a handful of field stores on the host instead of twenty"""
        key = (slot_in, slot_out, aer_out, stats_out, upd_out, shard_set, shards_out, shards_clear, aer_cols, obs_mirror, argmax)
        ent = self._pcache.get(key)
        if ent is None:
            p = _lib.ssa_step_params()
            C.memmove(C.byref(p), C.byref(self._p), C.sizeof(p))
            p.x_true_in, p.x_true_out = self._bx_t + slot_in * self._sx, self._bx_t + slot_out * self._sx
            p.x_in, p.x_out = self._bx + slot_in * self._sx, self._bx + slot_out * self._sx
            p.P_in, p.P_out = self._bP + slot_in * self._sP, self._bP + slot_out * self._sP
            p.obs = self._bo + slot_out * self._so
            p.metrics = self._bm + slot_out * self._sm
            p.upd = upd_out if upd_out else self._bu + slot_out * self._su
            p.stats = stats_out if stats_out else self._bs + slot_out * self._ss   # e.g. straight into a send buffer
            p.aer_out = aer_out
            p.stat_shards = self._shard_sets[shard_set].data_ptr() if shard_set >= 0 else 0
            p.stat_shards_prev, p.stats_prev, p.launch_mask = 0, 0, 0
            p.stat_shards_clear = 0
            p.spos_tiles = self._spos_ptr(shard_set) if (argmax and shard_set >= 0) else 0
            p.spos_tiles_prev = 0
            p.aer_cols = int(aer_cols)
            p.obs_mirror = obs_mirror
            if shards_out:      # raw-shard consumer (include/ssa_hip.h: stat_shards_clear): no fold, no `stats`
                p.stat_shards, p.stats, p.stat_shards_clear = shards_out, 0, shards_clear
            ent = (p, C.byref(p), int(p.stats or 0))
            if len(self._pcache) > 4096:
                self._pcache.clear()
            self._pcache[key] = ent
        return ent

    def launch_step(self, slot_in, slot_out, time_offset, actions_ptr=None, stream=None, aer_out=0, stats_out=0, upd_out=0,
                    fast_stats=False, defer_fold=False, profile_slot=None, shards_out=0, shards_clear=0, aer_cols=4, action=None,
                    obs_mirror=0, fold_inside=False, env_words=None, argmax_spos=False, mirror_f32=False):
        """enqueue the step; asynchronous, no host sync.  fast_stats: statistics by the step kernel's atomics (two
        launches, no arg-max of sigma_pos).  defer_fold (with fast_stats): ONE launch -- this step's
        statistics are folded by extra wavefronts of the NEXT deferred step, or by flush_stats().  action (one env): the
        action by value in the parameter block (SSA_LAUNCH_INLINE_ACTION) instead of a word in memory; obs_mirror: a second
        destination of the observation rows (host-mapped pinned memory: the observation reaches the host from inside the kernel).
        env_words = (time indices, actions) of all envs (n_env <= 8) by value in the parameter block (SSA_LAUNCH_INLINE_ENVS).
        argmax_spos (with fast_stats): np.argmax / np.max of sigma_pos in the step's statistics on the one-launch paths as well
        (ssa_step_params.spos_tiles; the 'shaped' reward) -- needs self.supports_argmax."""
        s = torch.cuda.current_stream().cuda_stream if stream is None else stream
        if shards_out:
            fast_stats, defer_fold = True, False
        defer = bool(defer_fold and fast_stats)
        if not defer and self._fold_pending is not None:
            self.flush_stats(s)       # a deferred step is followed by an immediate one: fold it first (same stream, in order)
        argmax = bool(argmax_spos and fast_stats and not shards_out)
        if argmax and not self.supports_argmax:
            raise _lib.SsaHipError("argmax_spos on the one-launch paths needs whole tiles per env (n_env == 1 or n_obj % 4 == 0)")
        p, pref, stats_ptr = self._step_params(slot_in, slot_out, aer_out, stats_out, upd_out, self._shard_cur if fast_stats else -1,
                                               shards_out, shards_clear, aer_cols, obs_mirror, argmax)
        p.time_offset = int(time_offset)
        p.actions = self._actions_ptr if actions_ptr is None else actions_ptr
        inline = _lib.LAUNCH_MIRROR_F32 if mirror_f32 else 0      # (obs_mirror / aer_out are float arrays: the host-facing copy in single precision)
        if env_words is not None:
            if self.E > _lib.INLINE_ENVS:
                raise _lib.SsaHipError("env_words: at most %d envs travel in the parameter block" % _lib.INLINE_ENVS)
            p.inline_time[:self.E] = env_words[0]
            p.inline_action[:self.E] = env_words[1]
            inline |= _lib.LAUNCH_INLINE_ENVS
        elif action is not None:
            p.action0, inline = int(action), inline | _lib.LAUNCH_INLINE_ACTION
        if fold_inside and fast_stats and not defer and not shards_out:
            inline |= _lib.LAUNCH_FOLD_INSIDE      # (the step kernel's last wavefront folds the statistics: no fold launch)
        if defer and self._fold_pending is not None:
            p.launch_mask = _lib.LAUNCH_DEFER_FOLD | inline
            p.stat_shards_prev = self._shard_ptr[self._fold_pending[0]]
            p.stats_prev = self._fold_pending[1]
            p.spos_tiles_prev = self._spos_ptr(self._fold_pending[0]) if self._fold_pending[2] else 0
        else:
            p.launch_mask = (_lib.LAUNCH_DEFER_FOLD if defer else 0) | inline
            p.stat_shards_prev, p.stats_prev, p.spos_tiles_prev = 0, 0, 0
        if profile_slot is None:
            rc = self._lib.ssa_env_step_f64(self._cref, pref, s)
        else:   # the dominant launch bracketed by event pair `profile_slot` (read back with profile_ms)
            rc = self._lib.ssa_env_step_profiled_f64(self._cref, pref, s, int(profile_slot))
        if rc:
            raise _lib.SsaHipError("ssa_env_step_f64 failed with code %d" % rc)
        if defer:
            self._fold_pending = (self._shard_cur, stats_ptr, argmax)
            self._shard_cur ^= 1

    def launch_rollout(self, slot_in, time_offset, actions, stream=None, argmax_spos=False):
        """K = actions.shape[0] consecutive steps in one launch (include/ssa_hip.h: ssa_env_rollout_f64): step k reads
        history slot (slot_in + k) % H, writes (slot_in + k + 1) % H and has time index time_offset + k.  `actions`
        is a device int32 tensor [K][E] (open-loop schedule).  Statistics: the last min(K, H) steps' slots."""
        if not (isinstance(actions, torch.Tensor) and actions.is_cuda and actions.dtype == torch.int32 and actions.is_contiguous()
                and actions.dim() == 2 and actions.shape[1] == self.E and actions.shape[0] >= 1):
            raise _lib.SsaHipError("rollout: actions must be a contiguous CUDA int32 tensor [K][n_env]")
        s = torch.cuda.current_stream().cuda_stream if stream is None else stream
        self.flush_stats(s)
        K = int(actions.shape[0])
        if getattr(self, "_roll_shards", None) is None or self._roll_shards.shape[0] < K:
            self._roll_shards = torch.zeros((K, self.E, _lib.STAT_SHARDS, _lib.STAT_SHARD_WORDS), dtype=torch.int64, device=self.dev)
        r = _lib.ssa_rollout_params()
        r.n_steps, r.history, r.slot_out = K, self.H, (int(slot_in) + 1) % self.H
        r.x_true_ring, r.x_ring, r.P_ring = self._bx_t, self._bx, self._bP
        r.obs_ring, r.metrics_ring, r.upd_ring, r.stats_ring = self._bo, self._bm, self._bu, self._bs
        r.actions, r.stat_shards = actions.data_ptr(), self._roll_shards.data_ptr()
        r.spos_tiles = 0
        if argmax_spos:     # per-step arg-max slots: every step's statistics carry np.argmax(sigma_pos) (the 'shaped' reward)
            if not self.supports_argmax:
                raise _lib.SsaHipError("argmax_spos needs whole tiles per env (n_env == 1 or n_obj % 4 == 0)")
            if getattr(self, "_roll_spos", None) is None or self._roll_spos.shape[0] < K:
                self._roll_spos = torch.zeros((K, self.ntiles, 2), dtype=torch.int64, device=self.dev)
            r.spos_tiles = self._roll_spos.data_ptr()
        p = self._p
        p.time_offset = int(time_offset)
        p.launch_mask, p.stat_shards_prev, p.stats_prev, p.aer_out = 0, 0, 0, 0
        p.spos_tiles, p.spos_tiles_prev = 0, 0
        rc = self._lib.ssa_env_rollout_f64(self._cref, self._pref, C.byref(r), s)
        if rc:
            raise _lib.SsaHipError("ssa_env_rollout_f64 failed with code %d" % rc)

    def launch_closed_loop(self, slot_in, time_offset, kind, actions, stats_out, upd_out=None, fallback=None, picks=None, stream=None,
                           argmax_spos=False, wait_ticks=0, debug_withhold=False):
        """K steps AND the K decisions of a greedy agent in ONE persistent launch (include/ssa_hip.h:
        ssa_env_closed_loop_f64).  `actions`: device int32 [K + 1], actions[0] = the first step's action (given), the kernel
        writes actions[1..K]; `stats_out` device float64 [K][STAT_STRIDE]; `upd_out` [K][UPD_STRIDE] or None; `fallback`
        int32 [K + 1] or None; `picks` int64 [K + 1][2] or None.  Step k reads history slot (slot_in + k) % H and writes
        (slot_in + k + 1) % H with time index time_offset + k.  Returns False -- nothing enqueued -- when the library
        declines the configuration (several envs, or more objects than resident wavefronts x 4): the caller then issues
        the per-step launches.  self.loop_error (host-mapped int32, cleared before every launch) turns 1 if the launch gave up:
        a wavefront waited longer than `wait_ticks` (100 MHz ticks; 0 = 2 s) for a decision.  argmax_spos: the statistics carry
        np.argmax(sigma_pos) (SSA_LOOP_ARGMAX_SPOS).  debug_withhold: diagnostic -- the decision is never published (the test of
        the give-up path)."""
        if self.E != 1:
            return False
        K = int(actions.numel()) - 1
        if K < 1:
            raise _lib.SsaHipError("closed loop: actions must hold K + 1 >= 2 words")
        for t, dt, n, nm in ((actions, torch.int32, K + 1, "actions"), (stats_out, f64, K * _lib.STAT_STRIDE, "stats_out"),
                             (upd_out, f64, K * _lib.UPD_STRIDE, "upd_out"), (fallback, torch.int32, K + 1, "fallback"),
                             (picks, torch.int64, 2 * (K + 1), "picks")):
            if t is None:
                continue
            if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == dt and t.is_contiguous() and t.numel() >= n):
                raise _lib.SsaHipError("closed loop: %s must be a contiguous CUDA %s tensor of >= %d elements" % (nm, dt, n))
        if getattr(self, "_loop_ws", None) is None:
            nb = int(self._lib.ssa_closed_loop_workspace_bytes(self.m, self.E))
            if nb <= 0:
                return False
            self._loop_ws = torch.zeros(nb // 8, dtype=torch.int64, device=self.dev)
            self._loop_err_host = torch.zeros(1, dtype=torch.int32).pin_memory()
            self.loop_error = self._loop_err_host.numpy()
        s = torch.cuda.current_stream().cuda_stream if stream is None else stream
        self.flush_stats(s)
        self.loop_error[0] = 0      # (a launch that gave up must not poison the next one: nothing is in flight here, run_agent synchronises)
        r = _lib.ssa_closed_loop_params()
        r.n_steps, r.history, r.slot_out, r.agent = K, self.H, (int(slot_in) + 1) % self.H, int(kind)
        r.x_true_ring, r.x_ring, r.P_ring = self._bx_t, self._bx, self._bP
        r.obs_ring, r.metrics_ring = self._bo, self._bm
        r.upd_out = upd_out.data_ptr() if upd_out is not None else 0
        r.stats_out, r.actions = stats_out.data_ptr(), actions.data_ptr()
        r.fallback = fallback.data_ptr() if fallback is not None else 0
        r.picks = picks.data_ptr() if picks is not None else 0
        r.error = self._loop_err_host.data_ptr()
        r.workspace, r.workspace_bytes = self._loop_ws.data_ptr(), self._loop_ws.numel() * 8
        r.wait_ticks = int(wait_ticks)
        r.flags = (_lib.LOOP_ARGMAX_SPOS if argmax_spos else 0) | (_lib.LOOP_DEBUG_WITHHOLD if debug_withhold else 0)
        r.slot_of = self._slot_of32.data_ptr() if self._order is not None else 0     # (a storage layout: its inverse table)
        p = self._p
        p.time_offset = int(time_offset)
        p.launch_mask, p.stat_shards_prev, p.stats_prev, p.aer_out = 0, 0, 0, 0
        p.spos_tiles, p.spos_tiles_prev = 0, 0
        rc = self._lib.ssa_env_closed_loop_f64(self._cref, self._pref, C.byref(r), s)
        if rc == _lib.E_UNSUPPORTED:
            return False
        if rc:
            raise _lib.SsaHipError("ssa_env_closed_loop_f64 failed with code %d" % rc)
        return True

    def launch_agent_select(self, slot_cur, time_offset, kind, action_ptr, fallback_ptr=0, pick_ptr=0, stream=None, have_prev=True):
        """enqueue the device-side agent (include/ssa_hip.h: ssa_agent_select_f64): choose, for every env, the action of
        the NEXT step from history slot `slot_cur` (and the slot before it for the Shannon agent) and store it in the
        int32 word(s) at `action_ptr` -- the pointer the next launch_step() is given as actions_ptr.  No host sync."""
        if getattr(self, "_agent_ws", None) is None:
            nb = self._lib.ssa_agent_select_workspace_bytes(self.m, self.E)
            self._agent_ws = torch.empty(max(int(nb), 16), dtype=torch.uint8, device=self.dev)
        s = torch.cuda.current_stream().cuda_stream if stream is None else stream
        sc = int(slot_cur) % self.H
        sp = (sc + self.H - 1) % self.H
        # (with a storage layout the candidates are named as the caller numbers them: ssa_agent_select_ids_f64)
        rc = self._lib.ssa_agent_select_ids_f64(self._cref, int(kind), self._bx_t + sc * self._sx, self._bx + sc * self._sx,
                                                self._bP + sc * self._sP, (self._bP + sp * self._sP) if have_prev else 0, self.trans.data_ptr(),
                                                self.env_time0.data_ptr(), int(time_offset), self.n_time, fallback_ptr,
                                                self._agent_ws.data_ptr(), action_ptr, pick_ptr, self.m, self.E,
                                                self._obj_ids.data_ptr() if self._order is not None else 0, s)
        if rc:
            raise _lib.SsaHipError("ssa_agent_select_f64 failed with code %d" % rc)

    def profile_ms(self, slot):
        """duration [ms] of the dominant kernel of the step launched with profile_slot=slot (waits for it)."""
        ms = C.c_float(0.0)
        rc = self._lib.ssa_env_step_profile_ms(int(slot), C.byref(ms))
        if rc:
            raise _lib.SsaHipError("ssa_env_step_profile_ms failed with code %d" % rc)
        return float(ms.value)

    def flush_stats(self, stream=None):
        """fold the statistics of the last deferred step (no-op when nothing is pending)."""
        if self._fold_pending is None:
            return
        s = torch.cuda.current_stream().cuda_stream if stream is None else stream
        k, dst, argmax = self._fold_pending
        if argmax:
            rc = self._lib.ssa_stats_fold_spos_f64(self._shard_sets[k].data_ptr(), self._spos_ptr(k), dst, self.m, self.E, s)
        else:
            rc = self._lib.ssa_stats_fold_f64(self._shard_sets[k].data_ptr(), dst, self.E, s)
        self._fold_pending = None
        if rc:
            raise _lib.SsaHipError("ssa_stats_fold_f64 failed with code %d" % rc)

    def set_actions(self, actions):
        a = torch.as_tensor(np.asarray(actions, dtype=np.int32).reshape(self.E))
        self.actions.copy_(a, non_blocking=True)
