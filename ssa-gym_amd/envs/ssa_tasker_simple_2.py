"""SSA_Tasker_Env: drop-in for the reference's envs/ssa_tasker_simple_2.py::SSA_Tasker_Env.

Same constructor dict, env id, spaces, return shapes and inspectable attributes
(SURVEY 8b); the per-object arithmetic of reset()/step() runs in the fused HIP kernels
(engine.HotPathEngine).  Host code here is orchestration only: RNG draws in the reference's
order, one launch per step, one small device->host copy for reward/done, lazy numpy views of
the device-resident history.  Citations: ssa_tasker_simple_2.py:line in the reference.
"""
import time
from copy import copy
from datetime import timedelta

import numpy as np

from .. import _lib, host
from . import transformations
from ._config import kernel_consts, resolve_kernel_variant
from ._gymshim import Env, np_random, spaces
from .results import error_failed


class _History:
    """numpy-indexable view of a device-resident history tensor [H][E*m][...] for ONE env.

    `hist[i]` -> numpy array of step i (copied from HBM on access); `hist[i, j]`, `hist[i][mask]`,
    negative indices and slices over the time axis work like on the reference's (n, m, ...) arrays.
    Only the last H steps are resident when the env was built with a shorter history."""

    def __init__(self, env, tensor, m_axis_len, tail_shape):
        self._env, self._t = env, tensor
        self.shape = (env.n, m_axis_len) + tuple(tail_shape)
        self.dtype = np.dtype(np.float64)
        self.ndim = len(self.shape)

    def __len__(self):
        return self.shape[0]

    def _slot(self, i):
        env = self._env
        i = int(i)
        if i < 0:
            i += env.n
        if not 0 <= i < env.n:
            raise IndexError(i)
        env._caller_order()       # (the history arrays are read as the env numbers the objects: a storage layout ends here)
        H = env._engine.H
        if i > env.i or i <= env.i - H:
            if i > env.i:   # not simulated yet: the reference arrays hold zeros there after reset()
                return np.zeros(self.shape[1:])
            raise IndexError("step %d is no longer resident (history depth %d, current step %d); build the env "
                             "with config['history'] = 'full'" % (i, H, env.i))
        return self._t[i % H].cpu().numpy().reshape(self.shape[1:])

    def __getitem__(self, idx):
        if isinstance(idx, tuple):
            head, rest = idx[0], idx[1:]
        else:
            head, rest = idx, ()
        if isinstance(head, slice):
            arr = np.stack([self._slot(i) for i in range(*head.indices(self.shape[0]))])
            return arr[(slice(None),) + rest] if rest else arr
        arr = self._slot(head)
        return arr[rest] if rest else arr

    def __array__(self, dtype=None, copy=None):
        a = np.stack([self._slot(i) for i in range(self.shape[0])])
        return a.astype(dtype) if dtype is not None else a


class _Sparse:
    """reference-shaped (n, m, k...) view of a quantity the reference stores only at [i, action]
    (z_true, y: NaN elsewhere -- ssa_tasker_simple_2.py:139-142, 202)."""

    def __init__(self, env, store, tail):
        self._env, self._s = env, store
        self.shape = (env.n, env.m) + tuple(tail)

    def __getitem__(self, idx):
        env = self._env
        if isinstance(idx, tuple):
            i, rest = idx[0], idx[1:]
        else:
            i, rest = idx, ()
        if isinstance(i, slice):
            return np.stack([self[k] for k in range(*i.indices(env.n))])[(slice(None),) + rest]
        i = int(i) + (env.n if int(i) < 0 else 0)
        row = np.full(self.shape[1:], np.nan)
        a = env._upd_action[i]
        if a >= 0:
            row[a] = self._s[i]
        return row[rest] if rest else row

    def __array__(self, dtype=None, copy=None):
        return np.stack([self[i] for i in range(self.shape[0])])


class _FailureMessages:
    """`failed_filters_msg` of the reference (ssa_tasker_simple_2.py:147, 380): a list of m entries, "None" until filter j fails, then
    [message].  The message -- 'Object j failed on predict step i, LinAlgError. [dpos dvel spos svel]' -- is FORMATTED WHEN IT IS READ, from the
    record the kernel wrote at the failure: an episode of the default env loses a few filters per step late on, and formatting each as it
    happened cost a gym-style step tens of microseconds."""
    KINDS = {_lib.ST_PREDICT_NAN: ('predict', ', predict returned nan. '), _lib.ST_PREDICT_LINALG: ('predict', ', LinAlgError. '),
             _lib.ST_UPDATE_NAN: ('update', ', update returned nan. '), _lib.ST_UPDATE_LINALG: ('update', ', LinAlgError. ')}

    def __init__(self, m):
        self._m, self._rec = int(m), {}

    def record(self, j, step, status, err):
        self._rec[j] = (step, status, err)

    def __len__(self):
        return self._m

    def __getitem__(self, j):
        if isinstance(j, slice):
            return [self[k] for k in range(*j.indices(self._m))]
        j = int(j)
        if j < 0:
            j += self._m
        if not 0 <= j < self._m:
            raise IndexError(j)
        r = self._rec.get(j)
        if r is None:
            return "None"
        activity, error_type = self.KINDS[r[1]]
        return ["".join(['Object ', str(j), ' failed on ', activity, ' step ', str(r[0]), error_type, str(np.round(r[2], 2))])]

    def __iter__(self):
        return (self[j] for j in range(self._m))


def _action_word(a):
    """what a policy handed back, as the int32 word the next step launch reads: a CUDA int32 tensor with one element, or -- what torch.argmax
    returns -- an int64 one, whose LOW word is the action (little endian; -1 stays -1): no cast kernel (5 us at 20 000 objects inside a replayed
    graph, profiles/r04_run_policy_timeline.txt)."""
    import torch
    if isinstance(a, torch.Tensor) and a.is_cuda and a.numel() == 1:
        if a.dtype == torch.int32:
            return a
        if a.dtype == torch.int64:
            return a.reshape(1).view(torch.int32)[:1]
    raise TypeError("run_policy: the policy must return a CUDA int32 (or int64) tensor with one element (the action)")


def _never_destroy(graph):
    """a torch.cuda.CUDAGraph whose capture FAILED must not be finalised: capture_end() threw before the graph let go of the RNG generator
    state it registered with, and in this torch build (2.10 + ROCm 7) its destructor then fails a TORCH_CHECK ("The graph should be registered
    to the state") -- an exception out of a C++ destructor: the process aborts, whenever the object happens to be freed (at the return of the
    capturing function, or later by the garbage collector).  One leaked reference keeps the few hundred bytes alive for the life of the
    process (build_ablate/r04_run20.sh isolated it: tests/test_env_gpu.py::test_run_policy_replayed_from_a_graph_equals_the_eager_loop)."""
    import ctypes
    ctypes.pythonapi.Py_IncRef(ctypes.py_object(graph))


class SSA_Tasker_Env(Env):
    metadata = {'render.modes': ['live', 'none']}
    visualization = None

    def __init__(self, config=None):
        s = time.time()
        if config is None:
            from . import env_config as config
        self.runtime = {'__init__': 0, 'reset': 0, 'step': 0, 'step prep': 0, 'propagate next true state': 0,
                        'perform predictions': 0, 'update with observation': 0, 'Observations and Reward': 0,
                        'filter_error': 0, 'visible_objects': 0, 'object_visibility': 0, 'anees': 0,
                        'failed_filters': 0, 'plot_sigma_delta': 0, 'plot_rewards': 0, 'plot_anees': 0,
                        'plot_actions': 0, 'all_true_obs': 0, 'plot_visibility': 0, 'predict method': 0}
        # ---- simulation configuration (:81-96)
        self.t_0 = config['t_0']
        self.dt = config['time_step']
        self.n = config['steps']
        self.m = config['rso_count']
        self.obs_limit = np.radians(config['obs_limit'])
        self.obs_returned = config['obs_returned']
        self.reward_type = config['reward_type']
        self.orbits = config['orbits']
        self.obs_lla = np.array(config['observer']) * [host.deg2rad, host.deg2rad, 1]
        self.obs_itrs = host.lla2ecef(self.obs_lla)
        self.update_interval = config['update_interval']
        self.i = 0
        # ---- filter configuration (:97-118)
        self.obs_type = config['obs_type']
        if self.obs_type == 'aer':
            self.z_sigma = config['z_sigma'] * np.array([host.arcsec2rad, host.arcsec2rad, 1])
        elif self.obs_type == 'xyz':
            self.z_sigma = np.asarray(config['z_sigma'], dtype=np.float64)
        else:
            print('Invalid Observation Type: ' + str(config['obs_type']))
            raise SystemExit
        self.x_sigma = np.array(config['x_sigma'])
        self.Q = host.Q_discrete_white_noise(dim=2, dt=self.dt, var=config['q_sigma'] ** 2, block_size=3,
                                             order_by_dim=False)
        self.fx, self.hx = config['fx'], config['hx']
        self.mean_z, self.residual_z, self.msqrt = config['mean_z'], config['residual_z'], config['msqrt']
        self.alpha, self.beta, self.kappa = config['alpha'], config['beta'], config['kappa']
        # operator plug points -> fused kernel variant (no CPU fallback for foreign callables)
        self._model, propagator = resolve_kernel_variant(config)
        # ---- arrays (:120-161)
        if config['P_0'] is None:
            self.P_0 = np.copy(np.diag(self.x_sigma ** 2))
        else:
            self.P_0 = np.copy(config['P_0'])
        if config['R'] is None:
            self.R = np.diag(self.z_sigma ** 2)
        else:
            self.R = np.copy(config['R'])
        self.time = [self.t_0 + (timedelta(seconds=self.dt) * i) for i in range(self.n)]
        if config.get('trans_matrix') is not None:
            self.trans_matrix = np.asarray(config['trans_matrix'], dtype=np.float64).reshape(-1, 3, 3)
        else:
            self.trans_matrix = transformations.trans_matrix_table(self.t_0, self.dt, self.n)
        self.x_noise = np.empty(shape=(self.m, 6))
        self.filters = []   # the reference keeps one filterpy object per RSO; state lives in HBM here
        self.rewards = np.empty(self.n)
        self.failed_filters_id = []
        self.failed_filters_msg = _FailureMessages(self.m)
        self.actions = np.empty(self.n, dtype=int)
        self.obs_taken = np.empty(self.n, dtype=bool)
        self.x_failed = np.copy(host.X_FAILED)
        self.P_failed = np.copy(host.P_FAILED)
        self.visibility = []
        self.sigmas_h = np.empty((self.n, 13, 3))
        self.S = np.empty((self.n, self.m, 3, 3)) if self.n * self.m <= (1 << 22) else None
        # ---- spaces (:163-177)
        self.action_space = spaces.Discrete(self.m)
        if self.obs_returned == 'flatten':
            shp = (self.m * 12,)
        elif self.obs_returned == 'aer':
            shp = (self.m * 4,)
            self.observation = np.zeros(self.m * 4)
        else:
            shp = (self.m, 12)
        # config['obs_dtype'] = 'float64' (default, the reference's) | 'float32' (EXTENSION): the observation handed to the host in single
        # precision -- the step kernel writes its host-facing copy that way (SSA_LAUNCH_MIRROR_F32): half the bytes over PCIe, for consumers
        # that cast to float32 anyway (every RL framework does).  The device-resident history and everything computed stay float64.
        self._obs_f32 = np.dtype(config.get('obs_dtype', np.float64)) == np.float32
        if np.dtype(config.get('obs_dtype', np.float64)) not in (np.dtype(np.float32), np.dtype(np.float64)):
            raise ValueError("config['obs_dtype'] must be float64 or float32")
        self.observation_space = spaces.Box(low=np.full(shp, -np.inf), high=np.full(shp, np.inf), dtype=np.float32 if self._obs_f32 else np.float64)
        # ---- device engine
        hist = config.get('history', 'auto')
        bytes_per_step = self.m * (6 + 6 + 36 + 12 + 4) * 8
        if hist == 'auto':
            hist = 'full' if self.n * bytes_per_step <= 64 * 2 ** 30 else 2
        self._H = self.n if hist == 'full' else max(2, int(hist))
        self._consts, _ = kernel_consts(config, self.Q, self.R, self.dt, self.obs_limit, self.obs_lla)
        self._engine = None
        self._device_rng = bool(config.get('device_rng', False))
        # config['storage_layout'] = None (default) | 'regime': how the engine STORES the objects (reset()); invisible but for speed
        self._storage_layout = config.get('storage_layout', None)
        if self._storage_layout not in (None, 'regime'):
            raise ValueError("config['storage_layout'] must be None or 'regime', not %r" % (self._storage_layout,))
        self._obs_buffers = config.get('obs_buffers', 2)
        # step() hands out a FRESH array per call for the 'flatten' and (m, 12) observations, as the reference does (:360-366: `.flatten()` /
        # a row of the history that no later step overwrites) -- a consumer may keep it as long as it likes (replay buffers, sample
        # collectors, GAE targets).  config['obs_zero_copy'] = True (opt-in): a VIEW of the host-mapped ring the kernel writes, valid until
        # `obs_buffers` (default 2) further steps have been taken -- no 1.9 MB host copy per step at 20 000 objects
        self._obs_zero_copy = bool(config.get('obs_zero_copy', False))
        self._obs_pool_cap = config.get('obs_pool', 64)      # (pinned buffers that may be out with the consumer at once; beyond: copies)
        # the persistent closed loop's bound on any wait inside the launch (100 MHz ticks; 0 = the library's 2 s)
        self._loop_wait_ticks = int(config.get('closed_loop_wait_ticks', 0))
        self._loop_debug_withhold = False
        # config['obs_device'] = True (opt-in, for policies that live on the GPU): step() returns the observation as a CUDA tensor
        # -- a view of the device-resident history slot ('aer': of the persistent (4 m,) block) -- and nothing crosses PCIe
        self._obs_device = bool(config.get('obs_device', False))
        # run_agent(): the persistent closed-loop kernel (one launch per chunk) unless config['closed_loop'] == 'per_step'
        self._closed_loop_persistent = config.get('closed_loop', 'persistent') != 'per_step'
        self.np_random = None
        self.init_seed = self.seed(config.get('seed'))[0]
        self.reset()
        self.runtime['__init__'] += time.time() - s

    # ------------------------------------------------------------------ gym API
    def seed(self, seed=None):
        self.np_random, seed = np_random(seed)   # old-gym hash_seed -> RandomState (:188-191)
        self.init_seed = seed
        return [seed]

    def _build_engine(self):
        from .. import engine
        self._engine = engine.HotPathEngine(self._consts, self.m, 1, self.trans_matrix, self._z_noise_dev, self._H)
        e = self._engine
        import torch
        self._aer_dev = torch.zeros(self.m * 4, dtype=torch.float64, device="cuda")
        self._stream = torch.cuda.current_stream()
        # host-mapped mailboxes (pinned memory is addressable from the GPU): the kernels read the action
        # from / write statistics and the update record to host memory directly, so a step needs one
        # stream synchronisation and one observation copy instead of four blocking transfers
        self._stats_host = torch.zeros(_lib.STAT_STRIDE, dtype=torch.float64).pin_memory()
        self._upd_host = torch.zeros(_lib.UPD_STRIDE, dtype=torch.float64).pin_memory()
        # The observation reaches the host FROM INSIDE the step kernel: its epilogue writes the (az, el, range, trace P) block
        # ('aer') or a second copy of the observation rows (other modes) straight into host-mapped pinned memory, overlapped
        # with the other wavefronts' arithmetic -- no copy-engine pass behind the kernel (1.92 MB 'flatten' vector: 44 us).
        # 'aer' hands out ONE persistent array refreshed in place, as the reference does (:362-363); the other modes alternate
        # between TWO pinned arrays the kernel writes, and step() returns a COPY of the current one (the reference returns a fresh
        # array per step) unless config['obs_zero_copy'] asks for the view itself, which stays intact until step i + 2 is taken
        # (config['obs_buffers'] = k >= 2 deepens that ring).
        aer = self.obs_returned == 'aer'
        nobs = self.m * (4 if aer else 12)
        nbuf = 1 if aer else max(2, int(self._obs_buffers))
        f32 = self._obs_f32 and not self._obs_device        # (obs_device hands out the float64 device tensors themselves)
        self._mirror_f32 = f32
        self._obs_ring = [torch.zeros(nobs, dtype=torch.float32 if f32 else torch.float64).pin_memory() for _ in range(nbuf)]
        shape = (nobs,) if self.obs_returned in ('aer', 'flatten') else (self.m, 12)
        self._obs_ring_np = [b.numpy().reshape(shape) for b in self._obs_ring]
        self._obs_ring_ptr = [b.data_ptr() for b in self._obs_ring]
        # (numpy views and raw pointers of the mailboxes, taken once: each .numpy() / .data_ptr() costs the step a microsecond)
        self._upd_np, self._stats_np = self._upd_host.numpy(), self._stats_host.numpy()
        self._upd_ptr, self._stats_ptr = self._upd_host.data_ptr(), self._stats_host.data_ptr()
        if aer:
            self.observation = self._obs_ring_np[0]
        # default hand-out of the 'flatten' / (m, 12) observation: a buffer nobody holds, written by the kernel, returned as a fresh array,
        # taken back when the consumer lets go of it (envs/_obspool.py) -- the reference's semantics without the copy
        from ._obspool import ObsPool
        self._obs_pool = None if (aer or self._obs_zero_copy or self._obs_device) else ObsPool(nobs, shape, cap=int(self._obs_pool_cap),
                                                                                               dtype=np.float32 if f32 else np.float64)
        self.x_true = _History(self, e.x_true, self.m, (6,))
        self.x_filter = _History(self, e.x_filter, self.m, (6,))
        self.P_filter = _History(self, e.P_filter, self.m, (6, 6))
        self.obs = _History(self, e.obs, self.m, (12,))
        self._met = [_History(self, e.metrics[:, 0, k], self.m, ()) for k in range(4)]
        self.delta_pos, self.delta_vel, self.sigma_pos, self.sigma_vel = self._met

    def reset(self):
        s = time.time()
        import torch
        m, n = self.m, self.n
        x_true0 = np.empty((m, 6))
        N = self.orbits.shape[0]
        if self._device_rng:   # bulk draws (this mode is not seed-compatible with the reference anyway): 1 ms instead of 40 at m = 20 000
            x_true0[:] = self.orbits[self.np_random.randint(low=0, high=N, size=m)]
            self.x_noise[:] = self.np_random.normal(size=(m, 6)) * self.x_sigma
        else:
            for j in range(m):   # draw order of :206-209: (row, 6 normals) per object ...
                x_true0[j] = self.orbits[self.np_random.randint(low=0, high=N), :]
                self.x_noise[j] = self.np_random.normal(size=6) * self.x_sigma
        x_filter0 = x_true0 + self.x_noise
        # ... then n*m*3 normals (:219-221); RandomState.normal keeps its Box-Muller cache across
        # calls, so one bulk draw consumes the stream exactly like the reference's n*m size-3 draws
        if self._device_rng:
            gen = torch.Generator(device="cuda").manual_seed(int(self.np_random.randint(0, 2 ** 31 - 1)))
            zs = torch.as_tensor(self.z_sigma, dtype=torch.float64, device="cuda")
            self._z_noise_dev = torch.randn((1, n, m, 3), dtype=torch.float64, device="cuda", generator=gen) * zs
            self.z_noise = None
        else:
            self.z_noise = self.np_random.normal(size=(n, m, 3)) * self.z_sigma
            self._z_noise_dev = torch.as_tensor(self.z_noise, dtype=torch.float64).to("cuda")
        if self._engine is None:
            self._build_engine()
        else:
            self._engine.z_noise.copy_(self._z_noise_dev.reshape(self._engine.z_noise.shape))
        # STORAGE LAYOUT (round 4, config['storage_layout'] = 'regime'; default off): the engine keeps objects of one orbit regime in the same
        # wavefronts -- ascending semi-major axis, dealt tile by tile over the XCDs (catalogue.regime_order) -- because late in an episode
        # the diverged filters are the LEO objects, and packed they cost a launch 15 % less.  Nothing of it shows: the
        # step kernel speaks the env's own indices wherever an index enters or leaves it (actions, failure records, arg-max of sigma_pos, the
        # observation rows it writes for the host), an object's arithmetic does not depend on its position (bit-identical episodes,
        # build_ablate/layout_episode_ab.py), and whatever reads the device state as the env numbers it -- the history arrays, the device-side
        # agents' scores and policies' views, rollout -- puts the state back first (_caller_order(): the layout is then off until the next
        # reset(); run_agent keeps it: its kernels take the table).  OFF by default HERE: what step() gains in the kernel (2-5 us late in an episode) it loses on the way to the host -- the
        # observation rows leave the kernel row by row at the env's indices instead of tile by tile (1.9 MB over PCIe in 96-byte pieces: 'flatten'
        # 73 -> 84 us per step at 20 000 objects).  It pays for launch sequences that keep the observations on the device: the engine-level
        # loops (HotPathEngine.set_layout; bench.py's `value`), C-ABI callers (ssa_step_params.obj_ids).
        lay = self._storage_layout == 'regime' and m >= 64 and not self._obs_device
        if lay:
            from ..catalogue import regime_order
            self._engine.set_layout(regime_order(x_true0))
        else:
            self._engine.set_layout(None)
        self._engine.load_state(0, x_true0, x_filter0, np.broadcast_to(self.P_0, (m, 6, 6)))
        # tracking variables (:222-231)
        self.actions[:], self.obs_taken[:], self.failed_filters_id, self.visibility = -1, False, [], []
        self.failed_filters_msg = _FailureMessages(self.m)
        self.rewards[:] = 0
        self.sigmas_h[:] = 0
        if self.S is not None:
            self.S[:] = np.nan
        self._y = np.full((n, 3), np.nan)
        self._z_true = np.full((n, 3), np.nan)
        self._S_sel = np.full((n, 3, 3), np.nan)
        self._upd_action = np.full(n, -1, dtype=int)
        self.y = _Sparse(self, self._y, (3,))
        self.z_true = _Sparse(self, self._z_true, (3,))
        self._n_failed = 0
        self._fail_read, self._fail_pending, self._fail_chunk_total = 0, {}, 0      # records of the kernel's failure log consumed so far
        self._engine.fail_log[:] = 0.0             # (time index 0 = "not written": steps count from 1)
        self._ring_head = None
        self._argmax_sigma_prev = None
        self.i = 0
        self._fetch_small(0)
        self.runtime['reset'] += time.time() - s
        return self._obs_out(reset=True)

    def _fetch_small(self, i):
        e = self._engine
        slot = i % e.H
        self._stats = e.stats[slot, 0].cpu().numpy()     # synchronises the stream
        self._argmax_sigma = int(self._stats[_lib.STAT_ARGMAX_SPOS])

    def _caller_order(self):
        """the device state back in the env's own object order (drops the engine's storage layout until the next reset(); a no-op without)"""
        if self._engine is not None:
            self._engine.to_caller_order()

    def _host_obs(self, arr):
        """an observation that reached the host through a device-to-host copy, in the dtype step() hands out (config['obs_dtype'])"""
        return arr.astype(np.float32) if getattr(self, "_mirror_f32", False) else arr

    def _obs_out(self, reset=False):
        """the observation of the current step through the slow path (reset(), rollout(), run_agent()): a device-to-host copy (gathered into
        the env's object order while a storage layout is set: a reset does not cost the layout)"""
        e, slot = self._engine, self.i % self._engine.H
        if self.obs_returned == 'flatten':
            return self._host_obs(e.caller_rows(e.obs[slot]).cpu().numpy().reshape(-1))
        elif self.obs_returned == 'aer':
            return self.aer_obs(self.observation)
        return self._host_obs(e.caller_rows(e.obs[slot]).cpu().numpy())

    def step(self, a):
        step_s = time.time()
        assert self.action_space.contains(a), "%r (%s) invalid" % (a, type(a))
        self._argmax_sigma_prev = self._argmax_sigma
        self._ring_head = None
        self.i += 1
        i = self.i
        self.actions[i] = np.copy(a)
        e = self._engine
        s = time.time()
        self.runtime['step prep'] += s - step_s
        # propagate + predict + update + observations/metrics + statistics: ONE launch (:265-322; the step kernel's last wavefront folds
        # the statistics, SSA_LAUNCH_FOLD_INSIDE).  The action travels by value in the parameter block; statistics, update record
        # and the observation are written by the kernel straight into host-mapped pinned memory: ONE stream synchronisation, no copy
        cur = self._stream                    # (the stream the engine was built in; torch.cuda.current_stream() costs 3 us per call,
        #                                        and every step ends with a synchronisation, so later work in any stream sees its results)
        aer = self.obs_returned == 'aer'
        k = 0 if aer else i % len(self._obs_ring)
        shaped = self.reward_type == 'shaped'      # needs np.argmax(sigma_pos[i - 1]) (:346): the arg-max slots of the one-launch path
        if self._obs_device:
            e.launch_step((i - 1) % e.H, i % e.H, i, action=int(a), aer_out=self._aer_dev.data_ptr() if aer else 0,
                          stats_out=self._stats_ptr, upd_out=self._upd_ptr, stream=cur.cuda_stream,
                          fast_stats=True, fold_inside=True, argmax_spos=shaped)
            obs_np = self._aer_dev if aer else (e.obs[i % e.H].reshape(-1) if self.obs_returned == 'flatten' else e.obs[i % e.H])
        else:
            pool = self._obs_pool
            kp = pool.acquire() if pool is not None else None
            e.launch_step((i - 1) % e.H, i % e.H, i, action=int(a),
                          aer_out=self._obs_ring_ptr[0] if aer else 0,
                          obs_mirror=0 if aer else (pool.ptrs[kp] if kp is not None else self._obs_ring_ptr[k]),
                          stats_out=self._stats_ptr, upd_out=self._upd_ptr, stream=cur.cuda_stream,
                          fast_stats=True, fold_inside=True, argmax_spos=shaped, mirror_f32=self._mirror_f32)
            obs_np = self._obs_ring_np[k]
        cur.synchronize()
        rec = self._upd_np
        self._stats = self._stats_np.copy()
        self._argmax_sigma = int(self._stats[_lib.STAT_ARGMAX_SPOS])
        t_dev = time.time()
        self.runtime['perform predictions'] += t_dev - s
        self._book_update(i, a, rec)
        n_failed = int(self._stats[_lib.STAT_N_FAILED])
        if n_failed != self._n_failed:
            self._record_failures()
        done = self._reward_done(i, a, self._stats, self._argmax_sigma_prev)
        if i + 1 >= self.n:
            done = True
        # 'aer' hands out its ONE persistent array refreshed in place, as the reference does (:362-363: self.observation); the other
        # modes a fresh copy unless config['obs_zero_copy']
        if aer or self._obs_device or self._obs_zero_copy:
            obs = obs_np
        elif kp is not None:
            obs = self._obs_pool.hand_out(kp)       # fresh array, no copy: the buffer comes back when the consumer drops it
        else:
            obs = obs_np.copy()                     # (more than `obs_pool` observations alive at once)
        e_t = time.time()
        self.runtime['Observations and Reward'] += e_t - t_dev
        self.runtime['step'] += e_t - step_s
        if self.obs_returned == 'flatten':
            return obs, self.rewards[i], done, {}
        r = self.rewards[i]   # np.nan_to_num(..., nan=0.5, posinf=0.5, neginf=0.5) of a scalar (:365-367)
        return obs, (r if np.isfinite(r) else np.float64(0.5)), done, {}

    # ------------------------------------------------------------------ per-step host bookkeeping
    def _book_update(self, i, a, rec):
        """update record of step i (:292-315) into the env's sparse histories"""
        if rec[_lib.UPD_ACTION] >= 0:
            self._upd_action[i] = int(a)
            self._z_true[i] = rec[_lib.UPD_Z_TRUE:_lib.UPD_Z_TRUE + 3]
            if rec[_lib.UPD_OBS_TAKEN] == 1.0:
                self._y[i] = rec[_lib.UPD_Y:_lib.UPD_Y + 3]
                self._S_sel[i] = rec[_lib.UPD_S:_lib.UPD_S + 9].reshape(3, 3)
                if self.S is not None:
                    self.S[i, int(a)] = self._S_sel[i]
                self.sigmas_h[i] = rec[_lib.UPD_SIGMAS_H:_lib.UPD_SIGMAS_H + 39].reshape(13, 3)
                self.obs_taken[i] = True

    def _reward_done(self, i, a, st, argmax_sigma_prev):
        """reward / done of step i from its statistics (:324-354); fills self.rewards[i]"""
        max_dpos = st[_lib.STAT_MAX_DPOS]
        done = False
        if self.reward_type == 'jones':
            if max_dpos > 5e6:
                done, self.rewards[i] = True, 0
            elif max_dpos < 3e4:
                done, self.rewards[i] = True, 1
            elif i + 1 >= self.n:
                done, self.rewards[i] = True, 0
            else:
                done, self.rewards[i] = False, 0
        elif self.reward_type == 'trinary':   # results.py:432
            self.rewards[i] = (st[_lib.STAT_CNT_LT_1E4] + st[_lib.STAT_CNT_LT_1E7]) / self.m / 2
        elif self.reward_type == 'shaped':
            if max_dpos > 5e6:
                done, self.rewards[i] = True, 0
            elif max_dpos < 3e4:
                done, self.rewards[i] = True, 1 - np.sum(self.rewards[:i])
            elif a == argmax_sigma_prev:
                self.rewards[i] = 1 / self.n
            else:
                self.rewards[i] = -1 / self.n
        return done

    def rollout(self, actions):
        """Open-loop extension (no reference counterpart as ONE call): apply `actions` as consecutive step()
        calls would -- the loop of the reference's agent_naive_random / round-robin drivers (agents.py,
        tests.py:584-603) -- with up to H-1 steps per kernel launch (ssa_env_rollout_f64: state resident on
        chip across the steps, results bit-identical to step()).  Stops at the first `done`.  Returns
        (observation after the last executed step, rewards[k], dones[k], info).  Every reward type ('shaped': the arg-max of
        sigma_pos of every step comes from the rollout's arg-max slots, ssa_rollout_params.spos_tiles)."""
        self._caller_order()
        import torch
        shaped = self.reward_type == 'shaped'     # (np.argmax(sigma_pos) of every step from the arg-max slots of the rollout)
        actions = np.asarray(actions, dtype=np.int64).ravel()
        for a in actions:
            assert self.action_space.contains(int(a)), "%r invalid" % (a,)
        e = self._engine
        K = min(len(actions), self.n - 1 - self.i)
        rewards, dones = [], []
        pos, done = 0, False
        while pos < K and not done:
            kk = min(K - pos, e.H - 1)
            i0 = self.i
            act = torch.as_tensor(actions[pos:pos + kk].astype(np.int32)).view(kk, 1).to(e.dev)
            e.launch_rollout(i0 % e.H, i0 + 1, act, argmax_spos=shaped)
            self._ring_head = i0 + kk
            slots = [(i0 + 1 + k) % e.H for k in range(kk)]
            stats = e.stats[slots, 0].cpu().numpy()          # synchronises the stream
            upd = e.upd[slots, 0].cpu().numpy()
            self._fail_chunk_total = int(stats[-1][_lib.STAT_N_FAILED])
            for k in range(kk):
                self.i += 1
                i, a = self.i, int(actions[pos + k])
                self.actions[i] = a
                self._book_update(i, a, upd[k])
                self._stats = stats[k]
                if int(stats[k][_lib.STAT_N_FAILED]) != self._n_failed:
                    self._record_failures(at_step=i)
                done = self._reward_done(i, a, stats[k], self._argmax_sigma) or (i + 1 >= self.n)
                self._argmax_sigma = int(stats[k][_lib.STAT_ARGMAX_SPOS])      # (-1 unless 'shaped' asked for it)
                r = self.rewards[i]
                rewards.append(r if (self.obs_returned == 'flatten' or np.isfinite(r)) else np.float64(0.5))
                dones.append(done)
                if done:
                    break
            pos += kk
        slot = self.i % e.H
        if self.obs_returned == 'aer':
            from .. import device
            M = e.trans[self.i % e.n_time].reshape(3, 3)
            device.aer_obs(e.x_filter[slot], e.P_filter[slot], M, self._consts, out=self._aer_dev.view(self.m, 4))
            self.observation[:] = self._aer_dev.cpu().numpy()
            obs = self.observation
        elif self.obs_returned == 'flatten':
            obs = self._host_obs(e.obs[slot].cpu().numpy().reshape(-1))
        else:
            obs = self._host_obs(e.obs[slot].cpu().numpy())
        return obs, np.asarray(rewards), np.asarray(dones, dtype=bool), {}

    AGENT_KINDS = {'agent_naive_greedy': _lib.AGENT_NAIVE_GREEDY, 'agent_visible_greedy': _lib.AGENT_VISIBLE_GREEDY,
                   'agent_visible_greedy_aer': _lib.AGENT_VISIBLE_GREEDY, 'agent_shannon': _lib.AGENT_SHANNON,
                   'agent_pos_error_greedy': _lib.AGENT_POS_ERROR, 'agent_vel_error_greedy': _lib.AGENT_VEL_ERROR}

    def run_agent(self, agent, n_steps, fallback_actions=None):
        """Closed loop on the device (no reference counterpart as ONE call): the loop
            a = agent(obs, env); obs, r, done, _ = env.step(a)
        of the reference's drivers (run_environment.py, compare_agents.py) for one of its greedy agents (agents.py: `agent`
        is the function or its name), with the agent's arg-max computed on the GPU and handed to the next step's launch
        in-stream -- no host round trip per step.  `fallback_actions[k]` replaces the reference's action_space.sample()
        when no object is visible at decision k (default: draws from the env's action space, as the reference does).
        Stops at the first `done`.  Returns (observation after the last executed step, actions[k], rewards[k], dones[k]).
        Every reward type.  If the persistent launch gives up (a wavefront waited longer than config['closed_loop_wait_ticks'] for a
        decision: something else holds the GPU's wavefront slots) the env restores the state the chunk started from, takes the
        per-step launches for this and every later call, and warns once."""
        import torch
        name = agent if isinstance(agent, str) else getattr(agent, "__name__", None)
        if name not in self.AGENT_KINDS:
            raise NotImplementedError("run_agent: %r has no device-side version (supported: %s)" % (agent, sorted(self.AGENT_KINDS)))
        shaped = self.reward_type == 'shaped'     # (np.argmax(sigma_pos) of every step travels with the decision / the arg-max slots)
        kind = self.AGENT_KINDS[name]
        e = self._engine
        K = min(int(n_steps), self.n - 1 - self.i)
        if fallback_actions is None:
            fallback_actions = [self.action_space.sample() for _ in range(K + 1)]
        fbh = np.asarray(fallback_actions, dtype=np.int32)[:K + 1]
        assert fbh.size >= K, "one fallback action per decision"
        if fbh.size < K + 1:      # (the decision AFTER the last step is computed too and nobody uses it)
            fbh = np.concatenate([fbh, np.full(K + 1 - fbh.size, -1, dtype=np.int32)])
        fb = torch.as_tensor(fbh).to(e.dev)
        log = torch.full((K + 1,), -1, dtype=torch.int32, device=e.dev)     # log[k] = action of step i0 + k + 1
        actions, rewards, dones = [], [], []
        pos, done = 0, False
        e.launch_agent_select(self.i, self.i, kind, log.data_ptr(), fallback_ptr=fb.data_ptr(), have_prev=self.i >= 1)
        persistent = bool(self._closed_loop_persistent)
        while pos < K and not done:
            # 'trinary' has no early `done`: the whole run is one chunk; otherwise every step of a chunk must stay resident
            # in the history ring, because the step that turns out to be the last one is the state this call returns
            kk = (K - pos) if (persistent and self.reward_type == 'trinary') else min(K - pos, e.H - 1)
            i0 = self.i
            used = False
            if persistent:
                # ONE launch for the kk steps and their kk decisions (ssa_env_closed_loop_f64)
                stats_d = torch.empty((kk, _lib.STAT_STRIDE), dtype=torch.float64, device=e.dev)
                upd_d = torch.empty((kk, _lib.UPD_STRIDE), dtype=torch.float64, device=e.dev)
                snap = e.snapshot_state(i0 % e.H)      # (what a launch that gives up is undone to: ~9 MB device-to-device at 20 000 objects)
                first = log[pos:pos + 1].clone()
                used = e.launch_closed_loop(i0 % e.H, i0 + 1, kind, log[pos:pos + kk + 1], stats_d, upd_d, fallback=fb[pos:pos + kk + 1],
                                            argmax_spos=shaped, wait_ticks=self._loop_wait_ticks, debug_withhold=self._loop_debug_withhold)
                if used:
                    stats = stats_d.cpu().numpy()                      # synchronises the stream
                    upd = upd_d.cpu().numpy()
                    if int(e.loop_error[0]) != 0:
                        # the launch gave up (every wait inside it is bounded): the rings hold a partial chunk.  Back to the state the
                        # chunk started from, and per-step launches from here on -- for this env: whatever took the wavefront slots
                        # (another stream's kernels, another process on the card) may well stay
                        import warnings
                        warnings.warn("ssa_env_closed_loop_f64 gave up (a wavefront waited longer than the bound for a decision); "
                                      "the chunk is re-run with per-step launches, which this env uses from now on", RuntimeWarning)
                        e.restore_state(i0 % e.H, snap)
                        log[pos:pos + 1].copy_(first)
                        log[pos + 1:pos + kk + 1].fill_(-1)
                        self.loop_gave_up = getattr(self, "loop_gave_up", 0) + 1
                        self._closed_loop_persistent = False
                        used = False
                if not used:
                    persistent = False
                    kk = min(kk, e.H - 1)
                del snap
            if not used:
                for k in range(kk):
                    i = i0 + k + 1
                    e.launch_step((i - 1) % e.H, i % e.H, i, actions_ptr=log.data_ptr() + 4 * (pos + k), fast_stats=True, defer_fold=True,
                                  argmax_spos=shaped)
                    if pos + k + 1 < K:     # (the decision for the step after this one)
                        e.launch_agent_select(i, i, kind, log.data_ptr() + 4 * (pos + k + 1), fallback_ptr=fb.data_ptr() + 4 * (pos + k + 1))
                e.flush_stats()
                slots = [(i0 + 1 + k) % e.H for k in range(kk)]
                stats = e.stats[slots, 0].cpu().numpy()          # synchronises the stream
                upd = e.upd[slots, 0].cpu().numpy()
            acts = log[pos:pos + kk].cpu().numpy()
            self._ring_head = i0 + kk
            self._fail_chunk_total = int(stats[-1][_lib.STAT_N_FAILED])
            for k in range(kk):
                self.i += 1
                i, a = self.i, int(acts[k])
                self.actions[i] = a
                self._book_update(i, a, upd[k])
                self._stats = stats[k]
                if int(stats[k][_lib.STAT_N_FAILED]) != self._n_failed:
                    self._record_failures(at_step=i)
                done = self._reward_done(i, a, stats[k], self._argmax_sigma) or (i + 1 >= self.n)
                self._argmax_sigma = int(stats[k][_lib.STAT_ARGMAX_SPOS])
                r = self.rewards[i]
                actions.append(a)
                rewards.append(r if (self.obs_returned == 'flatten' or np.isfinite(r)) else np.float64(0.5))
                dones.append(done)
                if done:
                    break
            pos += kk
        slot = self.i % e.H
        if self.obs_returned == 'aer':
            from .. import device
            M = e.trans[self.i % e.n_time].reshape(3, 3)
            device.aer_obs(e.x_filter[slot], e.P_filter[slot], M, self._consts, out=self._aer_dev.view(self.m, 4))
            self.observation[:] = e.caller_rows(self._aer_dev.view(self.m, 4)).cpu().numpy().reshape(-1)     # (a storage layout is kept here)
            obs = self.observation
        elif self.obs_returned == 'flatten':
            obs = self._host_obs(e.caller_rows(e.obs[slot]).cpu().numpy().reshape(-1))
        else:
            obs = self._host_obs(e.caller_rows(e.obs[slot]).cpu().numpy())
        return obs, np.asarray(actions, dtype=int), np.asarray(rewards), np.asarray(dones, dtype=bool)

    # ------------------------------------------------------------------ closed loop with ANY policy that lives on the GPU
    class PolicyView:
        """what a device-side policy sees at decision time: CUDA tensors of the env's CURRENT state (views of the history slot --
        valid until the next step is launched; nothing is copied, nothing crosses PCIe)."""

        def __init__(self, env, i, tix_off=None):
            self.env, self.i = env, i
            # inside a captured graph the step's time index lives on the DEVICE (engine.env_time0, which the graph advances between
            # replays) and this decision sits `tix_off` steps behind it: the GCRS -> ITRS matrix is then picked by the kernels themselves
            # (ssa_*_at_f64) instead of by a host integer that a capture would freeze
            self._tix_off = tix_off

        # (views are formed on access: a tensor slice costs the host 1-2 us, and most policies read one or two of them)
        obs = property(lambda s: s.env._engine.obs[s.i % s.env._engine.H])            # [m, 12]: x_filter | diag P   (results.py:61)
        x_filter = property(lambda s: s.env._engine.x_filter[s.i % s.env._engine.H])
        P_filter = property(lambda s: s.env._engine.P_filter[s.i % s.env._engine.H])
        x_true = property(lambda s: s.env._engine.x_true[s.i % s.env._engine.H])
        P_filter_prev = property(lambda s: s.env._engine.P_filter[(s.i - 1) % s.env._engine.H] if s.i >= 1 else None)

        def visible(self):
            """uint8 CUDA mask [m]: object_visibility() of the true states (ssa_tasker_simple_2.py:427-434)"""
            from .. import device
            e = self.env._engine
            if self._tix_off is not None:
                return device.visible_mask_at(self.x_true, e.trans, e.env_time0, self._tix_off, self.env._consts)
            return device.visible_mask(self.x_true, e.trans[self.i % e.n_time].reshape(3, 3), self.env._consts)

        def argmax(self, score, mask=None):
            """the policy's arg-max head in ONE launch: np.argmax(score[mask != 0]) mapped back to object indices (first maximum, NaN
            skipped, -1 when nothing is selected) as the int32 CUDA tensor [1] run_policy expects.  torch.argmax + a cast are two launches
            and 16 us at 20 000 objects (profiles/r04_run_policy_timeline.txt); this is 3-4."""
            from .. import device
            env = self.env
            if getattr(env, "_argmax_ws", None) is None or env._argmax_ws_n < score.shape[0]:     # (owned by the env: its launches share a stream)
                env._argmax_ws, env._argmax_ws_n = device.masked_argmax_workspace(score.shape[0], score.device), score.shape[0]
            return device.masked_argmax_action(score, mask, env._argmax_ws)

        def scores(self):
            """(scores[4, m], mask[m]) of the reference's heuristic agents (trace P, visible, log-det ratio, delta_pos)"""
            from .. import device
            e = self.env._engine
            if self._tix_off is not None:
                return device.agent_scores_at(self.x_true, self.x_filter, self.P_filter, self.P_filter_prev, e.trans, e.env_time0, self._tix_off,
                                              self.env._consts)
            return device.agent_scores(self.x_true, self.x_filter, self.P_filter, self.P_filter_prev, e.trans[self.i % e.n_time].reshape(3, 3),
                                       self.env._consts)

    # ---- run_policy as a replayed hipGraph: K x [the policy's kernels + the step launch] captured once, replayed per chunk
    GRAPH_CHUNK = 32

    def _policy_graph(self, policy, K, i0):
        """capture (once per policy / chunk length / history phase) K steps of the closed loop -- for every step the policy's own kernels on
        the current history slot, then the step launch reading the action word the policy produced -- into ONE hipGraph.  What changes
        from replay to replay lives in device memory: the time index (engine.env_time0, advanced by K at the graph's end; the steps
        read env_time0 + their position), the history slots by parity (K is a multiple of the ring depth).  Returns the cache entry
        or None when the policy cannot be captured (it synchronises, allocates outside the graph's pool, ...): the caller enqueues
        eagerly."""
        import torch
        e = self._engine
        key = (id(policy), K, i0 % e.H)
        ent = self._policy_graphs.get(key)
        if ent is not None or key in self._policy_graphs:
            return ent
        stats_d = torch.zeros((K, _lib.STAT_STRIDE), dtype=torch.float64, device=e.dev)
        upd_d = torch.zeros((K, _lib.UPD_STRIDE), dtype=torch.float64, device=e.dev)
        acts_d = torch.full((K,), -1, dtype=torch.int32, device=e.dev)
        shaped = self.reward_type == 'shaped'

        acts_t = []          # the policy's K action tensors: they live in the graph's memory pool, at the same addresses in every replay

        def enqueue():
            for k in range(K):
                i = i0 + k + 1
                a = _action_word(policy(self.PolicyView(self, i - 1, tix_off=k)))
                acts_t.append(a)          # (read by the step below; gathered into acts_d ONCE per replay, behind the graph)
                e.launch_step((i - 1) % e.H, i % e.H, k + 1, actions_ptr=a.data_ptr(), fast_stats=True, defer_fold=True,
                              stats_out=stats_d[k].data_ptr(), upd_out=upd_d[k].data_ptr(), argmax_spos=shaped)
            e.flush_stats()
            torch.cat([a.reshape(1) for a in acts_t], out=acts_d)
            e.env_time0.add_(K)
        if getattr(self, "_policy_streams", None) is None:      # capture / replay stream and the copy stream of the pipelined chunks: one pair per env
            self._policy_streams = (torch.cuda.Stream(device=e.dev), torch.cuda.Stream(device=e.dev))
        stream, copy_stream = self._policy_streams
        # two sets of pinned host buffers for the replay's results (statistics, update records, actions): chunk c is booked from one while
        # the copy behind replay c + 1 fills the other
        hosts = tuple(tuple(torch.empty(d.shape, dtype=d.dtype, pin_memory=True) for d in (stats_d, upd_d, acts_d)) for _ in range(2))
        g = torch.cuda.CUDAGraph()
        ok = True
        try:
            policy(self.PolicyView(self, i0))      # (eagerly once, result unused: lazy initialisation must not happen inside the capture)
            # no garbage collection inside the capture: a collected CUDAGraph of an env that went out of scope is DESTROYED by its finaliser,
            # hipGraphDestroy is not permitted while a stream captures, and the error thrown from that destructor ends the process
            # (seen once in the GPU suite under -s; torch.cuda.graph() collects before it captures for the same reason)
            import gc
            gc.collect()
            torch.cuda.synchronize()
            gc_was_on = gc.isenabled()
            gc.disable()
            try:
                with torch.cuda.stream(stream):
                    g.capture_begin(capture_error_mode="thread_local")
                    try:
                        enqueue()
                        g.capture_end()
                    except BaseException:
                        try:
                            g.capture_end()
                        except Exception:  # noqa: BLE001
                            pass
                        raise
            finally:
                if gc_was_on:
                    gc.enable()
        except TypeError:
            raise
        except Exception as exc:  # noqa: BLE001  (not capture-safe: remembered, the eager loop takes over)
            ok = False
            self.policy_graph_error = repr(exc)
            e._fold_pending = None
            _never_destroy(g)
        torch.cuda.current_stream().wait_stream(stream)
        ent = (g, stats_d, upd_d, acts_d, stream, hosts, copy_stream) if ok else None
        self._policy_graphs[key] = ent
        return ent

    def run_policy(self, policy, n_steps, graph='auto'):
        """Closed loop with an ARBITRARY policy evaluated on the GPU (a torch module, a hand-written rule):
            a = policy(view)          # view: SSA_Tasker_Env.PolicyView -- CUDA tensors; returns an int32 (or int64: torch.argmax) CUDA tensor [1]
            step(a)
        repeated n_steps times with NO host round trip: the action never leaves the device (the step kernel reads it from the
        tensor the policy returned), the statistics and update records go to device rings, ONE synchronisation at the end, then the
        env's bookkeeping (actions, rewards, dones, failures, z_true / y / S records) is filled in as step() would have.  The
        reference's loop `a = agent(obs, env); env.step(a)` (run_environment.py:26-29) for agents that are not one of the built-in
        greedy ones (those: run_agent, one persistent launch).  Every reward type; a data-dependent `done` ('jones', 'shaped') is
        honoured at the bookkeeping -- the steps launched behind it are discarded (chunks of history - 1 steps, as run_agent).
        Returns (actions[k], rewards[k], dones[k])."""
        self._caller_order()          # (the policy's views are the env's own object order)
        import torch
        shaped = self.reward_type == 'shaped'
        e = self._engine
        K = min(int(n_steps), self.n - 1 - self.i)
        actions, rewards, dones = [], [], []
        pos, done = 0, False
        # graph = 'auto' | True: chunks of GRAPH_CHUNK steps replayed from a captured hipGraph where that is possible -- a reward without a
        # data-dependent `done` ('trinary': every step of the call is wanted), a history ring whose depth divides the chunk, a policy
        # that can be captured; everything else (and graph = False) takes the eager loop below, step by step from the host
        G = self.GRAPH_CHUNK
        use_graph = bool(graph) and self.reward_type == 'trinary' and G % e.H == 0
        if not hasattr(self, "_policy_graphs"):
            self._policy_graphs, self.policy_graph_error = {}, None
        # The chunks are PIPELINED: replay c + 1 is enqueued before the host books chunk c.  A replay writes its statistics / update records /
        # actions at fixed device addresses, so right behind every replay a copy stream moves them into one of two pinned host buffers
        # (12 KB), and the next replay waits for that copy alone; the host then fills in chunk c's bookkeeping (5 us per step) while the
        # GPU runs chunk c + 1.  (Round 4 measurement, profiles/r04_run_policy_timeline.txt: inside a replay the GPU idles < 1 us between
        # kernels, but synchronise - copy - book - replay left it idle for 16 us per step at chunk boundaries.)  An invalid action is
        # therefore reported one chunk late: the steps enqueued behind it have run (with no update: the kernel ignores an action out of range).
        pend = None            # (i0, host arrays, event) of the replay whose bookkeeping is outstanding
        state = {"done": False}

        def book(i0, host, ev):
            ev.synchronize()
            stats, upd, acts = host
            self._ring_head = i_start + launched
            self._fail_chunk_total = int(stats[-1][_lib.STAT_N_FAILED])
            for k in range(G):
                self.i += 1
                i, a = self.i, int(acts[k])
                if not (0 <= a < self.m):
                    raise ValueError("run_policy: the policy chose action %d at step %d (valid: 0 .. %d)" % (a, i, self.m - 1))
                self.actions[i] = a
                self._book_update(i, a, upd[k])
                self._stats = stats[k]
                if int(stats[k][_lib.STAT_N_FAILED]) != self._n_failed:
                    self._record_failures(at_step=i)
                state["done"] = self._reward_done(i, a, stats[k], self._argmax_sigma) or (i + 1 >= self.n)
                self._argmax_sigma = int(stats[k][_lib.STAT_ARGMAX_SPOS])
                actions.append(a)
                rewards.append(self.rewards[i] if (self.obs_returned == 'flatten' or np.isfinite(self.rewards[i])) else np.float64(0.5))
                dones.append(state["done"])
        launched, i_start, gstream = 0, self.i, None      # steps enqueued by replays (self.i follows as the chunks are booked)
        try:
            while use_graph and K - launched >= G and not state["done"]:
                i0 = i_start + launched
                ent = self._policy_graph(policy, G, i0)
                if ent is None:
                    break
                g, stats_d, upd_d, acts_d, gstream, hosts, copy_stream = ent
                if launched == 0:
                    e.flush_stats()
                    e.env_time0.fill_(i0)
                    gstream.wait_stream(torch.cuda.current_stream())
                slot = (launched // G) % 2
                with torch.cuda.stream(gstream):
                    g.replay()
                    ready = torch.cuda.Event()
                    ready.record(gstream)
                copy_stream.wait_event(ready)
                with torch.cuda.stream(copy_stream):
                    for h, d in zip(hosts[slot], (stats_d, upd_d, acts_d)):
                        h.copy_(d, non_blocking=True)
                    copied = torch.cuda.Event()
                    copied.record(copy_stream)
                gstream.wait_event(copied)         # the NEXT replay overwrites the device buffers only behind this copy
                launched += G
                if pend is not None:
                    book(*pend)                    # chunk c - 1, while the GPU runs chunk c
                pend = (i0, tuple(h.numpy() for h in hosts[slot]), copied)
            if pend is not None:
                book(*pend)
        finally:
            if gstream is not None:
                torch.cuda.current_stream().wait_stream(gstream)
                e.env_time0.zero_()
        pos, done = launched, state["done"]
        while pos < K and not done:
            kk = (K - pos) if self.reward_type == 'trinary' else min(K - pos, e.H - 1)
            i0 = self.i
            stats_d = torch.empty((kk, _lib.STAT_STRIDE), dtype=torch.float64, device=e.dev)
            upd_d = torch.empty((kk, _lib.UPD_STRIDE), dtype=torch.float64, device=e.dev)
            acts_d = []
            for k in range(kk):
                i = i0 + k + 1
                a = _action_word(policy(self.PolicyView(self, i - 1)))
                acts_d.append(a)          # (kept alive until the launches that read it have run)
                e.launch_step((i - 1) % e.H, i % e.H, i, actions_ptr=a.data_ptr(), fast_stats=True, defer_fold=True,
                              stats_out=stats_d[k].data_ptr(), upd_out=upd_d[k].data_ptr(), argmax_spos=shaped)
                self.i = i                # (the view of the next decision indexes the history by it)
            e.flush_stats()
            stats = stats_d.cpu().numpy()                  # synchronises the stream
            upd = upd_d.cpu().numpy()
            acts = torch.cat([a.reshape(1) for a in acts_d]).cpu().numpy()
            self.i = i0
            self._ring_head = i0 + kk
            self._fail_chunk_total = int(stats[-1][_lib.STAT_N_FAILED])
            for k in range(kk):
                self.i += 1
                i, a = self.i, int(acts[k])
                if not (0 <= a < self.m):
                    raise ValueError("run_policy: the policy chose action %d at step %d (valid: 0 .. %d)" % (a, i, self.m - 1))
                self.actions[i] = a
                self._book_update(i, a, upd[k])
                self._stats = stats[k]
                if int(stats[k][_lib.STAT_N_FAILED]) != self._n_failed:
                    self._record_failures(at_step=i)
                done = self._reward_done(i, a, stats[k], self._argmax_sigma) or (i + 1 >= self.n)
                self._argmax_sigma = int(stats[k][_lib.STAT_ARGMAX_SPOS])
                r = self.rewards[i]
                actions.append(a)
                rewards.append(r if (self.obs_returned == 'flatten' or np.isfinite(r)) else np.float64(0.5))
                dones.append(done)
                if done:
                    break
            pos += kk
        return np.asarray(actions, dtype=int), np.asarray(rewards), np.asarray(dones, dtype=bool)

    # ------------------------------------------------------------------ failures (:369-382)
    def _record_failures(self, at_step=None):
        """filter_error() bookkeeping (:369-382) for the filters that failed in step self.i (at_step: the step being booked after a rollout
        / closed-loop / policy launch that ran several).  The KERNEL wrote the records -- object, status code, step, error_failed() of the
        state it failed from -- into host-mapped memory (ssa_step_params.fail_log) as the filters failed: nothing is copied here.  The
        reference loses 2-3 % of its filters over an episode (and so does this env's default, the behaviour-faithful variant): a few per
        step late in an episode."""
        s = time.time()
        e = self._engine
        step = self.i if at_step is None else at_step
        total = int(self._stats[_lib.STAT_N_FAILED])               # filters failed by the END of `step`
        # records appended since the last call (a multi-step launch appends in the order its wavefronts reach the failures, not by step)
        n_dev = max(total, self._fail_read) if at_step is None else self._fail_chunk_total
        while self._fail_read < n_dev:
            r = e.fail_log[self._fail_read]
            if r[_lib.FAIL_TIME] == 0.0:         # (not written: a status word set from outside the kernels is counted in the statistics but has no record)
                break
            self._fail_pending.setdefault(int(r[_lib.FAIL_TIME]), []).append(r.copy())
            self._fail_read += 1
        for t in sorted(k for k in self._fail_pending if k <= step):
            for r in self._fail_pending.pop(t):
                j = int(r[_lib.FAIL_OBJ])
                self.failed_filters_msg.record(j, t, int(r[_lib.FAIL_STATUS]), r[_lib.FAIL_ERR:_lib.FAIL_ERR + 4])
                self.failed_filters_id.append(j)
        self._n_failed = len(self.failed_filters_id)
        self.runtime['filter_error'] += time.time() - s

    def _latest_resident(self):
        """the newest step the device history holds (a rollout / closed-loop launch has advanced the rings beyond self.i while the
        host is still booking its steps one by one)"""
        head = getattr(self, "_ring_head", None)
        return self.i if head is None else head

    def anees(self):
        """:436-446 -- average normalised estimation error squared over the episode so far: NEES on the device for every
        resident (step, object) of the history (the reference loops n * m numpy inversions); fills self.nees (n, m)."""
        self._caller_order()
        from .. import device
        s = time.time()
        e = self._engine
        self.nees = np.full((self.n, self.m), np.nan)
        lo = max(0, self.i - e.H + 1)
        for i in range(lo, self.i + 1):
            sl = i % e.H
            self.nees[i] = device.nees(e.x_true[sl], e.x_filter[sl], e.P_filter[sl]).cpu().numpy()
        self.runtime['anees'] += time.time() - s
        return float(np.mean(self.nees[lo:self.i + 1]))

    def nis(self):
        """normalised innovation squared of every update taken so far (fitness_test(), :750-754); NaN where no update ran"""
        import torch
        from .. import device
        out = np.full(self.n, np.nan)
        k = np.where(self.obs_taken[:self.i + 1])[0]
        if len(k):
            out[k] = device.nis(torch.as_tensor(self._y[k]).to("cuda"), torch.as_tensor(self._S_sel[k]).to("cuda")).cpu().numpy()
        return out

    # two-sided chi-square critical points stats.chi2.ppf([alpha / 2, 1 - alpha / 2], df) for the reference's alpha = 0.05
    # (:756, :762); other alphas need scipy
    _CHI2_95 = {3: (0.21579528262389788, 9.348403604496145), 6: (1.2373442457912032, 14.449375335447922)}

    @classmethod
    def _chi2_points(cls, alpha, df):
        if abs(alpha - 0.05) < 1e-15 and df in cls._CHI2_95:
            return cls._CHI2_95[df]
        from scipy import stats
        lo, hi = stats.chi2.ppf([alpha / 2, 1 - alpha / 2], df=df)
        return float(lo), float(hi)

    def fitness_chi2(self, alpha=0.05):
        """Tests 2 and 4 of fitness_test() (:750-775): the percentage of normalised innovations squared (NaN dropped, :757)
        and of normalised estimation errors squared (NaN kept in the mean, :771) inside the two-sided (1 - alpha) chi-square
        interval, counted ON THE DEVICE (ssa_nis_f64 / ssa_nees_f64 + ssa_chi2_contained_f64) over the steps simulated so
        far that are still resident.  The reference's Test 4 covers the WHOLE episode (its history arrays hold every step): build the
        env with config['history'] = 'full' for that -- with a short ring (history = 2, what 'auto' picks for very large envs) the NEES
        window is the last `history` steps only, and 'nees_steps' in the result says how many it was.
        Returns {'Test 2: NIS chi2': pct, 'Test 4: NEES chi2': pct, counts...}."""
        self._caller_order()
        import torch
        from .. import device
        e = self._engine
        out = {}
        k = np.where(self.obs_taken[:self.i + 1])[0]
        lo, hi = self._chi2_points(alpha, 3)
        if len(k):
            nis = device.nis(torch.as_tensor(self._y[k]).to("cuda"), torch.as_tensor(self._S_sel[k]).to("cuda"))
            inside, valid = device.chi2_contained(nis, lo, hi)
        else:
            inside, valid = 0, 0
        out['Test 2: NIS chi2'] = round(100.0 * inside / valid, 2) if valid else float('nan')
        out['nis_inside'], out['nis_valid'] = inside, valid
        first = max(0, self.i - e.H + 1)
        slots = [i % e.H for i in range(first, self.i + 1)]
        if slots == list(range(slots[0], slots[0] + len(slots))):     # contiguous in the history tensors: no gather
            sl = slice(slots[0], slots[0] + len(slots))
            xt, x, P = e.x_true[sl], e.x_filter[sl], e.P_filter[sl]
        else:
            xt, x, P = e.x_true[slots], e.x_filter[slots], e.P_filter[slots]
        nees = device.nees(xt.reshape(-1, 6), x.reshape(-1, 6), P.reshape(-1, 6, 6))
        lo, hi = self._chi2_points(alpha, 6)
        inside, _ = device.chi2_contained(nees, lo, hi)
        out['Test 4: NEES chi2'] = round(100.0 * inside / nees.numel(), 2)
        out['nees_inside'], out['nees_total'], out['nees_steps'] = inside, int(nees.numel()), len(slots)
        return out

    def failed_filters(self):
        if not self.failed_filters_id:
            print("No failed Objects")
        else:
            print("Failed Objects: ", self.failed_filters_id)
            for rso_id in self.failed_filters_id:
                print(self.failed_filters_msg[rso_id])

    # ------------------------------------------------------------------ visibility (:410-434)
    def _mask(self):
        from .. import device
        self._caller_order()
        e = self._engine
        M = e.trans[self.i % e.n_time].reshape(3, 3)
        return device.visible_mask(e.x_true[self.i % e.H], M, self._consts).cpu().numpy().astype(bool)

    def visible_objects(self):
        s = time.time()
        viz = np.where(self._mask())[0]
        self.runtime['visible_objects'] += time.time() - s
        return viz

    def object_visible(self, RSO_ID=[]):
        if len(RSO_ID) == 0:
            print('RSO ID expected, but not supplied')
            return RSO_ID
        return self._mask()[np.asarray(RSO_ID)]

    def object_visibility(self):
        s = time.time()
        viz = self._mask()
        self.runtime['object_visibility'] += time.time() - s
        return viz

    def agent_scores(self):
        """device tensors (scores[4, m], mask[m]) for the heuristic agents (ssa_gym_amd.agents)."""
        from .. import device
        self._caller_order()
        e = self._engine
        cur, prev = self.i % e.H, (self.i - 1) % e.H
        M = e.trans[self.i % e.n_time].reshape(3, 3)
        P_prev = e.P_filter[prev] if self.i >= 1 else None
        return device.agent_scores(e.x_true[cur], e.x_filter[cur], e.P_filter[cur], P_prev, M, self._consts)

    def aer_obs(self, obs):
        """:834-840 -- [az, el, range, trace(P)] per object, NaN/inf -> 0.001."""
        from .. import device
        e = self._engine
        slot = self.i % e.H
        M = e.trans[self.i % e.n_time].reshape(3, 3)
        out = e.caller_rows(device.aer_obs(e.x_filter[slot], e.P_filter[slot], M, self._consts).view(self.m, 4)).cpu().numpy().reshape(-1)
        obs[:] = out
        return obs

    def render(self, mode='live'):
        raise NotImplementedError("rendering/plots are outside the hot-path scope (SURVEY section 2, rows 1b/6b)")
