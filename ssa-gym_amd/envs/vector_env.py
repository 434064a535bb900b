"""SSA_Tasker_VecEnv: E independent copies of SSA_Tasker_Env advanced by ONE kernel launch per step
(BASELINE config 5: "20 000 objects x 64 parallel env instances (RLlib vectorised rollout)").

The reference reaches env-level parallelism with one Python process per env (RLlib `num_workers`,
RLLib_training.py:40-56).  Here the E environments are one batch of E*m objects in HBM; the step
kernel takes per-env actions and per-env time indices, so envs that terminated are reset in place
while the others continue (the usual vector-env auto-reset).  Semantics per env are those of
SSA_Tasker_Env (same reward / done logic, same RNG draw order for reset, seed = base seed + env index);
measurement noise is drawn on the device per (env, time step) -- only the object the action selects
consumes noise in a step (ssa_tasker_simple_2.py:301).
"""
import numpy as np

from .. import _lib, host
from . import transformations
from ._config import kernel_consts
from ._gymshim import np_random, spaces


class SSA_Tasker_VecEnv:
    def __init__(self, config, num_envs, seed=0):
        import torch
        from .. import engine
        self.E, self.m, self.n, self.dt = int(num_envs), config['rso_count'], config['steps'], config['time_step']
        self._bulk_draws = bool(config.get('device_rng', False))
        self.obs_returned, self.reward_type = config['obs_returned'], config['reward_type']
        self.orbits = config['orbits']
        self.x_sigma = np.array(config['x_sigma'])
        self.obs_type = config['obs_type']
        self.z_sigma = (config['z_sigma'] * np.array([host.arcsec2rad, host.arcsec2rad, 1]) if self.obs_type == 'aer'
                        else np.asarray(config['z_sigma'], dtype=np.float64))
        self.P_0 = np.diag(self.x_sigma ** 2) if config['P_0'] is None else np.copy(config['P_0'])
        R = np.diag(self.z_sigma ** 2) if config['R'] is None else np.copy(config['R'])
        Q = host.Q_discrete_white_noise(dim=2, dt=self.dt, var=config['q_sigma'] ** 2, block_size=3, order_by_dim=False)
        obs_lla = np.array(config['observer']) * [host.deg2rad, host.deg2rad, 1]
        self._consts, model = kernel_consts(config, Q, R, self.dt, np.radians(config['obs_limit']), obs_lla)   # as SSA_Tasker_Env
        trans = (np.asarray(config['trans_matrix']) if config.get('trans_matrix') is not None
                 else transformations.trans_matrix_table(config['t_0'], self.dt, self.n))
        self._gen = torch.Generator(device="cuda").manual_seed(int(seed))
        self._zs = torch.as_tensor(self.z_sigma, dtype=torch.float64, device="cuda")
        z = torch.randn((self.E, self.n, 1, 3), dtype=torch.float64, device="cuda", generator=self._gen) * self._zs
        self._eng = engine.HotPathEngine(self._consts, self.m, self.E, trans, z, history=2,
                                         zn_stride_env=self.n * 3, zn_stride_time=3, zn_stride_obj=0)
        self._rng = [np_random(seed + e)[0] for e in range(self.E)]
        self.single_action_space = spaces.Discrete(self.m)
        shp = {'flatten': (self.m * 12,), 'aer': (self.m * 4,)}.get(self.obs_returned, (self.m, 12))
        self.single_observation_space = spaces.Box(low=np.full(shp, -np.inf), high=np.full(shp, np.inf), dtype=np.float64)
        self.num_envs = self.E
        self._aer = torch.zeros((self.E * self.m, 4), dtype=torch.float64, device="cuda")
        # host side of a step: time indices and actions leave from pinned staging (asynchronous copies), the statistics and the
        # observation vectors arrive in host-mapped pinned memory written by the kernels themselves (the 'aer' block by the step
        # kernel's epilogue, the observation rows as its second destination): one stream synchronisation per vector step, no
        # device-to-host copy pass.  Two observation buffers alternate: what step k returned stays intact until step k + 2.
        self._ta_host = torch.zeros(2 * self.E, dtype=torch.int32).pin_memory()      # [time indices | actions]: one copy per step
        self._time_np, self._act_np = self._ta_host.numpy()[:self.E], self._ta_host.numpy()[self.E:]
        # up to 8 envs: time indices and actions travel BY VALUE in the launch's parameter block (no copy in front of the step, and
        # with no torch call left in step() the stream handle is looked up once); the statistics of every env are folded by the last
        # wavefront that adds to them (SSA_LAUNCH_FOLD_INSIDE): ONE launch per vector step
        self._inline = self.E <= _lib.INLINE_ENVS
        self._stream = torch.cuda.current_stream()
        # config['obs_device'] = True (opt-in, for policies that live on the GPU): step() returns the observations as ONE CUDA tensor
        # [E, ...] -- a view of device memory the step kernel wrote -- and nothing but the statistics crosses PCIe
        self._obs_device = bool(config.get('obs_device', False))
        # step() returns a FRESH array (gym's vector envs copy their observation buffer by default, and so does the reference's single
        # env for 'flatten'); config['obs_zero_copy'] = True: a view of the host-mapped ring the kernel writes, valid until step k + 2
        self._obs_zero_copy = bool(config.get('obs_zero_copy', False))
        self._stats_host = torch.zeros((self.E, _lib.STAT_STRIDE), dtype=torch.float64).pin_memory()
        self._stats_np = self._stats_host.numpy()
        per = self.m * (4 if self.obs_returned == 'aer' else 12)
        oshape = (self.E, self.m, 12) if self.obs_returned not in ('aer', 'flatten') else (self.E, per)
        # config['obs_dtype'] = 'float32' (EXTENSION; default float64, the reference's): the host-facing observations in single precision,
        # written that way by the step kernel (SSA_LAUNCH_MIRROR_F32) -- half of the 5 MB a vector step sends over PCIe
        self._mirror_f32 = np.dtype(config.get('obs_dtype', np.float64)) == np.float32 and not self._obs_device
        if self._mirror_f32 and self.reward_type == 'shaped' and not self._eng.supports_argmax:
            raise ValueError("obs_dtype float32 with the 'shaped' reward needs rso_count % 4 == 0 (the one-launch statistics path)")
        self._obs_ring = [torch.zeros(self.E * per, dtype=torch.float32 if self._mirror_f32 else torch.float64).pin_memory() for _ in range(2)]
        self._obs_ring_np = [b.numpy().reshape(oshape) for b in self._obs_ring]
        self._obs_ring_ptr = [b.data_ptr() for b in self._obs_ring]
        # default hand-out: a buffer nobody holds, written by the kernel, returned as a fresh array and taken back when the consumer lets go
        # of it (envs/_obspool.py) -- fresh-array semantics without the 5 MB copy per step
        from ._obspool import ObsPool
        self._obs_pool = None if (self._obs_zero_copy or self._obs_device) else ObsPool(self.E * per, oshape, cap=int(config.get('obs_pool', 16)),
                                                                                            dtype=np.float32 if self._mirror_f32 else np.float64)
        # config['storage_layout'] = 'regime' (opt-in): every env's objects stored sorted by orbit regime (catalogue.regime_order_env;
        # HotPathEngine.set_layout with one permutation per env) -- actions, rewards and observations stay in the env's own numbering.  It pays
        # where the observations stay on the GPU (obs_device): host-facing rows would leave the kernel one by one instead of tile by tile
        self._layout = config.get('storage_layout', None)
        if self._layout not in (None, 'regime'):
            raise ValueError("storage_layout: None or 'regime'")
        if self._layout and self.m % 4:
            raise ValueError("storage_layout with several envs needs rso_count % 4 == 0")
        self._obs_dev_rows = None       # (layout + obs_device, 'flatten' / rows: the step kernel's second copy of the observation, at the caller's rows)
        self.i = np.zeros(self.E, dtype=np.int64)       # per-env step index
        self.tick = 0
        self.rewards_sum = np.zeros(self.E)
        self._argmax_prev = np.zeros(self.E, dtype=np.int64)
        self.reset()

    # ------------------------------------------------------------------
    def _draw(self, e):
        rs, N = self._rng[e], self.orbits.shape[0]
        if self._bulk_draws:      # config['device_rng']: bulk draws, 1 ms instead of 40 per env at m = 20 000 (auto-reset cost)
            xt = self.orbits[rs.randint(low=0, high=N, size=self.m)]
            return xt, xt + rs.normal(size=(self.m, 6)) * self.x_sigma
        xt = np.empty((self.m, 6))
        noise = np.empty((self.m, 6))
        for j in range(self.m):   # reset() draw order of the reference (:206-209)
            xt[j] = self.orbits[rs.randint(low=0, high=N), :]
            noise[j] = rs.normal(size=6) * self.x_sigma
        return xt, xt + noise

    def _reset_env(self, e, slot, draw=None):
        import torch
        xt, xf = self._draw(e) if draw is None else draw
        if self._layout and draw is None:        # (reset() of all envs has set the whole table already)
            from ..catalogue import regime_order_env
            self._eng.set_env_layout(e, regime_order_env(xt, e, self.E))
        self._eng.load_env_state(slot, e, xt, xf, self.P_0)
        self._eng.z_noise[e].copy_(torch.randn((self.n, 1, 3), dtype=torch.float64, device="cuda", generator=self._gen) * self._zs)
        self.i[e] = 0
        self.rewards_sum[e] = 0.0

    def reset(self):
        slot = self.tick % 2
        draws = [self._draw(e) for e in range(self.E)]
        if self._layout:
            from ..catalogue import regime_order_env
            self._eng.set_layout(np.stack([regime_order_env(draws[e][0], e, self.E) for e in range(self.E)]))
        for e in range(self.E):
            self._reset_env(e, slot, draw=draws[e])
        self._refresh_stats(slot)
        return self._obs(slot, reset=True)

    def _refresh_stats(self, slot):
        st = self._eng.stats[slot].cpu().numpy()
        self._argmax_prev = st[:, _lib.STAT_ARGMAX_SPOS].astype(np.int64)
        return st

    def _obs(self, slot, reset=False):
        e = self._eng
        if self._obs_device:
            if self.obs_returned == 'aer':
                if reset:
                    self._aer_reset_rows()
                return self._aer.view(self.E, self.m * 4)
            rows = e.obs[slot]
            if self._layout:       # the kernel's second copy, at the caller's rows (a reset: gathered from the state just loaded); one per
                if self._obs_dev_rows is None:      # history slot, so that what step k returned stays intact until step k + 2
                    import torch
                    self._obs_dev_rows = [torch.zeros_like(e.obs[0]) for _ in range(2)]
                if reset:
                    self._obs_dev_rows[slot].copy_(e.caller_rows(e.obs[slot]))
                rows = self._obs_dev_rows[slot]
            return rows.view(self.E, self.m * 12) if self.obs_returned == 'flatten' else rows.view(self.E, self.m, 12)
        cast = (lambda a: a.astype(np.float32)) if self._mirror_f32 else (lambda a: a)
        if self.obs_returned == 'flatten':
            return cast(e.caller_rows(e.obs[slot]).cpu().numpy().reshape(self.E, self.m * 12))
        if self.obs_returned == 'aer':
            if reset:
                self._aer_reset_rows()
            return cast(self._aer.cpu().numpy().reshape(self.E, self.m * 4))
        return cast(e.caller_rows(e.obs[slot]).cpu().numpy().reshape(self.E, self.m, 12))

    def _aer_reset_rows(self):
        """the 'aer' block of the CURRENT state of every env (reset time only: a step's block is the step kernel's epilogue)"""
        from .. import device
        e, slot = self._eng, self.tick % 2
        for k in range(self.E):
            sl = slice(k * self.m, (k + 1) * self.m)
            M = e.trans[int(self.i[k]) % e.n_time].reshape(3, 3)
            device.aer_obs(e.x_filter[slot, sl], e.P_filter[slot, sl], M, self._consts, out=self._aer[sl])
            if self._layout:
                self._aer[sl].copy_(e.env_caller_rows(k, self._aer[sl]))

    def step(self, actions):
        import torch
        actions = np.asarray(actions, dtype=np.int64).reshape(self.E)
        assert 0 <= int(actions.min()) and int(actions.max()) < self.m, "invalid action"
        e = self._eng
        argmax_prev = self._argmax_prev
        self.i += 1
        self.tick += 1
        sin, sout = (self.tick - 1) % 2, self.tick % 2
        aer = self.obs_returned == 'aer'
        k = self.tick % 2
        shaped = self.reward_type == 'shaped'
        # 'shaped' needs np.argmax(sigma_pos[i - 1]) per env: from the arg-max slots of the one-launch path when every env is whole
        # tiles (rso_count % 4 == 0), through the three-launch exact statistics otherwise
        fast = (not shaped) or e.supports_argmax
        if self._obs_device:
            aer_out, mirror = (self._aer.data_ptr() if aer else 0), (self._obs_dev_rows[sout].data_ptr() if (self._layout and not aer) else 0)
        else:
            kp = self._obs_pool.acquire() if self._obs_pool is not None else None
            dst = self._obs_pool.ptrs[kp] if kp is not None else self._obs_ring_ptr[k]
            aer_out, mirror = (dst if aer else 0), (0 if aer else dst)
        if self._inline:
            cur = self._stream
            e.launch_step(sin, sout, 0, aer_out=aer_out, obs_mirror=mirror, stats_out=self._stats_host.data_ptr(), stream=cur.cuda_stream,
                          fast_stats=fast, fold_inside=True, env_words=(self.i.tolist(), actions.tolist()), argmax_spos=shaped and fast,
                          mirror_f32=self._mirror_f32 and fast)
        else:
            self._time_np[:] = self.i
            self._act_np[:] = actions
            e.time_actions.copy_(self._ta_host, non_blocking=True)
            cur = torch.cuda.current_stream()     # (the stream the time / action copy above was enqueued in)
            e.launch_step(sin, sout, 0, aer_out=aer_out, obs_mirror=mirror, stats_out=self._stats_host.data_ptr(), stream=cur.cuda_stream,
                          fast_stats=fast, fold_inside=True, argmax_spos=shaped and fast, mirror_f32=self._mirror_f32 and fast)
        cur.synchronize()
        st = self._stats_np            # (host-mapped: the step kernel's folds wrote it; stable until the next launch)
        if shaped:
            self._argmax_prev = st[:, _lib.STAT_ARGMAX_SPOS].astype(np.int64)
        mx = st[:, _lib.STAT_MAX_DPOS]
        last = self.i + 1 >= self.n
        if self.reward_type == 'trinary':
            rewards = (st[:, _lib.STAT_CNT_LT_1E4] + st[:, _lib.STAT_CNT_LT_1E7]) / self.m / 2
            dones = last
        elif self.reward_type == 'jones':
            lost, won = mx > 5e6, mx < 3e4
            dones = lost | won | last
            rewards = np.zeros(self.E)
            rewards[won & ~lost] = 1.0
        elif self.reward_type == 'shaped':
            lost, won = mx > 5e6, mx < 3e4
            hit = actions == argmax_prev
            rewards = np.where(hit, 1.0 / self.n, -1.0 / self.n)
            rewards[won] = 1.0 - self.rewards_sum[won]
            rewards[lost] = 0.0
            dones = lost | won | last
        else:
            rewards, dones = np.zeros(self.E), np.zeros(self.E, dtype=bool)
        self.rewards_sum += rewards
        if self._obs_device:
            obs = self._obs(sout)
        elif self._obs_zero_copy:
            obs = self._obs_ring_np[k]
        elif kp is not None:
            obs = self._obs_pool.hand_out(kp)       # fresh array, no copy
        else:
            obs = self._obs_ring_np[k].copy()       # (more than `obs_pool` observations alive at once)
        infos = [{} for _ in range(self.E)]
        if dones.any():   # auto-reset in place; the returned observation of a finished env is its new first one
            for d in np.where(dones)[0]:
                infos[d]['terminal_observation'] = obs[d].clone() if self._obs_device else obs[d].copy()
                self._reset_env(int(d), sout)
            st_dev = self._eng.stats[sout].cpu().numpy()          # (the reset wrote the new envs' statistics on the device)
            self._argmax_prev[dones] = st_dev[dones, _lib.STAT_ARGMAX_SPOS].astype(np.int64)
            obs = self._obs(sout, reset=(self.obs_returned == 'aer') or bool(self._layout))
        # (ssa_tasker_simple_2.py:365-367 passes the reward of the other modes through nan_to_num(nan=.5, inf=.5): the rewards formed above are
        # finite by construction -- counts of comparisons, constants -- so there is nothing for it to replace)
        return obs, rewards, dones, infos

    # inspection helpers (per env)
    def P_filter(self, e):
        return self._eng.env_caller_rows(e, self._eng.P_filter[self.tick % 2, e * self.m:(e + 1) * self.m]).cpu().numpy()

    def x_filter(self, e):
        return self._eng.env_caller_rows(e, self._eng.x_filter[self.tick % 2, e * self.m:(e + 1) * self.m]).cpu().numpy()

    def x_true(self, e):
        return self._eng.env_caller_rows(e, self._eng.x_true[self.tick % 2, e * self.m:(e + 1) * self.m]).cpu().numpy()
