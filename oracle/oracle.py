"""ctypes front-end to the C parity oracle (oracle/ssa_oracle.c).

TEST INFRASTRUCTURE ONLY: may be imported by tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg -- never by the product package.

`Oracle(long_double=False)` is the fp64 restatement in the reference's order of
operations; `Oracle(long_double=True)` runs the same formulas in x87 80-bit
arithmetic (the conditioning witness).  All arrays are C-contiguous float64.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)

ST_OK, ST_PREDICT_NAN, ST_PREDICT_LINALG, ST_UPDATE_NAN, ST_UPDATE_LINALG = range(5)
X_FAILED = np.array([1e20, 1e20, 1e20, 1e12, 1e12, 1e12])


def build(force=False):
    """compile both oracle libraries with gcc (oracle/Makefile)."""
    if force:
        subprocess.check_call(["make", "-C", _HERE, "clean"], stdout=subprocess.DEVNULL)
    subprocess.check_call(["make", "-C", _HERE], stdout=subprocess.DEVNULL)


def merwe_weights(alpha, beta, kappa, n=6):
    """filterpy MerweScaledSigmaPoints._compute_weights.
    Returns (Wm[13], Wc[13], n + lambda)."""
    lambda_ = alpha ** 2 * (n + kappa) - n
    c = .5 / (n + lambda_)
    Wc = np.full(2 * n + 1, c)
    Wm = np.full(2 * n + 1, c)
    Wc[0] = lambda_ / (n + lambda_) + (1 - alpha ** 2 + beta)
    Wm[0] = lambda_ / (n + lambda_)
    return Wm, Wc, lambda_ + n


def _a(x):
    return np.ascontiguousarray(x, dtype=np.float64)


def _p(a):
    return a.ctypes.data_as(_dp)


class Oracle:
    def __init__(self, long_double=False, omp=False):
        name = "libssa_oracle_ld.so" if long_double else ("libssa_oracle_omp.so" if omp else "libssa_oracle.so")
        path = os.path.join(_HERE, name)
        if not os.path.exists(path):
            build()
        self.lib = C.CDLL(path)
        self.long_double = bool(self.lib.orc_is_long_double())
        L = self.lib
        L.orc_propagate.argtypes = [_dp, _dp, C.c_long, C.c_double]
        L.orc_kepler_intermediates.argtypes = [_dp, _dp, C.c_long, C.c_double]
        L.orc_robust_cholesky.argtypes = [_dp, _dp, _ip]
        L.orc_sigma_points.argtypes = [_dp, _dp, C.c_double, _dp]
        L.orc_hx_aer.argtypes = [_dp, _dp, _dp, _dp, _dp, C.c_long]
        L.orc_ecef2aer.argtypes = [_dp, _dp, _dp, _dp]
        L.orc_lla2ecef.argtypes = [_dp, _dp]
        L.orc_mean_z_uvw.argtypes = [_dp, C.c_int, _dp, C.c_int, _dp]
        L.orc_residual_z_aer.argtypes = [_dp, _dp, _dp, C.c_long]
        L.orc_ukf_predict.argtypes = [_dp, _dp, _dp, C.c_double, C.c_double, _dp, _dp, C.c_int, _dp]
        L.orc_ukf_update.argtypes = [_dp, _dp, _dp, _dp, _dp, _dp, _dp, C.c_double, C.c_int, C.c_int,
                                     C.c_int, _dp, _dp, _dp, _dp, _dp, _dp]
        L.orc_observe.argtypes = [_dp, _dp, _dp, _dp, _dp, C.c_long]
        L.orc_env_step.argtypes = [_dp, _dp, _dp, _dp, _dp, _dp, _ip, C.c_long, C.c_double, _dp, _dp,
                                   C.c_double, _dp, _dp, C.c_int, C.c_int, C.c_int, C.c_long, C.c_int,
                                   _dp, _dp, _dp, C.c_double, _dp, _dp, _dp, _dp, _dp]
        L.orc_aer_obs.argtypes = [_dp, _dp, _dp, _dp, _dp, _dp, C.c_long]
        L.orc_omp_threads.argtypes = [C.c_int]
        L.orc_omp_threads.restype = C.c_int

    # ---- P1-P5
    def propagate(self, x, dt):
        x = _a(x).reshape(-1, 6)
        out = np.empty_like(x)
        self.lib.orc_propagate(_p(x), _p(out), x.shape[0], float(dt))
        return out

    def kepler_intermediates(self, x, dt):
        x = _a(x).reshape(-1, 6)
        out = np.empty((x.shape[0], 8))
        self.lib.orc_kepler_intermediates(_p(x), _p(out), x.shape[0], float(dt))
        return out

    # ---- U1 / U2
    def robust_cholesky(self, A):
        """returns (U upper, rung) ; rung -1 = no jitter, 0..15 = 10^(rung-6), raises on 16."""
        A = _a(A).reshape(6, 6)
        U = np.zeros((6, 6))
        rung = C.c_int(0)
        rc = self.lib.orc_robust_cholesky(_p(A), _p(U), C.byref(rung))
        if rc:
            raise np.linalg.LinAlgError("robust_cholesky: jitter ladder exhausted")
        return U, rung.value

    def sigma_points(self, x, P, scale):
        x, P = _a(x), _a(P)
        sig = np.zeros((13, 6))
        rc = self.lib.orc_sigma_points(_p(x), _p(P), float(scale), _p(sig))
        if rc:
            raise np.linalg.LinAlgError("sigma_points")
        return sig

    # ---- H1 / H3 / H4 / T2
    def hx_aer(self, x, M, obs_lla, obs_itrs):
        x = _a(x).reshape(-1, 6)
        z = np.empty((x.shape[0], 3))
        M, obs_lla, obs_itrs = _a(M), _a(obs_lla), _a(obs_itrs)
        self.lib.orc_hx_aer(_p(x), _p(M), _p(obs_lla), _p(obs_itrs), _p(z), x.shape[0])
        return z

    def ecef2aer(self, obs_lla, sat, obs):
        out = np.empty(3)
        a, b, c = _a(obs_lla), _a(sat), _a(obs)
        self.lib.orc_ecef2aer(_p(a), _p(b), _p(c), _p(out))
        return out

    def lla2ecef(self, lla):
        out = np.empty(3)
        a = _a(lla)
        self.lib.orc_lla2ecef(_p(a), _p(out))
        return out

    def mean_z_uvw(self, sigmas, Wm, centred=False):
        s, w = _a(sigmas).reshape(-1, 3), _a(Wm)
        out = np.empty(3)
        self.lib.orc_mean_z_uvw(_p(s), s.shape[0], _p(w), int(centred), _p(out))
        return out

    def residual_z_aer(self, a, b):
        a, b = _a(a).reshape(-1, 3), _a(b).reshape(-1, 3)
        out = np.empty_like(a)
        self.lib.orc_residual_z_aer(_p(a), _p(b), _p(out), a.shape[0])
        return out

    # ---- U3 / U5 (one filter)
    def ukf_predict(self, x, P, Q, dt, Wm, Wc, scale, centred=False):
        x, P = _a(x).copy(), _a(P).copy()
        Q, Wm, Wc = _a(Q), _a(Wm), _a(Wc)
        sf = np.zeros((13, 6))
        rc = self.lib.orc_ukf_predict(_p(x), _p(P), _p(Q), float(dt), float(scale), _p(Wm), _p(Wc),
                                      int(centred), _p(sf))
        return rc, x, P, sf

    def ukf_update(self, x, P, sigmas_f, z, R, Wm, Wc, scale, M, obs_lla, obs_itrs, obs_type=0,
                   centred=False, resample=False):
        x, P, sf = _a(x).copy(), _a(P).copy(), _a(sigmas_f).copy()
        z, R, Wm, Wc, M = _a(z), _a(R), _a(Wm), _a(Wc), _a(M)
        obs_lla, obs_itrs = _a(obs_lla), _a(obs_itrs)
        y, S, sh = np.zeros(3), np.zeros((3, 3)), np.zeros((13, 3))
        rc = self.lib.orc_ukf_update(_p(x), _p(P), _p(sf), _p(z), _p(R), _p(Wm), _p(Wc), float(scale),
                                     int(obs_type), int(centred), int(resample), _p(M), _p(obs_lla),
                                     _p(obs_itrs), _p(y), _p(S), _p(sh))
        return rc, x, P, y, S, sh

    # ---- O1 / O2 / O4
    def observe(self, x_true, x, P):
        x_true, x, P = _a(x_true), _a(x), _a(P)
        m = x.shape[0]
        obs, met = np.empty((m, 12)), np.empty((4, m))
        self.lib.orc_observe(_p(x_true), _p(x), _p(P), _p(obs), _p(met), m)
        return obs, met

    def aer_obs(self, x, P, M, obs_lla, obs_itrs):
        x, P, M, obs_lla, obs_itrs = _a(x), _a(P), _a(M), _a(obs_lla), _a(obs_itrs)
        m = x.shape[0]
        out = np.empty(4 * m)
        self.lib.orc_aer_obs(_p(x), _p(P), _p(M), _p(obs_lla), _p(obs_itrs), _p(out), m)
        return out

    # ---- E1: whole env step
    def env_step(self, x_true, x, P, status, dt, Q, R, Wm, Wc, scale, action, M, obs_lla, obs_itrs,
                 obs_limit, z_noise3, obs_type=0, centred=False, resample=False, do_update=True):
        """one reference step for m objects.  `status` (int32[m]) is updated in place.
        returns dict(x_true, x, P, obs, metrics(4,m), obs_taken, z_true, y, S, sigmas_h)."""
        x_true, x, P = _a(x_true), _a(x), _a(P)
        m = x.shape[0]
        assert status.dtype == np.int32 and status.flags.c_contiguous
        Q, R, Wm, Wc, M = _a(Q), _a(R), _a(Wm), _a(Wc), _a(M)
        obs_lla, obs_itrs, zn = _a(obs_lla), _a(obs_itrs), _a(z_noise3)
        xt_o, x_o, P_o = np.empty_like(x_true), np.empty_like(x), np.empty_like(P)
        obs, met = np.empty((m, 12)), np.empty((4, m))
        upd, sh = np.empty(16), np.full((13, 3), np.nan)
        self.lib.orc_env_step(_p(x_true), _p(xt_o), _p(x), _p(P), _p(x_o), _p(P_o),
                              status.ctypes.data_as(_ip), m, float(dt), _p(Q), _p(R), float(scale),
                              _p(Wm), _p(Wc), int(centred), int(resample), int(obs_type), int(action),
                              int(do_update), _p(M), _p(obs_lla), _p(obs_itrs), float(obs_limit), _p(zn),
                              _p(obs), _p(met), _p(upd), _p(sh))
        return dict(x_true=xt_o, x=x_o, P=P_o, obs=obs, metrics=met, obs_taken=bool(upd[0]),
                    z_true=upd[1:4].copy(), y=upd[4:7].copy(), S=upd[7:16].reshape(3, 3).copy(),
                    sigmas_h=sh)


def reward_done(reward_type, delta_pos, sigma_pos_prev, action, i, n, rewards_so_far):
    """O3: reward / done logic of ssa_tasker_simple_2.py:324-354 and
    results.py:432 reward_proportional_trinary_true, restated on the metric arrays."""
    done = False
    reward = 0.0
    mx = np.max(delta_pos)
    if reward_type == 'jones':
        if mx > 5e6:
            done, reward = True, 0.0
        elif mx < 3e4:
            done, reward = True, 1.0
        elif i + 1 >= n:
            done, reward = True, 0.0
    elif reward_type == 'trinary':
        reward = float(np.mean(((delta_pos < 1e4) * 1 + (delta_pos < 1e7) * 1)) / 2)
    elif reward_type == 'shaped':
        if mx > 5e6:
            done, reward = True, 0.0
        elif mx < 3e4:
            done, reward = True, 1.0 - float(np.sum(rewards_so_far))
        elif action == int(np.argmax(sigma_pos_prev)):
            reward = 1.0 / n
        else:
            reward = -1.0 / n
    if i + 1 >= n:
        done = True
    return reward, done
