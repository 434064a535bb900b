"""Mirror of the hot functions of the reference's envs/results.py (observations, error,
dist3d, var3d, reward_proportional_trinary_true, error_failed).  Array-level calls run the
device operators; the plotting half of the reference module is out of scope."""
import numpy as np

from .. import _lib


def _dev():
    import torch
    from .. import device
    if not torch.cuda.is_available():
        raise _lib.SsaHipError("results.* need a GPU: the hot path has no CPU fallback")
    return device


def observations(filters_x, filters_P):
    """results.py:61 -- (m,6),(m,6,6) -> (m,12) [x | diag P]."""
    device = _dev()
    x = device.as_dev(np.asarray(filters_x, dtype=np.float64))
    P = device.as_dev(np.asarray(filters_P, dtype=np.float64))
    obs, _ = device.observe(x, x, P)
    return obs.cpu().numpy()


def error(states, obs):
    """results.py:37 -- delta_pos, delta_vel, sigma_pos, sigma_vel from true states and obs rows."""
    device = _dev()
    obs = np.asarray(obs, dtype=np.float64)
    m = obs.shape[0]
    P = np.zeros((m, 6, 6))
    P[:, np.arange(6), np.arange(6)] = obs[:, 6:]
    _, met = device.observe(device.as_dev(np.asarray(states, dtype=np.float64)), device.as_dev(obs[:, :6].copy()),
                            device.as_dev(P))
    met = met.cpu().numpy()
    return met[0], met[1], met[2], met[3]


def reward_proportional_trinary_true(delta_pos):
    """results.py:432 from the O3 statistics kernel."""
    device = _dev()
    import torch
    d = np.asarray(delta_pos, dtype=np.float64)
    met = np.zeros((1, 4, d.size))
    met[0, 0] = d
    st = device.reward_stats(device.as_dev(met), torch.zeros(d.size, dtype=torch.int32, device="cuda"), d.size, 1)
    st = st.cpu().numpy()[0]
    return (st[_lib.STAT_CNT_LT_1E4] + st[_lib.STAT_CNT_LT_1E7]) / d.size / 2


def error_failed(state, x, P):
    """results.py:51 (formats the failure message; three subtractions, host side)."""
    r = np.zeros(4)
    r[0] = np.sqrt(np.sum((x[:3] - state[:3]) ** 2))
    r[1] = np.sqrt(np.sum((x[3:] - state[3:]) ** 2))
    r[2] = np.sqrt(np.sum(np.diag(P)[:3])) if np.ndim(P) == 2 else np.sqrt(np.sum(P[:3]))
    r[3] = np.sqrt(np.sum(np.diag(P)[3:])) if np.ndim(P) == 2 else np.sqrt(np.sum(P[3:]))
    return r
