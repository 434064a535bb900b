"""Host-side (init-time) logic of the hot path: everything the reference computes ONCE
per environment in SSA_Tasker_Env.__init__ -- filter weights, process noise, observer
geometry -- packed into the `ssa_consts` block the kernels take by value.

None of this is per-step work; it runs in numpy.  Citations: file:line under the
reference root.
"""
from fractions import Fraction

import numpy as np

from . import _lib

arcsec2rad = np.pi / 648000   # envs/transformations.py:14
deg2rad = np.pi / 180         # envs/transformations.py:15
# WGS84 (envs/transformations.py:11-13: a, f = erfa.eform(1))
WGS84_A = 6378137.0
WGS84_F = 0.0033528106647474805

X_FAILED = np.array([1e20, 1e20, 1e20, 1e12, 1e12, 1e12])   # ssa_tasker_simple_2.py:157
P_FAILED = np.diag([1e20, 1e20, 1e20, 1e12, 1e12, 1e12])    # ssa_tasker_simple_2.py:158


def merwe_weights(alpha, beta, kappa, n=6):
    """filterpy MerweScaledSigmaPoints._compute_weights as used at
    ssa_tasker_simple_2.py:211-214.  Returns Wm[2n+1], Wc[2n+1], (n + lambda)."""
    lambda_ = alpha ** 2 * (n + kappa) - n
    c = .5 / (n + lambda_)
    Wc = np.full(2 * n + 1, c)
    Wm = np.full(2 * n + 1, c)
    Wc[0] = lambda_ / (n + lambda_) + (1 - alpha ** 2 + beta)
    Wm[0] = lambda_ / (n + lambda_)
    return Wm, Wc, lambda_ + n


def exact_weight_sums(Wm, Wc):
    """sum(Wm) - 1 and sum(Wc) of the DOUBLE weights, evaluated exactly.  At alpha=1e-4 the
    doubles Wm0 ~ -2e8 and 12*Wi ~ 2e8+1 do not sum to exactly 1 (ulp(2e8) = 3e-8): the
    reference's np.dot(Wm, sigmas) therefore carries the factor sum(Wm) on sigma_0, and the
    centred form used on the device reproduces it."""
    sm = sum(Fraction(float(w)) for w in Wm) - 1
    sc = sum(Fraction(float(w)) for w in Wc)
    return float(sm), float(sc)


def Q_discrete_white_noise(dim, dt=1., var=1., block_size=1, order_by_dim=True):
    """filterpy.common.Q_discrete_white_noise, dim == 2 (call: ssa_tasker_simple_2.py:110)."""
    if dim != 2:
        raise ValueError("the reference only uses dim=2")
    q = np.array([[.25 * dt ** 4, .5 * dt ** 3], [.5 * dt ** 3, dt ** 2]])
    if order_by_dim:
        return np.kron(np.eye(block_size), q) * var
    return np.kron(q, np.eye(block_size)) * var


def lla2ecef(obs_lla):
    """envs/transformations.py:217-235 (observer position, init-time)."""
    a, f = WGS84_A, WGS84_F
    e = np.sqrt(f * (2 - f))
    lat, lon, alt = obs_lla
    N = a / np.sqrt(1 - e ** 2 * np.sin(lat) ** 2)
    x = (N + alt) * np.cos(lat) * np.cos(lon)
    y = (N + alt) * np.cos(lat) * np.sin(lon)
    z = (N * (1 - e ** 2) + alt) * np.sin(lat)
    return np.array([x, y, z])


def enu_matrix(obs_lla):
    """trans_uvw_ecef of ecef2aer (envs/transformations.py:341-343); constant per observer."""
    lat, lon = obs_lla[0], obs_lla[1]
    return np.array([[-np.sin(lat) * np.cos(lon), -np.sin(lon), np.cos(lat) * np.cos(lon)],
                     [-np.sin(lat) * np.sin(lon), np.cos(lon), np.cos(lat) * np.sin(lon)],
                     [np.cos(lat), 0, np.sin(lat)]])


J2_EARTH = 0.00108263      # poliastro Earth.J2 (what the reference's ecosystem would plug into ad=J2_perturbation)
R_EQ_EARTH = 6378136.6     # poliastro Earth.R [m]


def make_consts(Q, R, alpha, beta, kappa, dt, obs_limit_rad, obs_lla, obs_type='aer', propagator='fg',
                resample=False, update_interval=1, j2=J2_EARTH, r_eq=R_EQ_EARTH, rk4_substeps=None, covariance=None):
    """pack the per-environment constants for the kernels (include/ssa_hip.h: ssa_consts).

    covariance: 'reference' = the prior covariance in the reference's own arithmetic (SSA_FLAG_REFERENCE_COV: reproduces the
    reference's episode-level filter failures), 'centred' = the cancellation-free expansion; None = 'reference' with the
    'elements' and 'hybrid' propagators (the behaviour-faithful variants), 'centred' otherwise."""
    Wm, Wc, scale = merwe_weights(alpha, beta, kappa)
    sm, sc = exact_weight_sums(Wm, Wc)
    c = _lib.ssa_consts()
    c.Q[:] = np.asarray(Q, dtype=np.float64).reshape(36)
    c.R[:] = np.asarray(R, dtype=np.float64).reshape(9)
    c.Wm0, c.Wc0, c.Wi = float(Wm[0]), float(Wc[0]), float(Wm[1])
    c.sum_wm_m1, c.sum_wc, c.scale = sm, sc, float(scale)
    c.dt, c.obs_limit = float(dt), float(obs_limit_rad)
    obs_lla = np.asarray(obs_lla, dtype=np.float64)
    c.enu[:] = enu_matrix(obs_lla).reshape(9)
    c.obs_itrs[:] = lla2ecef(obs_lla)
    c.obs_type = {'aer': _lib.OBS_AER, 'xyz': _lib.OBS_XYZ}[obs_type]
    c.propagator = {'elements': _lib.PROP_ELEMENTS, 'fg': _lib.PROP_FG, 'j2': _lib.PROP_J2_RK4, 'hybrid': _lib.PROP_HYBRID}[propagator]
    c.j2, c.r_eq = float(j2), float(r_eq)
    # RK4 sub-step <= 5 s: local error (n h)^5/120 |r| < 1e-6 m even in LEO
    c.rk4_substeps = int(rk4_substeps) if rk4_substeps else max(1, int(np.ceil(abs(dt) / 5.0)))
    if covariance is None:
        covariance = 'reference' if propagator in ('elements', 'hybrid') else 'centred'
    if covariance not in ('reference', 'centred'):
        raise ValueError("covariance must be 'reference' or 'centred', got %r" % (covariance,))
    c.flags = (_lib.FLAG_RESAMPLE if resample else 0) | (_lib.FLAG_REFERENCE_COV if covariance == 'reference' else 0)
    c.update_interval = int(update_interval)
    return c
