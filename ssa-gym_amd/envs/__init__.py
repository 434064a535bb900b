"""Mirror of the reference's `envs` package (envs/__init__.py:1-28): `env_config` (the dict
that IS the reference's config system and operator plug API), `sample_orbits`, and the gym
registration of 'ssa_tasker_simple-v2'."""
import os
from datetime import datetime

import numpy as np

from ..catalogue import synthetic_catalogue
from ..host import arcsec2rad
from . import _gymshim
from .dynamics import (fx_xyz_farnocchia as fx, hx_aer_erfa as hx, mean_z_uvw as mean_z,  # noqa: F401
                       residual_z_aer as residual_z, robust_cholesky)

sample_orbits_file = '1.5_hour_viz_20000_of_20000_sample_orbits_seed_0.npy'


def _find_catalogue():
    """the reference walks os.getcwd() for its catalogue file (envs/__init__.py:12-16); honour
    that when the file is present (running inside a reference checkout), else $SSA_GYM_ORBITS,
    else a synthetic catalogue of the same shape and regime mix (catalogue.py)."""
    p = os.environ.get("SSA_GYM_ORBITS")
    if p and os.path.exists(p):
        return np.load(p)
    for cand in (os.path.join(os.getcwd(), "envs", sample_orbits_file), os.path.join(os.getcwd(), sample_orbits_file)):
        if os.path.exists(cand):
            return np.load(cand)
    return synthetic_catalogue(20000, seed=0)


sample_orbits = _find_catalogue()

# The reference's default configuration (envs/__init__.py:23-28), key for key: this dict IS the reference's
# config system -- scripts mutate it in place and hand it to the env -- so the names and defaults are the API.
_X_SIGMA = (1e5,) * 3 + (1e2,) * 3                      # initial estimate noise [m, m/s]
env_config = dict(
    # simulation
    steps=480, rso_count=10, time_step=20., t_0=datetime(2020, 5, 4, 0, 0, 0),
    observer=(38.828198, -77.305352, 20.0), obs_limit=-90, update_interval=1,
    orbits=sample_orbits, obs_returned='flatten', reward_type='jones',
    # measurement model: az / el in arc seconds, range in metres
    obs_type='aer', z_sigma=(1, 1, 1e3), R=np.diag([arcsec2rad ** 2, arcsec2rad ** 2, 1e3 ** 2]),
    # filter
    x_sigma=_X_SIGMA, P_0=np.diag(np.square(_X_SIGMA)), q_sigma=0.000025,
    alpha=0.0001, beta=2., kappa=3 - 6,
    # operator plug points (tokens selecting fused kernel variants, see dynamics.py)
    fx=fx, hx=hx, mean_z=mean_z, residual_z=residual_z, msqrt=robust_cholesky,
)

if _gymshim.register is not None:  # pragma: no cover - only with gym / gymnasium installed
    try:
        _gymshim.register(id='ssa_tasker_simple-v2', entry_point='ssa_gym_amd.envs.ssa_tasker_simple_2:SSA_Tasker_Env')
    except Exception:  # noqa: BLE001  (already registered)
        pass


def make(id='ssa_tasker_simple-v2', config=None):
    """gym.make('ssa_tasker_simple-v2', config=cfg) equivalent that also works without gym."""
    if id not in ('ssa_tasker_simple-v2', 'ssa_tasker_simple_2-v0'):
        raise ValueError("unknown env id %r" % id)
    from .ssa_tasker_simple_2 import SSA_Tasker_Env
    return SSA_Tasker_Env(env_config if config is None else config)
