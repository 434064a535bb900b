"""ssa-gym hot path on AMD MI355X (gfx950).

propagate -> UKF predict -> one UKF update -> observation / metrics / reward for every
tracked space object, as hand-written fp64 HIP kernels behind the reference's gym.Env API.

    _lib      ctypes binding of libssa_hip.so (C ABI: include/ssa_hip.h); no CPU fallback
    host      init-time constants (Merwe weights, Q, observer geometry)
    device    one function per hot-path operator on CUDA tensors
    engine    device-resident state + per-step launch sequence
    envs      drop-in mirror of the reference's `envs` package (SSA_Tasker_Env, env_config)
"""
__version__ = "0.1.0"

from . import _build, _lib, host  # noqa: F401


def build(force=False, verbose=False):
    """compile the HIP library in-tree (hipcc, gfx950)."""
    return _build.build_library(force=force, verbose=verbose)
