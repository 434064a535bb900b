"""ctypes binding of libssa_hip.so (include/ssa_hip.h).

There is NO CPU fallback: if the library is missing or a call fails this module
raises.  Use `_build.build_library()` (or `python __graft_entry__.py`) to build it.
"""
import ctypes as C
import os

from . import _build

c_dp = C.c_void_p  # device pointers travel as integers


class ssa_consts(C.Structure):
    _fields_ = [
        ("Q", C.c_double * 36), ("R", C.c_double * 9),
        ("Wm0", C.c_double), ("Wc0", C.c_double), ("Wi", C.c_double),
        ("sum_wm_m1", C.c_double), ("sum_wc", C.c_double), ("scale", C.c_double),
        ("dt", C.c_double), ("obs_limit", C.c_double),
        ("enu", C.c_double * 9), ("obs_itrs", C.c_double * 3),
        ("obs_type", C.c_int32), ("propagator", C.c_int32), ("flags", C.c_uint32),
        ("update_interval", C.c_int32),
        ("j2", C.c_double), ("r_eq", C.c_double), ("rk4_substeps", C.c_int32), ("reserved0", C.c_int32),
    ]


class ssa_step_params(C.Structure):
    _fields_ = [
        ("n_obj", C.c_int64), ("n_env", C.c_int32), ("time_offset", C.c_int32),
        ("x_true_in", c_dp), ("x_true_out", c_dp), ("x_in", c_dp), ("x_out", c_dp),
        ("P_in", c_dp), ("P_out", c_dp), ("status", c_dp), ("obs", c_dp), ("metrics", c_dp),
        ("upd", c_dp), ("trans", c_dp), ("env_time", c_dp), ("actions", c_dp), ("z_noise", c_dp),
        ("zn_stride_env", C.c_int64), ("zn_stride_time", C.c_int64), ("zn_stride_obj", C.c_int64),
        ("n_time", C.c_int32), ("launch_mask", C.c_uint32), ("stats", c_dp), ("work", c_dp), ("stat_ws", c_dp),
        ("stat_shards", c_dp), ("stat_shards_prev", c_dp), ("stats_prev", c_dp), ("aer_out", c_dp),
        ("stat_shards_clear", c_dp), ("aer_cols", C.c_int32), ("action0", C.c_int32), ("obs_mirror", c_dp),
        ("inline_time", C.c_int32 * 8), ("inline_action", C.c_int32 * 8),
        ("spos_tiles", c_dp), ("spos_tiles_prev", c_dp),
        ("fail_log", c_dp), ("fail_count", c_dp), ("fail_cap", C.c_int32), ("reserved1", C.c_int32),
        ("obj_ids", c_dp),
    ]


class ssa_rollout_params(C.Structure):
    _fields_ = [
        ("n_steps", C.c_int32), ("history", C.c_int32), ("slot_out", C.c_int32), ("reserved", C.c_int32),
        ("x_true_ring", c_dp), ("x_ring", c_dp), ("P_ring", c_dp), ("obs_ring", c_dp), ("metrics_ring", c_dp),
        ("upd_ring", c_dp), ("stats_ring", c_dp), ("actions", c_dp), ("stat_shards", c_dp), ("spos_tiles", c_dp),
    ]


class ssa_closed_loop_params(C.Structure):
    _fields_ = [
        ("n_steps", C.c_int32), ("history", C.c_int32), ("slot_out", C.c_int32), ("agent", C.c_int32),
        ("x_true_ring", c_dp), ("x_ring", c_dp), ("P_ring", c_dp), ("obs_ring", c_dp), ("metrics_ring", c_dp),
        ("upd_out", c_dp), ("stats_out", c_dp), ("actions", c_dp), ("fallback", c_dp), ("picks", c_dp), ("error", c_dp),
        ("workspace", c_dp), ("workspace_bytes", C.c_int64),
        ("wait_ticks", C.c_int64), ("flags", C.c_uint32), ("reserved", C.c_uint32), ("slot_of", c_dp),
    ]


# constants of include/ssa_hip.h
E_INVALID, E_LAUNCH, E_UNSUPPORTED = -1, -2, -3
ABI_VERSION = 22
ST_OK, ST_PREDICT_NAN, ST_PREDICT_LINALG, ST_UPDATE_NAN, ST_UPDATE_LINALG = range(5)
OBS_AER, OBS_XYZ = 0, 1
PROP_ELEMENTS, PROP_FG, PROP_J2_RK4, PROP_HYBRID = 0, 1, 2, 3
FLAG_RESAMPLE = 1
FLAG_REFERENCE_COV = 2
UPD_STRIDE, UPD_OBS_TAKEN, UPD_Z_TRUE, UPD_Y, UPD_S, UPD_SIGMAS_H, UPD_VISIBLE, UPD_ACTION = 64, 0, 1, 4, 7, 16, 55, 56
STAT_SHARDS = 128
STAT_SHARD_WORDS = 16
PROFILE_SLOTS = 1024
LAUNCH_DEFER_FOLD = 8
LAUNCH_INLINE_ACTION = 16
LAUNCH_FOLD_INSIDE = 32
LAUNCH_INLINE_ENVS = 64
LAUNCH_MIRROR_F32 = 128
INLINE_ENVS = 8
LOOP_ARGMAX_SPOS, LOOP_DEBUG_WITHHOLD = 1, 2
FAIL_STRIDE, FAIL_ENV, FAIL_OBJ, FAIL_STATUS, FAIL_TIME, FAIL_ERR = 8, 0, 1, 2, 3, 4
AGENT_NAIVE_GREEDY, AGENT_VISIBLE_GREEDY, AGENT_SHANNON, AGENT_POS_ERROR, AGENT_VEL_ERROR = range(5)
STAT_STRIDE, STAT_MAX_DPOS, STAT_CNT_LT_1E4, STAT_CNT_LT_1E7, STAT_ARGMAX_SPOS, STAT_N_FAILED, STAT_MAX_SPOS = 8, 0, 1, 2, 3, 4, 5

# every symbol the header declares, with its ctypes signature
SIGNATURES = {
    "ssa_abi_version": (C.c_int, []),
    "ssa_build_info": (C.c_char_p, []),
    "ssa_env_step_f64": (C.c_int, [C.POINTER(ssa_consts), C.POINTER(ssa_step_params), c_dp]),
    "ssa_env_step_profiled_f64": (C.c_int, [C.POINTER(ssa_consts), C.POINTER(ssa_step_params), c_dp, C.c_int32]),
    "ssa_env_step_profile_ms": (C.c_int, [C.c_int32, C.POINTER(C.c_float)]),
    "ssa_stats_fold_f64": (C.c_int, [c_dp, c_dp, C.c_int32, c_dp]),
    "ssa_stats_fold_spos_f64": (C.c_int, [c_dp, c_dp, c_dp, C.c_int64, C.c_int32, c_dp]),
    "ssa_ladder_probe_f64": (C.c_int, [c_dp, C.c_double, c_dp, c_dp, c_dp, C.c_int64, c_dp]),
    "ssa_env_rollout_f64": (C.c_int, [C.POINTER(ssa_consts), C.POINTER(ssa_step_params), C.POINTER(ssa_rollout_params), c_dp]),
    "ssa_env_closed_loop_f64": (C.c_int, [C.POINTER(ssa_consts), C.POINTER(ssa_step_params), C.POINTER(ssa_closed_loop_params), c_dp]),
    "ssa_closed_loop_workspace_bytes": (C.c_int64, [C.c_int64, C.c_int32]),
    "ssa_env_step_work_bytes": (C.c_int64, [C.c_int64, C.c_int32]),
    "ssa_reward_stats_f64": (C.c_int, [c_dp, c_dp, c_dp, c_dp, C.c_int64, C.c_int32, c_dp]),
    "ssa_reward_stats_workspace_bytes": (C.c_int64, [C.c_int32]),
    "ssa_propagate_f64": (C.c_int, [c_dp, c_dp, C.c_int64, C.c_double, C.c_int32, c_dp]),
    "ssa_propagate_j2_f64": (C.c_int, [c_dp, c_dp, C.c_int64, C.c_double, C.c_double, C.c_double, C.c_int32, c_dp]),
    "ssa_kepler_elements_f64": (C.c_int, [c_dp, c_dp, C.c_int64, C.c_double, c_dp]),
    "ssa_robust_cholesky6_f64": (C.c_int, [c_dp, c_dp, c_dp, C.c_int64, c_dp]),
    "ssa_sigma_points_f64": (C.c_int, [c_dp, c_dp, C.c_double, c_dp, c_dp, C.c_int64, c_dp]),
    "ssa_hx_aer_f64": (C.c_int, [c_dp, C.c_int64, c_dp, C.POINTER(ssa_consts), c_dp, C.c_int64, c_dp]),
    "ssa_mean_z_uvw_f64": (C.c_int, [c_dp, C.POINTER(ssa_consts), c_dp, C.c_int64, c_dp]),
    "ssa_residual_z_aer_f64": (C.c_int, [c_dp, c_dp, c_dp, C.c_int64, c_dp]),
    "ssa_visible_mask_f64": (C.c_int, [c_dp, c_dp, C.POINTER(ssa_consts), c_dp, c_dp, C.c_int64, c_dp]),
    "ssa_observe_f64": (C.c_int, [c_dp, c_dp, c_dp, c_dp, c_dp, C.c_int64, c_dp]),
    "ssa_agent_scores_f64": (C.c_int, [c_dp, c_dp, c_dp, c_dp, c_dp, C.POINTER(ssa_consts), c_dp, c_dp, C.c_int64, c_dp]),
    "ssa_visible_mask_at_f64": (C.c_int, [c_dp, c_dp, c_dp, C.c_int32, C.c_int32, C.POINTER(ssa_consts), c_dp, c_dp, C.c_int64, c_dp]),
    "ssa_agent_scores_at_f64": (C.c_int, [c_dp, c_dp, c_dp, c_dp, c_dp, c_dp, C.c_int32, C.c_int32, C.POINTER(ssa_consts), c_dp, c_dp, C.c_int64, c_dp]),
    "ssa_masked_argmax_f64": (C.c_int, [c_dp, c_dp, C.c_int64, c_dp, c_dp]),
    "ssa_masked_argmax_ws_f64": (C.c_int, [c_dp, c_dp, C.c_int64, c_dp, c_dp, C.c_int64, c_dp]),
    "ssa_masked_argmax_workspace_bytes": (C.c_int64, [C.c_int64]),
    "ssa_peer_push_f64": (C.c_int, [c_dp, C.c_int64, c_dp, c_dp, C.c_int32, c_dp, C.c_uint64, c_dp, C.c_int64, c_dp, c_dp, c_dp]),
    "ssa_peer_wait": (C.c_int, [c_dp, C.c_int32, c_dp, C.c_uint64, C.c_int64, c_dp, c_dp]),
    "ssa_aer_obs_f64": (C.c_int, [c_dp, c_dp, c_dp, C.POINTER(ssa_consts), c_dp, C.c_int64, c_dp]),
    "ssa_agent_select_f64": (C.c_int, [C.POINTER(ssa_consts), C.c_int32, c_dp, c_dp, c_dp, c_dp, c_dp, c_dp, C.c_int32, C.c_int32,
                                       c_dp, c_dp, c_dp, c_dp, C.c_int64, C.c_int32, c_dp]),
    "ssa_agent_select_ids_f64": (C.c_int, [C.POINTER(ssa_consts), C.c_int32, c_dp, c_dp, c_dp, c_dp, c_dp, c_dp, C.c_int32, C.c_int32,
                                           c_dp, c_dp, c_dp, c_dp, C.c_int64, C.c_int32, c_dp, c_dp]),
    "ssa_agent_select_workspace_bytes": (C.c_int64, [C.c_int64, C.c_int32]),
    "ssa_nees_f64": (C.c_int, [c_dp, c_dp, c_dp, c_dp, C.c_int64, c_dp]),
    "ssa_nis_f64": (C.c_int, [c_dp, c_dp, c_dp, C.c_int64, c_dp]),
    "ssa_chi2_contained_f64": (C.c_int, [c_dp, C.c_int64, C.c_double, C.c_double, c_dp, c_dp]),
}

_lib = None


class SsaHipError(RuntimeError):
    pass


def library_path():
    return _build.LIB


def load():
    """dlopen libssa_hip.so and bind every declared symbol; raises if absent."""
    global _lib
    if _lib is not None:
        return _lib
    path = library_path()
    if not os.path.exists(path):
        raise SsaHipError(
            "libssa_hip.so is not built (%s). The ssa-gym hot path has no CPU fallback: run "
            "`python -c 'import __graft_entry__ as g; g.build()'` (needs hipcc)." % path)
    # One HIP/HSA runtime per process: torch (which owns the device memory and streams handed to this library) ships its own
    # libamdhip64 / libhsa-runtime64.  Loaded after torch, libssa_hip.so binds to those; loaded first it would pull in
    # /opt/rocm's copies and the second runtime to open the device finds none (hipErrorNoDevice on the first launch).
    import torch  # noqa: F401
    lib = C.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    if lib.ssa_abi_version() != ABI_VERSION:
        raise SsaHipError("libssa_hip.so ABI %d != binding %d: rebuild" % (lib.ssa_abi_version(), ABI_VERSION))
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        raise SsaHipError("%s failed with code %d" % (what, rc))
