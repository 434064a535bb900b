"""Thin torch front-end over the C ABI: PyTorch only owns device memory and streams here.

Every function takes/returns CUDA (ROCm) float64 tensors laid out exactly like the
reference's numpy arrays and launches on torch's current stream.  There is no CPU path:
a CPU tensor or a missing library raises.
"""
import ctypes as C

import torch

from . import _lib

f64 = torch.float64


def _chk(t, name, dtype=f64):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise _lib.SsaHipError("%s must be a CUDA tensor (the hot path has no CPU fallback)" % name)
    if t.dtype != dtype or not t.is_contiguous():
        raise _lib.SsaHipError("%s must be contiguous %s" % (name, dtype))
    return t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def as_dev(a, device="cuda", dtype=f64):
    if not isinstance(a, torch.Tensor):
        import numpy as np
        a = np.asarray(a)
        if not a.flags.writeable or not a.flags.c_contiguous:
            a = np.array(a, order="C")   # torch refuses read-only / strided numpy views
    return torch.as_tensor(a, dtype=dtype).contiguous().to(device)


def propagate(x, dt, propagator=_lib.PROP_FG, out=None):
    """P1-P5: fx_xyz_farnocchia for every row of x[n,6]."""
    lib = _lib.load()
    n = x.shape[0]
    out = torch.empty_like(x) if out is None else out
    _lib.check(lib.ssa_propagate_f64(_chk(x, "x"), _chk(out, "out"), n, float(dt), int(propagator), _stream()),
               "ssa_propagate_f64")
    return out


def propagate_j2(x, dt, j2, r_eq, substeps, out=None):
    """extension: two-body + J2, RK4 with `substeps` steps per dt."""
    lib = _lib.load()
    n = x.shape[0]
    out = torch.empty_like(x) if out is None else out
    _lib.check(lib.ssa_propagate_j2_f64(_chk(x, "x"), _chk(out, "out"), n, float(dt), float(j2), float(r_eq),
                                        int(substeps), _stream()), "ssa_propagate_j2_f64")
    return out


def kepler_elements(x, dt):
    lib = _lib.load()
    n = x.shape[0]
    coe = torch.empty((n, 8), dtype=f64, device=x.device)
    _lib.check(lib.ssa_kepler_elements_f64(_chk(x, "x"), _chk(coe, "coe"), n, float(dt), _stream()),
               "ssa_kepler_elements_f64")
    return coe


def robust_cholesky(A):
    """U2: returns (U[n,6,6] upper, rung[n])."""
    lib = _lib.load()
    n = A.shape[0]
    U = torch.empty_like(A)
    rung = torch.empty(n, dtype=torch.int32, device=A.device)
    _lib.check(lib.ssa_robust_cholesky6_f64(_chk(A, "A"), _chk(U, "U"), _chk(rung, "rung", torch.int32), n, _stream()),
               "ssa_robust_cholesky6_f64")
    return U, rung


def ladder_probe(A, scale):
    """U2 as the fused step kernels run it (ssa_ladder_probe_f64): returns (rung[n], mask[n], U[n,6,6]) -- the rung of the kernels' one-pass
    ladder on scale * A, and which of the sixteen rungs factorise in that arithmetic (bit i; bit 16 = the plain attempt)."""
    lib = _lib.load()
    n = A.shape[0]
    U = torch.empty_like(A)
    rung = torch.empty(n, dtype=torch.int32, device=A.device)
    mask = torch.empty(n, dtype=torch.int32, device=A.device)
    _lib.check(lib.ssa_ladder_probe_f64(_chk(A, "A"), float(scale), _chk(rung, "rung", torch.int32), _chk(mask, "mask", torch.int32),
                                        _chk(U, "U"), n, _stream()), "ssa_ladder_probe_f64")
    return rung, mask, U


def sigma_points(x, P, scale):
    """U1: returns (sigmas[n,13,6], fail[n])."""
    lib = _lib.load()
    n = x.shape[0]
    sig = torch.empty((n, 13, 6), dtype=f64, device=x.device)
    fail = torch.empty(n, dtype=torch.int32, device=x.device)
    _lib.check(lib.ssa_sigma_points_f64(_chk(x, "x"), _chk(P, "P"), float(scale), _chk(sig, "sig"),
                                        _chk(fail, "fail", torch.int32), n, _stream()), "ssa_sigma_points_f64")
    return sig, fail


def hx_aer(x, M, consts):
    """H1: az/el/range of x[n,>=3] seen through GCRS->ITRS matrix M[3,3]."""
    lib = _lib.load()
    n = x.shape[0]
    z = torch.empty((n, 3), dtype=f64, device=x.device)
    _lib.check(lib.ssa_hx_aer_f64(_chk(x, "x"), x.shape[1], _chk(M, "M"), C.byref(consts), _chk(z, "z"), n, _stream()),
               "ssa_hx_aer_f64")
    return z


def mean_z_uvw(sigmas, consts):
    lib = _lib.load()
    n = sigmas.shape[0]
    zp = torch.empty((n, 3), dtype=f64, device=sigmas.device)
    _lib.check(lib.ssa_mean_z_uvw_f64(_chk(sigmas, "sigmas"), C.byref(consts), _chk(zp, "zp"), n, _stream()),
               "ssa_mean_z_uvw_f64")
    return zp


def residual_z_aer(a, b):
    lib = _lib.load()
    n = a.shape[0]
    c = torch.empty_like(a)
    _lib.check(lib.ssa_residual_z_aer_f64(_chk(a, "a"), _chk(b, "b"), _chk(c, "c"), n, _stream()),
               "ssa_residual_z_aer_f64")
    return c


def visible_mask(x_true, M, consts, want_el=False):
    """V1: object_visibility() for every row of x_true[n,6]."""
    lib = _lib.load()
    n = x_true.shape[0]
    mask = torch.empty(n, dtype=torch.uint8, device=x_true.device)
    el = torch.empty(n, dtype=f64, device=x_true.device) if want_el else None
    _lib.check(lib.ssa_visible_mask_f64(_chk(x_true, "x_true"), _chk(M, "M"), C.byref(consts),
                                        _chk(mask, "mask", torch.uint8), _chk(el, "el") if want_el else None, n,
                                        _stream()), "ssa_visible_mask_f64")
    return (mask, el) if want_el else mask


def observe(x_true, x, P, obs=None, metrics=None):
    """O1/O2: observations() + error() -> obs[n,12], metrics[4,n]."""
    lib = _lib.load()
    n = x.shape[0]
    obs = torch.empty((n, 12), dtype=f64, device=x.device) if obs is None else obs
    metrics = torch.empty((4, n), dtype=f64, device=x.device) if metrics is None else metrics
    _lib.check(lib.ssa_observe_f64(_chk(x_true, "x_true"), _chk(x, "x"), _chk(P, "P"), _chk(obs, "obs"),
                                   _chk(metrics, "metrics"), n, _stream()), "ssa_observe_f64")
    return obs, metrics


def aer_obs(x, P, M, consts, out=None):
    """O4: aer_obs() -> out[n,4]."""
    lib = _lib.load()
    n = x.shape[0]
    out = torch.empty((n, 4), dtype=f64, device=x.device) if out is None else out
    _lib.check(lib.ssa_aer_obs_f64(_chk(x, "x"), _chk(P, "P"), _chk(M, "M"), C.byref(consts), _chk(out, "out"), n,
                                   _stream()), "ssa_aer_obs_f64")
    return out


_stats_ws = {}


def stats_workspace(n_env, dev):
    """device scratch of the two-launch statistics reduction (cached per (device, n_env))."""
    key = (str(dev), int(n_env))
    if key not in _stats_ws:
        nbytes = _lib.load().ssa_reward_stats_workspace_bytes(int(n_env))
        _stats_ws[key] = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    return _stats_ws[key]


def reward_stats(metrics, status, n_obj, n_env=1, out=None):
    """O3: per-env reductions -> stats[E, STAT_STRIDE]."""
    lib = _lib.load()
    out = torch.empty((n_env, _lib.STAT_STRIDE), dtype=f64, device=metrics.device) if out is None else out
    ws = stats_workspace(n_env, metrics.device)
    _lib.check(lib.ssa_reward_stats_f64(_chk(metrics, "metrics"), _chk(status, "status", torch.int32),
                                        _chk(out, "stats"), ws.data_ptr(), int(n_obj), int(n_env), _stream()),
               "ssa_reward_stats_f64")
    return out


def agent_scores(x_true, x_cur, P_cur, P_prev, M, consts, want_mask=True):
    """agents.py primitives: scores[4, n] (trace P | log det ratio | delta_pos | delta_vel) and the visibility mask."""
    lib = _lib.load()
    n = x_cur.shape[0]
    scores = torch.empty((4, n), dtype=f64, device=x_cur.device)
    mask = torch.empty(n, dtype=torch.uint8, device=x_cur.device) if want_mask else None
    _lib.check(lib.ssa_agent_scores_f64(_chk(x_true, "x_true"), _chk(x_cur, "x_cur"), _chk(P_cur, "P_cur"),
                                        _chk(P_prev, "P_prev") if P_prev is not None else None,
                                        _chk(M, "M") if want_mask else None, C.byref(consts), _chk(scores, "scores"),
                                        _chk(mask, "mask", torch.uint8) if want_mask else None, n, _stream()),
               "ssa_agent_scores_f64")
    return scores, mask


def agent_scores_at(x_true, x_cur, P_cur, P_prev, trans, env_time, time_offset, consts):
    """agent_scores() with the GCRS -> ITRS matrix picked ON THE DEVICE: row (env_time[0] + time_offset) % n_time of the table `trans`
    (callers inside a captured graph, whose time index the graph advances between replays)."""
    lib = _lib.load()
    n = x_cur.shape[0]
    scores = torch.empty((4, n), dtype=f64, device=x_cur.device)
    mask = torch.empty(n, dtype=torch.uint8, device=x_cur.device)
    _lib.check(lib.ssa_agent_scores_at_f64(_chk(x_true, "x_true"), _chk(x_cur, "x_cur"), _chk(P_cur, "P_cur"),
                                           _chk(P_prev, "P_prev") if P_prev is not None else None, _chk(trans, "trans"),
                                           _chk(env_time, "env_time", torch.int32), int(time_offset), int(trans.shape[0]), C.byref(consts),
                                           _chk(scores, "scores"), _chk(mask, "mask", torch.uint8), n, _stream()), "ssa_agent_scores_at_f64")
    return scores, mask


def visible_mask_at(x_true, trans, env_time, time_offset, consts):
    """visible_mask() with the matrix picked on the device (see agent_scores_at)"""
    lib = _lib.load()
    n = x_true.shape[0]
    mask = torch.empty(n, dtype=torch.uint8, device=x_true.device)
    _lib.check(lib.ssa_visible_mask_at_f64(_chk(x_true, "x_true"), _chk(trans, "trans"), _chk(env_time, "env_time", torch.int32), int(time_offset),
                                           int(trans.shape[0]), C.byref(consts), _chk(mask, "mask", torch.uint8), None, n, _stream()),
               "ssa_visible_mask_at_f64")
    return mask


def masked_argmax(score, mask=None):
    """index of the first maximum of score[mask != 0] (NaN skipped), -1 if nothing is selected."""
    lib = _lib.load()
    out = torch.empty(2, dtype=torch.int64, device=score.device)
    _lib.check(lib.ssa_masked_argmax_f64(_chk(score, "score"), _chk(mask, "mask", torch.uint8) if mask is not None else None,
                                         score.shape[0], out.data_ptr(), _stream()), "ssa_masked_argmax_f64")
    return int(out[0].item())


def masked_argmax_workspace(n, device):
    """zeroed workspace of ssa_masked_argmax_ws_f64 for up to n entries (one call at a time: keep it with the stream that uses it)"""
    lib = _lib.load()
    return torch.zeros(int(lib.ssa_masked_argmax_workspace_bytes(int(n))) // 8, dtype=torch.int64, device=device)


def masked_argmax_action(score, mask=None, workspace=None):
    """the arg-max head of a device-side policy: np.argmax of `score` over the entries with mask != 0 (first maximum; NaN skipped; -1 when
    nothing is selected) as an int32 CUDA tensor [1] -- the action word the next step launch reads.  ONE launch, nothing leaves the device
    (the kernel writes an int64 pair; its low word IS the int32 action: little endian).  workspace (masked_argmax_workspace): the fold
    spread over the chip instead of one workgroup (3 us instead of 8.6 at 20 000 entries)."""
    lib = _lib.load()
    out = torch.empty(2, dtype=torch.int64, device=score.device)
    m = _chk(mask, "mask", torch.uint8) if mask is not None else None
    if workspace is not None:
        _lib.check(lib.ssa_masked_argmax_ws_f64(_chk(score, "score"), m, score.shape[0], out.data_ptr(), workspace.data_ptr(),
                                                workspace.numel() * 8, _stream()), "ssa_masked_argmax_ws_f64")
    else:
        _lib.check(lib.ssa_masked_argmax_f64(_chk(score, "score"), m, score.shape[0], out.data_ptr(), _stream()), "ssa_masked_argmax_f64")
    return out.view(torch.int32)[:1]


def env_step(consts, params):
    """E1: the fused step.  `params` is a filled _lib.ssa_step_params."""
    lib = _lib.load()
    _lib.check(lib.ssa_env_step_f64(C.byref(consts), C.byref(params), _stream()), "ssa_env_step_f64")


def nees(x_true, x, P):
    """d^T inv(P) d for every row (anees() / fitness_test() of the reference)."""
    lib = _lib.load()
    n = x.shape[0]
    out = torch.empty(n, dtype=f64, device=x.device)
    _lib.check(lib.ssa_nees_f64(_chk(x_true, "x_true"), _chk(x, "x"), _chk(P, "P"), _chk(out, "nees"), n, _stream()), "ssa_nees_f64")
    return out


def nis(y, S):
    """y^T inv(S) y for every row (fitness_test() of the reference)."""
    lib = _lib.load()
    n = y.shape[0]
    out = torch.empty(n, dtype=f64, device=y.device)
    _lib.check(lib.ssa_nis_f64(_chk(y, "y"), _chk(S, "S"), _chk(out, "nis"), n, _stream()), "ssa_nis_f64")
    return out


def chi2_contained(values, lo, hi):
    """(inside, valid) = number of entries strictly inside (lo, hi) and number of non-NaN entries of a float64 device tensor
    (fitness_test()'s chi-square containment, ssa_tasker_simple_2.py:757-760, 770-771)."""
    lib = _lib.load()
    v = values.reshape(-1)
    out = torch.empty(2, dtype=torch.int64, device=v.device)
    _lib.check(lib.ssa_chi2_contained_f64(_chk(v, "values"), v.numel(), float(lo), float(hi), out.data_ptr(), _stream()),
               "ssa_chi2_contained_f64")
    c = out.cpu()
    return int(c[0]), int(c[1])
