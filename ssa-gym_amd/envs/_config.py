"""Resolution of the operator plug points of an env_config dict (envs/__init__.py:23-28 of the reference) into
the fused-kernel variant, shared by SSA_Tasker_Env and SSA_Tasker_VecEnv so that both accept -- and refuse --
exactly the same configurations."""
from . import dynamics

_MODELS = {(("hx", "aer"), ("mean_z", "uvw"), ("residual_z", "aer")): 'aer',
           (("hx", "xyz"), ("mean_z", "xyz"), ("residual_z", "xyz")): 'xyz'}


def resolve_kernel_variant(config):
    """(measurement model 'aer' | 'xyz', propagator 'hybrid' | 'fg' | 'elements' | 'j2') of a config dict.

    The propagator comes from config['propagator'] when given, else from the `fx` token (dynamics.fx_xyz_farnocchia
    -> 'hybrid', the behaviour-faithful variant; fx_xyz_farnocchia_fg -> 'fg', fx_xyz_farnocchia_elements -> 'elements',
    fx_xyz_j2_rk4 -> 'j2'); the reference's own function object of that name maps to 'hybrid'.  Foreign callables and
    hx / mean_z / residual_z combinations without a fused kernel raise NotImplementedError (there is no CPU fallback)."""
    fx_id = dynamics.kernel_id_of(config['fx'], "fx")
    ids = tuple(dynamics.kernel_id_of(config[k], k) for k in ("hx", "mean_z", "residual_z"))
    dynamics.kernel_id_of(config['msqrt'], "msqrt")
    model = _MODELS.get(ids)
    if model is None:
        raise NotImplementedError("hx/mean_z/residual_z combination %s has no fused kernel" % (ids,))
    if fx_id != ("fx", "farnocchia"):
        raise NotImplementedError("fx %r has no fused kernel" % (config['fx'],))
    propagator = config.get('propagator', getattr(dynamics.unwrap_partial(config['fx']), 'propagator', 'hybrid'))
    if propagator not in ('fg', 'elements', 'j2', 'hybrid'):
        raise NotImplementedError("unknown propagator %r" % (propagator,))
    return model, propagator


def resolve_perturbation(config):
    """(J2, R_eq) of the device integrator for this config, or None for the defaults: the acceleration bound to an
    fx_xyz_cowell token (`ad` / ad_kwargs, envs/dynamics.py:168-201 of the reference), or config['ad'] / config['ad_kwargs']
    next to a Cowell / J2 propagator."""
    fx = dynamics.unwrap_partial(config['fx'])
    if config.get('ad') is not None:
        fx = dynamics.fx_xyz_cowell.with_ad(config['ad'], **dict(config.get('ad_kwargs') or {}))
    if hasattr(fx, 'perturbation'):
        return fx.perturbation()
    return None


def kernel_consts(config, Q, R, dt, obs_limit_rad, obs_lla):
    """(ssa_consts, measurement model) for a config dict: the ONE place where SSA_Tasker_Env and SSA_Tasker_VecEnv turn the
    operator tokens and the optional keys `propagator`, `resample_sigmas`, `covariance_form` ('reference' | 'centred';
    default: 'reference' with the 'hybrid' and 'elements' propagators -- the behaviour-faithful variants -- else 'centred'),
    `ad` / `ad_kwargs` into kernel constants."""
    from .. import host
    model, propagator = resolve_kernel_variant(config)
    kw = {}
    pert = resolve_perturbation(config)
    if pert is not None:
        if propagator != 'j2':
            raise NotImplementedError("an acceleration (ad) needs the Cowell / J2 propagator, got %r" % (propagator,))
        kw.update(j2=pert[0], r_eq=pert[1])
    consts = host.make_consts(Q, R, config['alpha'], config['beta'], config['kappa'], dt, obs_limit_rad, obs_lla, obs_type=model,
                              propagator=propagator, resample=bool(config.get('resample_sigmas', False)),
                              update_interval=config['update_interval'], covariance=config.get('covariance_form'), **kw)
    return consts, model
