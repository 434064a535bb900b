"""Resolution of the operator plug points of an env_config dict (envs/__init__.py:23-28 of the reference) into
the fused-kernel variant, shared by SSA_Tasker_Env and SSA_Tasker_VecEnv so that both accept -- and refuse --
exactly the same configurations."""
from . import dynamics

_MODELS = {(("hx", "aer"), ("mean_z", "uvw"), ("residual_z", "aer")): 'aer',
           (("hx", "xyz"), ("mean_z", "xyz"), ("residual_z", "xyz")): 'xyz'}


def resolve_kernel_variant(config):
    """(measurement model 'aer' | 'xyz', propagator 'fg' | 'elements' | 'j2') of a config dict.

    The propagator comes from config['propagator'] when given, else from the `fx` token (dynamics.fx_xyz_farnocchia
    -> 'fg', fx_xyz_farnocchia_elements -> 'elements', fx_xyz_j2_rk4 -> 'j2'); the reference's own function object
    of that name maps to 'fg'.  Foreign callables and hx / mean_z / residual_z combinations without a fused kernel
    raise NotImplementedError (there is no CPU fallback)."""
    fx_id = dynamics.kernel_id_of(config['fx'], "fx")
    ids = tuple(dynamics.kernel_id_of(config[k], k) for k in ("hx", "mean_z", "residual_z"))
    dynamics.kernel_id_of(config['msqrt'], "msqrt")
    model = _MODELS.get(ids)
    if model is None:
        raise NotImplementedError("hx/mean_z/residual_z combination %s has no fused kernel" % (ids,))
    if fx_id != ("fx", "farnocchia"):
        raise NotImplementedError("fx %r has no fused kernel" % (config['fx'],))
    propagator = config.get('propagator', getattr(config['fx'], 'propagator', 'fg'))
    if propagator not in ('fg', 'elements', 'j2'):
        raise NotImplementedError("unknown propagator %r" % (propagator,))
    return model, propagator
