"""Loader for the reference's importable hot-path modules (build container only).

TEST INFRASTRUCTURE: used only by the golden-vector generator scripts in this
directory.  Nothing here (and nothing under /root/reference) travels to the
GPU box; the committed *.npz / *.npy files are the product of these scripts.

Recipe: the reference's `envs/farnocchia.py`,
`envs/transformations.py` and `envs/dynamics.py` are loaded from where they lie
under /root/reference with
  * an identity `numba` module (numba is not importable here) so that
    `@njit` / `@jit([...])` leave the plain Python functions in place,
  * a bare `envs` namespace package so `envs/__init__.py` (gym, cwd walk) is
    not executed,
  * placeholder modules for third-party imports the hot functions never call
    (poliastro propagators, pymap3d, astropy.units/coordinates).
The functions that then run are the reference's own source, unmodified.
"""
import importlib
import importlib.util
import sys
import types
import warnings

REF = "/root/reference"


def _identity_numba():
    nb = types.ModuleType("numba")

    def njit(*a, **k):
        if len(a) == 1 and callable(a[0]) and not k:
            return a[0]
        return lambda f: f

    nb.njit = njit
    nb.jit = njit
    sys.modules["numba"] = nb


def load_farnocchia():
    """reference envs/farnocchia.py (pure numpy once numba is the identity)."""
    warnings.filterwarnings("ignore")
    _identity_numba()
    spec = importlib.util.spec_from_file_location(
        "ref_farnocchia", REF + "/envs/farnocchia.py")
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def load_transformations_and_dynamics():
    """reference envs/transformations.py + envs/dynamics.py (needs pyerfa: run
    under /opt/conda/bin/python3.9)."""
    warnings.filterwarnings("ignore")
    _identity_numba()
    import astropy  # noqa: F401  (real package; provides astropy._erfa shim)
    import erfa  # noqa: F401

    envs = types.ModuleType("envs")
    envs.__path__ = [REF + "/envs"]
    sys.modules["envs"] = envs

    class _U:
        def __pow__(self, o): return self
        def __truediv__(self, o): return self
        def __mul__(self, o): return self
        __rmul__ = __mul__
        __rtruediv__ = __truediv__

    au = types.ModuleType("astropy.units")
    for name in ("m", "s", "km", "rad", "deg"):
        setattr(au, name, _U())
    sys.modules["astropy.units"] = au
    astropy.units = au
    ac = types.ModuleType("astropy.coordinates")
    for name in ("SkyCoord", "EarthLocation", "AltAz", "ITRS"):
        setattr(ac, name, object)
    sys.modules["astropy.coordinates"] = ac

    def _mod(name, **attrs):
        mod = types.ModuleType(name)
        for k, v in attrs.items():
            setattr(mod, k, v)
        sys.modules[name] = mod
        return mod

    class _Q:
        def __init__(self, v): self.v = v
        def to_value(self, unit=None): return self.v

    class _Earth:
        k = _Q(398600441800000.0)
        R_mean = _Q(6371008.4)
        R = _Q(6378136.6)

    _mod("poliastro")
    _mod("poliastro.core")
    _mod("poliastro.core.propagation", markley=None, vallado=None, pimienta=None,
         gooding=None, danby=None, farnocchia=None, mikkola=None, func_twobody=None)
    _mod("poliastro.core.elements", coe2rv=None, rv2coe=None)
    _mod("poliastro.bodies", Earth=_Earth)
    _mod("pymap3d")
    tr = importlib.import_module("envs.transformations")
    dy = importlib.import_module("envs.dynamics")
    return tr, dy


def load_eops():
    """EOP table parsed from the file vendored in the reference tree, exactly as
    transformations.get_eops() would build it (that function itself opens an
    ftp:// URL and is not called)."""
    import numpy as np
    import pandas as pd
    path = REF + "/hpiers.obspm.fr/iers/eop/eopc04/eopc04_IAU2000.62-now"
    array = np.genfromtxt(path, skip_header=14)
    headers = ['Year', 'Month', 'Day', 'MJD', 'x', 'y', 'UT1-UTC', 'LOD', 'dX',
               'dY', 'x Err', 'y Err', 'UT1-UTC Err', 'LOD Err', 'dX Err', 'dY Err']
    return pd.DataFrame(data=array, index=array[:, 3], columns=headers)
