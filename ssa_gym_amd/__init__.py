"""Import alias: the product package lives in the directory `ssa-gym_amd/` (the
name the project layout prescribes, which is not a valid Python identifier);
`import ssa_gym_amd` resolves to it."""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "ssa-gym_amd")]
__file__ = _os.path.join(__path__[0], "__init__.py")
with open(__file__) as _f:
    exec(compile(_f.read(), __file__, "exec"))
