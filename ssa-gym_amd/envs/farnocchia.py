"""envs/farnocchia.py mirror: the default `fx` the reference binds at envs/__init__.py:6."""
from .dynamics import fx_xyz_farnocchia, fx_xyz_farnocchia_elements  # noqa: F401
