"""Physical and sensor constants of the Cops-and-Thieves world.

The reference keeps these in ``pyproject.toml [tool.physical-params]`` and re-reads the
file from the CWD on every getter call (reference ``src/utils/toml_utils.py:40-46``,
values at ``pyproject.toml:12-19``).  Here they are read ONCE: defaults below equal the
reference's values, and :func:`load_physical_params` overlays a ``pyproject.toml`` if the
caller points at one (same table name, same keys).

Sensor constants are hard-coded literals in the reference (``src/agents/entity.py:84-86``,
ray radius at ``entity.py:196``); Space parameters are the Chipmunk2D defaults, which the
reference never changes (``src/environments/base_env.py:77``).
"""
from __future__ import annotations

import dataclasses
import enum
from pathlib import Path


class ObjectType(enum.Enum):
    """Ray-hit classes, same names/values as reference ``src/utils/object_types.py:4-9``."""

    WALL = 0
    COP = 1
    THIEF = 2
    MOVABLE = 3
    EMPTY = 4


@dataclasses.dataclass(frozen=True)
class PhysicalParams:
    unit_velocity: float = 10.0      # impulse per action            (pyproject.toml:13)
    unit_mass: float = 1.0           # agent body mass               (pyproject.toml:14)
    unit_size: float = 5.0           # agent circle radius           (pyproject.toml:15)
    max_speed: float = 125.0         # speed clamp after impulse     (pyproject.toml:16)
    pymunk_cop_category: int = 42    # shape-filter category         (pyproject.toml:17)
    pymunk_thief_category: int = 2137  #                             (pyproject.toml:18)
    termination_radius: float = 20.0  # capture distance, strict <   (pyproject.toml:19)


@dataclasses.dataclass(frozen=True)
class SensorParams:
    num_rays: int = 90               # entity.py:86 (BASELINE throughput configs use 64)
    ray_length: float = 400.0        # entity.py:84
    ray_radius: float = 1.0          # entity.py:196 (swept-circle radius of every ray)
    fov: float = 6.283185307179586   # entity.py:85 (2*pi)


@dataclasses.dataclass(frozen=True)
class SpaceParams:
    """Chipmunk2D ``cpSpaceInit`` defaults [CHIPMUNK-RECALL]."""

    iterations: int = 10
    collision_slop: float = 0.1
    collision_bias: float = 0.001797010299914434  # pow(1.0 - 0.1, 60.0)
    collision_persistence: int = 3
    wall_radius: float = 1.0         # pymunk.Poly(..., radius=1)    (src/maps/map.py:127)


_KEYS = tuple(f.name for f in dataclasses.fields(PhysicalParams))


def load_physical_params(pyproject: str | Path | None = None) -> PhysicalParams:
    """Reference-compatible read of ``[tool.physical-params]`` (done once, not per call)."""
    if pyproject is None:
        return PhysicalParams()
    try:
        import tomllib as _toml  # py311+
    except ModuleNotFoundError:  # py310 in this image
        import tomli as _toml
    with open(pyproject, "rb") as f:
        table = _toml.load(f)["tool"]["physical-params"]
    return PhysicalParams(**{k: table[k] for k in _KEYS if k in table})


DEFAULT_PHYSICAL = PhysicalParams()
DEFAULT_SENSOR = SensorParams()
DEFAULT_SPACE = SpaceParams()
