"""Simulation configuration shared by every host-side entry point.

One flat record; field order matches ``cat_config`` in ``include/cat_sim.h``.  Defaults are the
values both reference drivers run with: ``SimpleEnv`` (``src/environments/simple_env.py:14-38``:
dt = 1/60, max_step_count = 400), the physical constants of ``pyproject.toml:12-19``, the sensor
literals of ``src/agents/entity.py:84-86,196`` and Chipmunk2D's untouched Space defaults.
"""
from __future__ import annotations

import dataclasses
import math

from .constants import DEFAULT_PHYSICAL, DEFAULT_SENSOR, DEFAULT_SPACE, PhysicalParams, SensorParams, SpaceParams


@dataclasses.dataclass
class SimConfig:
    n_envs: int = 1
    n_cops: int = 2
    n_thieves: int = 1
    n_rays: int = DEFAULT_SENSOR.num_rays
    max_step_count: int = 400
    iterations: int = DEFAULT_SPACE.iterations
    persistence: int = DEFAULT_SPACE.collision_persistence
    bbtree_gate: int = 1
    env_id_offset: int = 0
    seed: int = 1
    dt: float = 1.0 / 60.0
    bias_coef: float = 0.0            # derived in __post_init__
    slop: float = DEFAULT_SPACE.collision_slop
    ray_length: float = DEFAULT_SENSOR.ray_length
    ray_radius: float = DEFAULT_SENSOR.ray_radius
    agent_radius: float = DEFAULT_PHYSICAL.unit_size
    agent_mass: float = DEFAULT_PHYSICAL.unit_mass
    impulse: float = DEFAULT_PHYSICAL.unit_velocity
    max_speed: float = DEFAULT_PHYSICAL.max_speed
    termination_radius: float = DEFAULT_PHYSICAL.termination_radius
    wall_radius: float = DEFAULT_SPACE.wall_radius
    collision_bias: float = DEFAULT_SPACE.collision_bias

    def __post_init__(self) -> None:
        # cpSpaceStep: biasCoef = 1 - pow(collisionBias, dt); evaluated on the host so that the
        # device never calls pow()
        self.bias_coef = 1.0 - math.pow(self.collision_bias, self.dt)

    @property
    def n_agents(self) -> int:
        return self.n_cops + self.n_thieves

    @property
    def sensor(self) -> SensorParams:
        return SensorParams(num_rays=self.n_rays, ray_length=self.ray_length,
                            ray_radius=self.ray_radius)

    @classmethod
    def from_params(cls, physical: PhysicalParams = DEFAULT_PHYSICAL,
                    sensor: SensorParams = DEFAULT_SENSOR, space: SpaceParams = DEFAULT_SPACE,
                    **kw) -> "SimConfig":
        return cls(n_rays=sensor.num_rays, ray_length=sensor.ray_length, ray_radius=sensor.ray_radius,
                   agent_radius=physical.unit_size, agent_mass=physical.unit_mass,
                   impulse=physical.unit_velocity, max_speed=physical.max_speed,
                   termination_radius=physical.termination_radius, iterations=space.iterations,
                   persistence=space.collision_persistence, slop=space.collision_slop,
                   wall_radius=space.wall_radius, collision_bias=space.collision_bias, **kw)


# field order of the C struct (collision_bias stays host-side)
C_FIELDS_I32 = ("n_envs", "n_cops", "n_thieves", "n_rays", "max_step_count", "iterations",
                "persistence", "bbtree_gate")
C_FIELDS_F64 = ("dt", "bias_coef", "slop", "ray_length", "ray_radius", "agent_radius", "agent_mass",
                "impulse", "max_speed", "termination_radius", "wall_radius")
