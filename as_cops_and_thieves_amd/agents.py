"""Agent descriptors: the host-side face of reference ``src/agents/{entity,cop,thief}.py``.

In the reference an ``Entity`` owns a Pymunk body and performs its own ray casts; here all of
that lives in the device kernels, and these objects carry what callers still read from an
agent: id, role, spaces, category, observation priorities, and a view of its body state.
"""
from __future__ import annotations

from typing import List, Tuple

import numpy as np

from . import spaces
from .constants import DEFAULT_PHYSICAL, ObjectType, PhysicalParams


class _BodyView:
    """``agent.body.position`` / ``.velocity`` read-through to the env's device state."""

    def __init__(self, env, index: int):
        self._env, self._index = env, index

    @property
    def position(self) -> Tuple[float, float]:
        return tuple(self._env._host_state("pos")[0, self._index].tolist())

    @property
    def velocity(self) -> Tuple[float, float]:
        return tuple(self._env._host_state("vel")[0, self._index].tolist())


class Entity:
    """Same public surface as reference ``Entity`` minus the Pymunk objects
    (``entity.py:41-124,243-247``)."""

    role = "entity"
    observation_priorities: List[ObjectType] = []

    def __init__(self, env, index: int, id: str, group: int, filter_category: int, num_rays: int,
                 ray_length: float, physical: PhysicalParams = DEFAULT_PHYSICAL):
        self._id, self._index, self.group = id, index, group
        self.filter_category = filter_category
        self._radius, self._speed, self._mass = physical.unit_size, physical.unit_velocity, physical.unit_mass
        self._max_speed = physical.max_speed
        self._ray_length, self._fov, self._num_rays = ray_length, 2 * np.pi, num_rays
        self.action_space = spaces.Discrete(4, start=0)                       # entity.py:88-90
        self.observation_space = spaces.Dict({                               # entity.py:92-107
            "distance": spaces.Box(low=0.0, high=ray_length, shape=(num_rays,), dtype=np.float16),
            "object_type": spaces.Box(low=0, high=max(t.value for t in ObjectType), shape=(num_rays,),
                                      dtype=np.uint8),
        })
        self.body = _BodyView(env, index)

    def get_radius(self) -> float:
        return self._radius

    def get_id(self) -> str:
        return self._id


class Cop(Entity):
    role = "cop"
    observation_priorities = [ObjectType.THIEF, ObjectType.MOVABLE, ObjectType.COP, ObjectType.WALL,
                              ObjectType.EMPTY]      # cop.py:41-47 (no observable effect)


class Thief(Entity):
    role = "thief"
    observation_priorities = [ObjectType.COP, ObjectType.MOVABLE, ObjectType.THIEF, ObjectType.WALL,
                              ObjectType.EMPTY]      # thief.py:40-46
