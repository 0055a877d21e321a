"""A single-env Cops-and-Thieves scene driven through the **Pymunk API itself** -- TEST INFRASTRUCTURE ONLY.

Purpose: the one thing that can PIN the oracle (and through it the HIP kernels).  Pymunk / Chipmunk2D is a third-party
dependency of the reference that this image does not hold (``import pymunk`` -> ModuleNotFoundError), so today every user of
this module skips; on a box that has the package, ``tests/test_pymunk_pin.py`` runs the oracle against it and
``bench.py``'s ``cpu_baseline`` gains the single-process Pymunk figure BASELINE.md asks for.

This is own code, not the reference's files: it makes the same Pymunk calls, in the same order, with the same arguments as the
reference does (cited per line: REF = /root/reference/src), and nothing else of the reference (no gymnasium / PettingZoo / pygame
surface).  Written against the Pymunk 6.x API from its documentation; it has never been executed in this image.
"""
from __future__ import annotations

import itertools
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import cat_oracle_host as host

WALL, COP, THIEF, MOVABLE, EMPTY = 0, 1, 2, 3, 4            # REF utils/object_types.py:4-9


def available() -> bool:
    try:
        import pymunk  # noqa: F401
        return True
    except Exception:   # noqa: BLE001 - absent, or present without its native library
        return False


class PymunkScene:
    """One ``BaseEnv`` worth of Pymunk objects and the reference's per-tick call sequence."""

    def __init__(self, map_data: dict, n_rays: int = host.NUM_RAYS, max_step_count: int = host.MAX_STEP_COUNT,
                 dt: float = host.DT, roster=None, start_positions=None, scale=None):
        import pymunk
        self.pm = pymunk
        self.space = pymunk.Space()                                                     # REF environments/base_env.py:77
        rings = host.map_rings(map_data)
        if scale is not None:
            rings = [[(x * scale[0], y * scale[1]) for x, y in r] for r in rings]
        self.shape_ids = {}                                                             # shape -> its place in the oracle's shape list (walls in file order, then agents)
        for ring in rings:                                                              # REF maps/map.py:124-128 populate_space
            wall = pymunk.Poly(self.space.static_body, ring, radius=1)
            self.shape_ids[wall] = len(self.shape_ids)
            self.space.add(wall)
        self.n_walls = len(self.shape_ids)
        n_cops, n_thieves, starts, _ = host.agent_tables(map_data, roster, start_positions, None)
        self.n_cops, self.n_thieves, self.A = n_cops, n_thieves, n_cops + n_thieves
        self.R, self.max_step_count, self.dt = n_rays, max_step_count, dt
        self.step_count = 0
        self.last_hit_shape = {}
        self.bodies, self.shapes, self.ray_filters, self.categories = [], [], [], []
        group = itertools.count(1)                                                      # base_env.py:89: cops first, then thieves
        for i, xy in enumerate(starts):
            cat = host.COP_CATEGORY if i < n_cops else host.THIEF_CATEGORY
            g = next(group)
            body = pymunk.Body(host.UNIT_MASS, pymunk.moment_for_circle(host.UNIT_MASS, inner_radius=0.0,
                                                                       outer_radius=host.UNIT_SIZE))   # REF agents/entity.py:109-114
            body.position = pymunk.Vec2d(float(xy[0]), float(xy[1]))                    # entity.py:115
            circle = pymunk.Circle(body, radius=host.UNIT_SIZE)                         # entity.py:116
            circle.filter = pymunk.ShapeFilter(group=g, categories=cat)                 # entity.py:118
            self.ray_filters.append(pymunk.ShapeFilter(group=g, categories=cat))        # entity.py:120-123
            self.space.add(body, circle)                                                # entity.py:124
            self.shape_ids[circle] = self.n_walls + i
            self.bodies.append(body); self.shapes.append(circle); self.categories.append(cat)
        self.force = {0: pymunk.Vec2d(-host.UNIT_VELOCITY, 0), 1: pymunk.Vec2d(0, host.UNIT_VELOCITY),
                      2: pymunk.Vec2d(host.UNIT_VELOCITY, 0), 3: pymunk.Vec2d(0, -host.UNIT_VELOCITY)}   # entity.py:76-81

    # -- Entity.reset (entity.py:148-157): position + velocity setters only; Chipmunk's shape caches stay stale
    def reset(self, positions: Sequence[Tuple[float, float]]) -> None:
        for body, p in zip(self.bodies, positions):
            body.position = self.pm.Vec2d(float(p[0]), float(p[1]))
            body.velocity = self.pm.Vec2d(0, 0)
        self.step_count = 0                                                             # base_env.py:350

    def spawn_is_free(self, agent: int, xy) -> bool:
        """base_env.py:153-158: ``not space.point_query_nearest(pos, entity.get_radius(), entity.ray_filter)``."""
        return not self.space.point_query_nearest(self.pm.Vec2d(float(xy[0]), float(xy[1])), host.UNIT_SIZE, self.ray_filters[agent])

    # -- Entity.get_observation (entity.py:182-215)
    def observe(self, i: int):
        pm = self.pm
        angles = np.linspace(0, host.FOV, self.R, endpoint=False)
        cosines, sines = np.cos(angles), np.sin(angles)
        origin = self.bodies[i].position
        ox, oy = origin[0], origin[1]
        endpoints = np.column_stack((ox + host.RAY_LENGTH * cosines, oy + host.RAY_LENGTH * sines))
        hits = [self.space.segment_query_first(origin, pm.Vec2d(*end), 1, self.ray_filters[i]) for end in endpoints]
        distances = np.full(self.R, host.RAY_LENGTH, dtype=np.float16)
        types = np.full(self.R, EMPTY, dtype=np.uint8)
        # which shape each ray's query returned (the oracle's `hit_shape`): what pins the ORDER of Chipmunk's visits (DESIGN D2)
        self.last_hit_shape[i] = np.array([-1 if h is None else self.shape_ids.get(h.shape, -2) for h in hits], dtype=np.int32)
        idx = [k for k, h in enumerate(hits) if h is not None]
        if idx:
            pts = np.array([hits[k].point for k in idx], dtype=np.float16)
            dx, dy = pts[:, 0] - ox, pts[:, 1] - oy
            distances[idx] = np.hypot(dx, dy).astype(np.float16)
            types[idx] = np.array([self._query_body(hits[k].shape) for k in idx], dtype=np.uint8)
        return distances, types

    def _query_body(self, shape) -> int:                                                 # entity.py:226-241
        pm = self.pm
        if shape.body.body_type == pm.Body.DYNAMIC:
            if isinstance(shape, pm.Poly):
                return MOVABLE
            if isinstance(shape, pm.Circle):
                return THIEF if shape.filter.categories == host.THIEF_CATEGORY else COP
        return WALL

    # -- BaseEnv._termination_criterion (base_env.py:521-554)
    def termination(self) -> Tuple[bool, bool]:
        pm = self.pm
        for t in range(self.n_cops, self.A):
            for c in range(self.n_cops):
                flt = pm.ShapeFilter(mask=~(self.categories[t] | self.categories[c]) & 0xFFFFFFFF)
                hit = self.space.segment_query_first(self.bodies[t].position, self.bodies[c].position, 0.0, shape_filter=flt)
                if hit is None and self.bodies[t].position.get_distance(self.bodies[c].position) < host.TERMINATION_RADIUS:
                    return True, False
        if self.step_count >= self.max_step_count:
            return False, True
        return False, False

    @staticmethod
    def _cop_reward(dist, types, term) -> float:                                          # REF agents/cop.py:61-75
        if term[0]:
            return 1.0
        if term[1]:
            return -1.0
        reward = -0.02
        mask = types == THIEF
        if mask.any():
            reward += 1.5 * np.exp(-dist[mask].min() / 50.0)
        else:
            reward -= 0.02
        return float(reward)

    @staticmethod
    def _thief_reward(dist, types, term) -> float:                                        # REF agents/thief.py:58-69
        if term[0]:
            return -1.0
        if term[1]:
            return 1.0
        mask = types == COP
        if mask.any():
            return float(np.tanh((np.min(dist[mask]) - 100.0) / 50.0) / 10.0)
        return 0.15

    # -- BaseEnv.step (base_env.py:378-413)
    def step(self, actions: Sequence[int]) -> dict:
        self.step_count += 1
        term = self.termination()
        dist = np.zeros((self.A, self.R), np.float16)
        types = np.zeros((self.A, self.R), np.uint8)
        reward = np.zeros(self.A, np.float32)
        for i in range(self.A):                                                          # Entity.step: act, observe, reward (entity.py:136-146)
            body = self.bodies[i]
            body.apply_impulse_at_local_point(self.force[int(actions[i])])               # entity.py:131
            if abs(body.velocity) > host.MAX_SPEED:                                      # entity.py:132-134
                body.velocity = body.velocity.normalized() * host.MAX_SPEED
            dist[i], types[i] = self.observe(i)
            reward[i] = (self._cop_reward if i < self.n_cops else self._thief_reward)(dist[i], types[i], term)
        self.space.step(self.dt)                                                         # base_env.py:392
        done = term[0] or term[1]
        return {"obs_distance": dist.view(np.uint16), "obs_type": types, "reward": reward, "terminated": int(done),
                "truncated": int(term[1]), "winner": (0 if term[0] else 1) if done else -1,
                "pos": np.array([[b.position.x, b.position.y] for b in self.bodies]),
                "vel": np.array([[b.velocity.x, b.velocity.y] for b in self.bodies])}


def time_random_rollout(map_data: dict, n_rays: int, seconds: float, roster=None, start_positions=None, scale=None,
                        spawn: Optional[List[Tuple[float, float]]] = None) -> Tuple[float, int]:
    """Env-steps/s of ONE Pymunk scene stepping with uniformly random actions and restarting finished episodes at ``spawn``
    (default: the start positions) -- the single-process Pymunk baseline of BASELINE.md section 3."""
    import time
    sc = PymunkScene(map_data, n_rays, roster=roster, start_positions=start_positions, scale=scale)
    home = spawn or [(b.position.x, b.position.y) for b in sc.bodies]
    rng = np.random.default_rng(0)
    sc.reset(home)
    t0, ticks = time.perf_counter(), 0
    while time.perf_counter() - t0 < seconds:
        out = sc.step(rng.integers(0, 4, sc.A))
        if out["terminated"]:
            sc.reset(home)
        ticks += 1
    return ticks / (time.perf_counter() - t0), ticks
