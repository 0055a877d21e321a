"""Host side of the CPU oracle -- TEST INFRASTRUCTURE ONLY (PARITY UNPINNED, see cat_oracle.h).

Everything the C restatement is fed at construction is built HERE, from the raw map JSON and from literal constants,
without importing the product package (``as_cops_and_thieves_amd.{maps,tables,config,constants}``): block parsing,
canvas scaling, convex hulls, splitting planes, bounding boxes, roster / spawn-region tables, the geometry blob, the ray
direction table, the two reward tables and the config record.  The product builds the same things with its own code
(monotone-chain hull, vectorised NumPy tables); ``tests/test_oracle_independent_host.py`` requires the two to be
byte-identical for the five maps, so a mistake on either side shows up as a difference instead of cancelling out.

Citations: REF = /root/reference/src, CP = Chipmunk2D 7.0.x as published (restated from recall, not from a checkout).
"""
from __future__ import annotations

import json
import math
import struct
import sys
from pathlib import Path
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

HERE = Path(__file__).resolve().parent
RAW_MAPS = HERE.parent / "tests" / "golden" / "reference_maps.json"   # the reference's own map data, tools/make_golden_maps_raw.py
_raw_cache: Dict[str, dict] = {}


def raw_map(name: str) -> Optional[dict]:
    """The reference's map file `name`.json as parsed JSON (None: not one of its maps)."""
    if not _raw_cache and RAW_MAPS.exists():
        _raw_cache.update(json.loads(RAW_MAPS.read_text())["maps"])
    return _raw_cache.get(name)

# ---- constants, each with the place the reference (or Chipmunk) states it ------------------------------------------------
UNIT_VELOCITY = 10.0          # pyproject.toml:13  impulse per action
UNIT_MASS = 1.0               # pyproject.toml:14
UNIT_SIZE = 5.0               # pyproject.toml:15  agent circle radius
MAX_SPEED = 125.0             # pyproject.toml:16
COP_CATEGORY = 42             # pyproject.toml:17
THIEF_CATEGORY = 2137         # pyproject.toml:18
TERMINATION_RADIUS = 20.0     # pyproject.toml:19
RAY_LENGTH = 400.0            # REF agents/entity.py:84
FOV = 2.0 * math.pi           # REF agents/entity.py:85
NUM_RAYS = 90                 # REF agents/entity.py:86
RAY_RADIUS = 1.0              # REF agents/entity.py:196 (third argument of segment_query_first)
WALL_RADIUS = 1.0             # REF maps/map.py:127  pymunk.Poly(..., radius=1)
DT = 1.0 / 60.0               # REF environments/simple_env.py:20
MAX_STEP_COUNT = 400          # REF environments/simple_env.py:19
ITERATIONS = 10               # CP cpSpaceInit: iterations
COLLISION_SLOP = 0.1          # CP cpSpaceInit: collisionSlop
COLLISION_BIAS = math.pow(1.0 - 0.1, 60.0)   # CP cpSpaceInit: collisionBias = cpfpow(1.0f - 0.1f, 60.0f)
COLLISION_PERSISTENCE = 3     # CP cpSpaceInit: collisionPersistence

BLOB_MAGIC, BLOB_VERSION = 0x31544143, 1     # "CAT1": cat_oracle.c parse_blob
CONFIG_I32 = ("n_envs", "n_cops", "n_thieves", "n_rays", "max_step_count", "iterations", "persistence", "bbtree_gate")
CONFIG_F64 = ("dt", "bias_coef", "slop", "ray_length", "ray_radius", "agent_radius", "agent_mass", "impulse", "max_speed",
              "termination_radius", "wall_radius")           # field order of cato_config (cat_oracle.h)
CONFIG_DEFAULTS = dict(n_envs=1, n_cops=2, n_thieves=1, n_rays=NUM_RAYS, max_step_count=MAX_STEP_COUNT, iterations=ITERATIONS,
                       persistence=COLLISION_PERSISTENCE, bbtree_gate=1, dt=DT, slop=COLLISION_SLOP, ray_length=RAY_LENGTH,
                       ray_radius=RAY_RADIUS, agent_radius=UNIT_SIZE, agent_mass=UNIT_MASS, impulse=UNIT_VELOCITY,
                       max_speed=MAX_SPEED, termination_radius=TERMINATION_RADIUS, wall_radius=WALL_RADIUS,
                       collision_bias=COLLISION_BIAS, env_id_offset=0, seed=1)

Vec = Tuple[float, float]


def config_values(cfg) -> Dict[str, float]:
    """The C config record from any object carrying the workload fields.  ``bias_coef`` is never taken from the caller:
    CP cpSpaceStep -> cpArbiterPreStep gets ``1 - cpfpow(space->collisionBias, dt)``, evaluated here."""
    v = {k: getattr(cfg, k, d) for k, d in CONFIG_DEFAULTS.items()}
    v["bias_coef"] = 1.0 - math.pow(v["collision_bias"], v["dt"])
    return v


# ---- map parsing (REF maps/map.py:35-61 _parse_block, :63-117 _parse_json_map) -------------------------------------------
def block_ring(blk: dict) -> List[Vec]:
    kind = blk.get("type", "rect")
    if kind == "rect":                                   # map.py:37-52: x, y required; w, h default to 1 (None too); negatives allowed
        x, y = blk.get("x"), blk.get("y")
        if x is None or y is None:
            raise ValueError("x and y coordinates are required for rectangle blocks.")
        w, h = blk.get("w"), blk.get("h")
        w = 1 if w is None else w
        h = 1 if h is None else h
        pts = [(x, y), (x + w, y), (x + w, y + h), (x, y + h), (x, y)]
    elif kind == "poly":                                 # map.py:53-60; shapely.Polygon(vs).exterior.coords is a closed ring (:61,126)
        vs = blk.get("vs")
        if vs is None:
            raise ValueError("Vertices are required for polygon blocks.")
        pts = [(v.get("x"), v.get("y")) for v in vs]
        if pts and pts[0] != pts[-1]:
            pts.append(pts[0])
    else:
        raise ValueError(f"Unknown block type: {kind}")
    return [(float(px), float(py)) for px, py in pts]


def map_rings(map_data: dict) -> List[List[Vec]]:
    if "objects" in map_data:                            # the reference's schema
        return [block_ring(b) for b in map_data["objects"]["blocks"]]
    return [[(float(r[i]), float(r[i + 1])) for i in range(0, len(r), 2)] for r in map_data["rings"]]   # bundled compact schema


def agent_tables(map_data: dict, roster, start_positions, spawn_regions):
    """-> (n_cops, n_thieves, start positions cops-then-thieves, spawn regions per agent id).  Agent ids count per type in
    file order (map.py:80-84); the env lists cops first, then thieves (REF environments/base_env.py:96)."""
    if roster is not None:
        agents = [{"type": t, "x": p[0], "y": p[1]} for t, p in zip(roster, start_positions)]
    else:
        agents = map_data["agents"]                      # KeyError for labyrinth.json, as map.py:75
    seen: Dict[str, int] = {}
    regions: Dict[str, List[dict]] = {}
    for a in agents:
        k = seen.get(a["type"], 0)
        aid = f"{a['type']}_{k}"
        if "spawn_regions" in a:                         # map.py:86-101: a list of dicts, or one dict
            d = a["spawn_regions"]
            if isinstance(d, list) and all(isinstance(it, dict) for it in d):
                regions[aid] = d
            elif isinstance(d, dict):
                regions[aid] = [d]
        elif "spawn_region" in a:                        # map.py:102-110
            if isinstance(a["spawn_region"], dict):
                regions[aid] = [a["spawn_region"]]
        seen[a["type"]] = k + 1
    if spawn_regions is not None:
        regions.update(spawn_regions)
    cops = [(a["x"], a["y"]) for a in agents if a["type"] == "cop"]
    thieves = [(a["x"], a["y"]) for a in agents if a["type"] == "thief"]
    return len(cops), len(thieves), cops + thieves, regions


# ---- CP cpConvexHull(count, verts, result, first, tol = 0): QuickHull -----------------------------------------------------
def _cross(a: Vec, b: Vec) -> float:
    return a[0] * b[1] - a[1] * b[0]


def _sub(a: Vec, b: Vec) -> Vec:
    return (a[0] - b[0], a[1] - b[1])


def _qhull_partition(v: List[Vec], lo: int, count: int, a: Vec, b: Vec) -> int:
    """[CP QHullPartition, tol 0] in place on v[lo : lo + count]: the points strictly right of a->b (cross(p - a, b - a) > 0) are
    moved to the front, the FIRST of those farthest from the line to position lo (``value > max`` is strict); returns their number."""
    if count == 0:
        return 0
    best, pivot = 0.0, 0
    delta = _sub(b, a)
    head, tail = 0, count - 1
    while head <= tail:
        value = _cross(_sub(v[lo + head], a), delta)
        if value > 0.0:
            if value > best:
                best, pivot = value, head
            head += 1
        else:
            v[lo + head], v[lo + tail] = v[lo + tail], v[lo + head]
            tail -= 1
    if pivot != 0:
        v[lo], v[lo + pivot] = v[lo + pivot], v[lo]
    return head


def _qhull_reduce(v: List[Vec], lo: int, count: int, a: Vec, pivot: Optional[Vec], b: Vec, result: List[Vec]) -> None:
    """[CP QHullReduce] appends the hull vertices strictly after a up to (excluding) b; ``pivot`` is known to be one of them."""
    if count < 0:
        return
    if count == 0:
        result.append(pivot)
        return
    left = _qhull_partition(v, lo, count, a, pivot)
    _qhull_reduce(v, lo + 1, left - 1, a, v[lo] if left > 0 else None, pivot, result)
    result.append(pivot)
    right = _qhull_partition(v, lo + left, count - left, pivot, b)
    _qhull_reduce(v, lo + left + 1, right - 1, pivot, v[lo + left] if right > 0 else None, b, result)


def convex_hull(points: Sequence[Vec]) -> List[Vec]:
    """[CP cpConvexHull(count, verts, result, NULL, tol = 0)], restated with its in-place swaps because its result depends on
    them in one case (below).  Starts at the minimum vertex in (x, then y) order [CP cpLoopIndexes], runs to the maximum through
    the points on the right of that chord and back through those on its left: counter-clockwise in Chipmunk's y-up sense.
    Duplicates and points collinear with a chord are dropped (``value > 0`` is strict) -- EXCEPT that among several points
    equally far from a chord (three collinear boundary points on a line parallel to it) the first one met becomes the pivot, and a
    pivot is always emitted, even the middle one of the three.  Such a redundant vertex splits one face into two coplanar ones;
    none of the five maps has the case (the product's strict hull and this one agree on all 149 blocks: D7 in DESIGN.md)."""
    v = [(float(x), float(y)) for x, y in points]
    start = end = 0
    lo = hi = v[0]
    for i in range(1, len(v)):                            # [CP cpLoopIndexes]
        p = v[i]
        if p[0] < lo[0] or (p[0] == lo[0] and p[1] < lo[1]):
            lo, start = p, i
        elif p[0] > hi[0] or (p[0] == hi[0] and p[1] > hi[1]):
            hi, end = p, i
    if start == end:
        return [v[0]]
    v[0], v[start] = v[start], v[0]
    e = start if end == 0 else end
    v[1], v[e] = v[e], v[1]
    a, b = v[0], v[1]
    hull = [a]
    _qhull_reduce(v, 2, len(v) - 2, a, b, a, hull)
    return hull


def hull_planes(hull: List[Vec]) -> np.ndarray:
    """[CP cpPolyShapeSetVerts / SetVerts + cpPolyShapeSegmentQuery's per-plane terms] plane i belongs to the edge v[i-1] -> v[i]:
    n = cpvnormalize(cpvrperp(v[i] - v[i-1])), v0 = v[i]; record = n, v0, dot(v0, n), cross(n, v0 of plane i-1), cross(n, v0), 0."""
    k = len(hull)
    rec = np.zeros((k, 8))
    dbl_min = sys.float_info.min
    for i in range(k):
        a, b = hull[i - 1], hull[i]
        e = _sub(b, a)
        rp = (e[1], -e[0])                                              # cpvrperp
        f = 1.0 / (math.sqrt(rp[0] * rp[0] + rp[1] * rp[1]) + dbl_min)  # cpvnormalize: cpvmult(v, 1/(cpvlength(v) + CPFLOAT_MIN))
        n = (rp[0] * f, rp[1] * f)
        rec[i] = (n[0], n[1], b[0], b[1], b[0] * n[0] + b[1] * n[1], _cross(n, a), _cross(n, b), 0.0)
    return rec


def compile_blob(map_data: dict, roster=None, start_positions=None, spawn_regions=None, scale=None,
                 wall_radius: float = WALL_RADIUS) -> bytes:
    """Raw map JSON (+ the build-side overrides of SURVEY 0.2 / Q9, as plain data) -> the geometry blob cat_oracle.c parses."""
    rings = map_rings(map_data)
    if scale is not None:                                 # canvas -> window units (declared by the reference, map.py:24-25, never applied)
        rings = [[(x * scale[0], y * scale[1]) for x, y in r] for r in rings]
    bbs, first, count, planes = [], [], [], []
    for ring in rings:
        hull = convex_hull(ring)
        if len(hull) < 3:
            raise ValueError(f"degenerate block (hull has {len(hull)} vertices): {ring}")
        xs, ys = [p[0] for p in hull], [p[1] for p in hull]
        bbs.append((min(xs) - wall_radius, min(ys) - wall_radius, max(xs) + wall_radius, max(ys) + wall_radius))   # CP cpPolyShapeCacheData
        first.append(sum(count))
        count.append(len(hull))
        planes.append(hull_planes(hull))
    n_cops, n_thieves, starts, regions = agent_tables(map_data, roster, start_positions, spawn_regions)
    ids = [f"cop_{i}" for i in range(n_cops)] + [f"thief_{i}" for i in range(n_thieves)]
    off, regs = [0], []
    for aid in ids:
        for r in regions.get(aid) or []:
            regs.append((r["x"], r["y"], r["w"], r["h"]))
        off.append(len(regs))
    window = tuple(map_data["window"].values())
    f64 = np.concatenate([np.asarray(window[:2], np.float64), np.asarray(bbs, np.float64).ravel(),
                          np.concatenate(planes).ravel() if planes else np.zeros(0), np.asarray(starts, np.float64).ravel(),
                          np.asarray(regs, np.float64).ravel()]).astype("<f8")
    i32 = np.asarray(first + count + off, dtype="<i4")
    hdr = struct.pack("<16i", BLOB_MAGIC, BLOB_VERSION, len(rings), int(sum(count)), n_cops + n_thieves, n_cops, n_thieves,
                      len(regs), *([0] * 8))
    return hdr + f64.tobytes() + i32.tobytes()


def blob_for(cmap) -> bytes:
    """A product ``CompiledMap`` is taken only as a pointer to its INPUTS (``cmap.spec``: the parsed map JSON and the override
    arguments ``Map`` was given); nothing it computed is read.  For the five bundled maps the JSON is re-read from the reference's
    own file (tests/golden/reference_maps.json), so the rect / poly block rules run here too."""
    spec = cmap.spec
    data = spec["map_data"]
    if spec.get("bundled") and raw_map(spec["name"]) is not None:
        data = raw_map(spec["name"])
    return compile_blob(data, spec.get("roster"), spec.get("start_positions"), spec.get("spawn_regions"), spec.get("scale"),
                        spec.get("wall_radius", WALL_RADIUS))


# ---- tables ----------------------------------------------------------------------------------------------------------------
def ray_table(num_rays: int, ray_length: float = RAY_LENGTH, fov: float = FOV):
    """REF agents/entity.py:182-193: ``angles = np.linspace(0, fov, R, endpoint=False)``; end point = origin + L*cos, L*sin."""
    angles = np.linspace(0, fov, num_rays, endpoint=False)
    return np.ascontiguousarray(ray_length * np.cos(angles)), np.ascontiguousarray(ray_length * np.sin(angles))


_LUTS: Dict[str, np.ndarray] = {}


def reward_tables():
    """The reference's non-terminal rewards are functions of ONE float16 scalar, the minimum distance at which the other team is
    seen (REF agents/cop.py:69-72, thief.py:63-66).  Evaluated as the reference evaluates them -- NumPy float16 SCALAR arithmetic,
    one value at a time -- for each of the 32768 non-negative float16 bit patterns."""
    if not _LUTS:
        import warnings
        cop, thief = np.zeros(32768, np.float32), np.zeros(32768, np.float32)
        with warnings.catch_warnings(), np.errstate(all="ignore"):
            warnings.simplefilter("ignore")
            for bits in range(32768):
                d = np.array([bits], np.uint16).view(np.float16)[0]
                reward = -0.02                             # cop.py:63 time_penalty_per_step
                reward += 1.5 * np.exp(-d / 50.0)          # cop.py:72
                cop[bits] = reward
                thief[bits] = np.tanh((d - 100.0) / 50.0) / 10.0    # thief.py:66
                if bits == 12345:
                    assert type(reward) is np.float16 and type(np.tanh((d - 100.0) / 50.0) / 10.0) is np.float16   # NumPy 2 scalar rules (Q4)
        _LUTS["cop"], _LUTS["thief"] = cop, thief
    return _LUTS["cop"], _LUTS["thief"]
