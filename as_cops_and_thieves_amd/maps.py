"""Map loading and compilation to the packed geometry blob the HIP kernels consume.

Host-side mirror of reference ``src/maps/map.py`` (``Map._parse_block`` :35-61,
``_parse_json_map`` :63-117, ``populate_space`` :119-128).  The reference turns every block
into ``pymunk.Poly(space.static_body, ring, radius=1)``; Chipmunk2D then replaces the ring by
its convex hull (``cpConvexHull`` with tolerance 0) and derives one splitting plane per hull
edge.  That construction is restated here ([CHIPMUNK-RECALL]) and the
result is serialised into a flat blob (the input format of the device library).

Differences from the reference that are build-side options (never silently applied):

* ``roster`` / ``start_positions`` / ``spawn_regions`` overrides — ``labyrinth.json`` has no
  ``"agents"`` key (the reference raises ``KeyError`` at ``map.py:75``; so does this loader
  unless an override is given) and BASELINE configs ask for rosters the files do not hold.
* ``scale=(sx, sy)`` — canvas→window scaling the reference declares (``map.py:24-25``) but
  never applies.
"""
from __future__ import annotations

import json
import math
import struct
import sys
from dataclasses import dataclass
from pathlib import Path
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from .constants import DEFAULT_PHYSICAL, DEFAULT_SPACE

BLOB_MAGIC = 0x31544143  # "CAT1" little-endian
BLOB_VERSION = 1
PLANE_STRIDE = 8  # doubles per plane record
_DBL_MIN = sys.float_info.min

Ring = List[Tuple[float, float]]


def _rect_ring(blk: dict) -> Ring:
    """Rect block -> closed 5-point ring; w/h default to 1, negatives allowed (map.py:37-52)."""
    x, y = blk.get("x"), blk.get("y")
    if x is None or y is None:
        raise ValueError("x and y coordinates are required for rectangle blocks.")
    w = blk.get("w") if blk.get("w") is not None else 1
    h = blk.get("h") if blk.get("h") is not None else 1
    return [(x, y), (x + w, y), (x + w, y + h), (x, y + h), (x, y)]


def parse_block(blk: dict) -> Ring:
    """Vertex ring of one JSON block, same accept/reject rules as ``Map._parse_block``."""
    blk_type = blk.get("type", "rect")
    if blk_type == "rect":
        ring = _rect_ring(blk)
    elif blk_type == "poly":
        vs = blk.get("vs")
        if vs is None:
            raise ValueError("Vertices are required for polygon blocks.")
        ring = [(v.get("x"), v.get("y")) for v in vs]
        # shapely's exterior ring is always closed (map.py:61,126)
        if ring and ring[0] != ring[-1]:
            ring = ring + [ring[0]]
    else:
        raise ValueError(f"Unknown block type: {blk_type}")
    return [(float(px), float(py)) for px, py in ring]


def convex_hull(points: Sequence[Tuple[float, float]]) -> Ring:
    """Strict convex hull, counter-clockwise (cross > 0), starting at the lexicographic
    minimum (x, then y) vertex — the cyclic sequence and start ``cpConvexHull(tol=0)``
    produces: duplicates and collinear points dropped."""
    pts = sorted(set((float(x), float(y)) for x, y in points))
    if len(pts) <= 2:
        return pts

    def cross(o, a, b):
        return (a[0] - o[0]) * (b[1] - o[1]) - (a[1] - o[1]) * (b[0] - o[0])

    lower: Ring = []
    for p in pts:
        while len(lower) >= 2 and cross(lower[-2], lower[-1], p) <= 0.0:
            lower.pop()
        lower.append(p)
    upper: Ring = []
    for p in reversed(pts):
        while len(upper) >= 2 and cross(upper[-2], upper[-1], p) <= 0.0:
            upper.pop()
        upper.append(p)
    return lower[:-1] + upper[:-1]  # starts at lexmin, CCW


@dataclass
class CompiledMap:
    """Device-ready geometry of one map (all float64 / int32, C-contiguous)."""

    name: str
    window: Tuple[float, float]
    shape_bb: np.ndarray      # [S,4]  l, b, r, t  (hull bb inflated by the wall radius)
    shape_first: np.ndarray   # [S]    first plane index
    shape_count: np.ndarray   # [S]    plane count
    planes: np.ndarray        # [P,8]  n.x n.y v0.x v0.y dot(v0,n) dtMin dtMax 0
    start_pos: np.ndarray     # [A,2]  cops first, then thieves
    region_off: np.ndarray    # [A+1]
    regions: np.ndarray       # [Rg,4] x y w h
    n_cops: int
    n_thieves: int
    spec: Optional[dict] = None   # the INPUTS this map was compiled from (parsed JSON + Map's override arguments), as plain data

    @property
    def n_agents(self) -> int:
        return self.n_cops + self.n_thieves

    @property
    def n_shapes(self) -> int:
        return int(self.shape_bb.shape[0])

    @property
    def n_planes(self) -> int:
        return int(self.planes.shape[0])

    def to_blob(self) -> bytes:
        hdr = struct.pack(
            "<16i", BLOB_MAGIC, BLOB_VERSION, self.n_shapes, self.n_planes, self.n_agents,
            self.n_cops, self.n_thieves, int(self.regions.shape[0]), 0, 0, 0, 0, 0, 0, 0, 0)
        f64 = np.concatenate([
            np.asarray(self.window, dtype=np.float64),
            self.shape_bb.ravel(), self.planes.ravel(), self.start_pos.ravel(),
            self.regions.ravel()]).astype("<f8")
        i32 = np.concatenate([self.shape_first, self.shape_count, self.region_off]).astype("<i4")
        return hdr + f64.tobytes() + i32.tobytes()


def _planes_for_hull(hull: Ring) -> np.ndarray:
    """Per hull edge i (from v[i-1] to v[i]): outward unit normal and the pre-folded terms of
    ``cpPolyShapeSegmentQuery`` — ``dot(v0,n)``, ``cross(n, v0_prev)``, ``cross(n, v0)``."""
    k = len(hull)
    out = np.zeros((k, PLANE_STRIDE), dtype=np.float64)
    for i in range(k):
        ax, ay = hull[i - 1]
        bx, by = hull[i]
        ex, ey = bx - ax, by - ay
        rx, ry = ey, -ex                        # cpvrperp
        inv = 1.0 / (math.sqrt(rx * rx + ry * ry) + _DBL_MIN)  # cpvnormalize
        nx, ny = rx * inv, ry * inv
        out[i, 0:4] = (nx, ny, bx, by)
        out[i, 4] = bx * nx + by * ny           # cpvdot(v0, n)
        out[i, 5] = nx * ay - ny * ax           # cpvcross(n, v0_{i-1})
        out[i, 6] = nx * by - ny * bx           # cpvcross(n, v0_i)
    return out


class Map:
    """Same constructor and attributes as the reference ``Map`` (``src/maps/map.py:11-33``),
    plus the documented build-side overrides.  ``blocks`` holds vertex rings (the reference
    holds shapely polygons built from the same rings)."""

    def __init__(
        self,
        map_path: str | Path,
        *,
        roster: Optional[Sequence[str]] = None,
        start_positions: Optional[Sequence[Tuple[float, float]]] = None,
        spawn_regions: Optional[Dict[str, List[dict]]] = None,
        scale: Optional[Tuple[float, float]] = None,
    ) -> None:
        self.window_dimensions: Tuple[int, int]
        self.canvas_dimensions: Tuple[int, int]
        self.blocks: List[Ring]
        self.cops_count: int
        self.thieves_count: int
        self.cops_positions: List[Tuple[float, float]]
        self.thieves_positions: List[Tuple[float, float]]
        self.agent_spawn_regions: Dict[str, List[dict]] = {}
        self.unit_size = DEFAULT_PHYSICAL.unit_size
        self.name = Path(map_path).name.split(".")[0]
        self._scale = scale
        self._parse_json_map(str(map_path), roster, start_positions, spawn_regions)
        # what was asked for, kept as data: a checker can rebuild the geometry from these with its own code
        self.spec = {"name": self.name, "map_data": self._map_data, "bundled": Path(map_path).resolve().parent == MAPS_DIR,
                     "roster": None if roster is None else list(roster),
                     "start_positions": None if start_positions is None else [tuple(p) for p in start_positions],
                     "spawn_regions": None if spawn_regions is None else dict(spawn_regions),
                     "scale": None if scale is None else tuple(scale)}

    # -- parsing ------------------------------------------------------------------------
    def _parse_json_map(self, map_path, roster, start_positions, spawn_regions) -> None:
        with open(map_path, "r") as f:
            map_data = json.load(f)
        self._map_data = map_data
        self.window_dimensions = tuple(map_data["window"].values())
        self.canvas_dimensions = tuple(map_data["canvas"].values())
        if "rings" in map_data:  # bundled compact schema (tools/import_reference_maps.py)
            self.blocks = [[(float(r[i]), float(r[i + 1])) for i in range(0, len(r), 2)]
                           for r in map_data["rings"]]
        else:                    # the reference's schema
            self.blocks = [parse_block(b) for b in map_data["objects"]["blocks"]]
        if self._scale is not None:
            sx, sy = self._scale
            self.blocks = [[(x * sx, y * sy) for x, y in ring] for ring in self.blocks]

        if roster is not None:
            if start_positions is None or len(start_positions) != len(roster):
                raise ValueError("roster override needs one start position per agent")
            agents = [{"type": t, "x": p[0], "y": p[1]} for t, p in zip(roster, start_positions)]
        else:
            agents = map_data["agents"]  # KeyError for labyrinth.json, as in the reference

        counts: Dict[str, int] = {}
        for agent in agents:
            agent_type = agent["type"]
            idx = counts.get(agent_type, 0)
            agent_id = f"{agent_type}_{idx}"
            if "spawn_regions" in agent:
                data = agent["spawn_regions"]
                if isinstance(data, list) and all(isinstance(it, dict) for it in data):
                    self.agent_spawn_regions[agent_id] = data
                elif isinstance(data, dict):
                    self.agent_spawn_regions[agent_id] = [data]
                    print(f"Warning: Agent {agent_id} 'spawn_regions' is a single dict. Converting to list.")
                else:
                    print(f"Warning: Agent {agent_id} 'spawn_regions' has invalid format. Ignored. Data: {data}")
            elif "spawn_region" in agent:
                data = agent["spawn_region"]
                if isinstance(data, dict):
                    self.agent_spawn_regions[agent_id] = [data]
                else:
                    print(f"Warning: Agent {agent_id} 'spawn_region' has invalid format. Ignored. Data: {data}")
            counts[agent_type] = idx + 1
        if spawn_regions is not None:
            self.agent_spawn_regions.update(spawn_regions)

        self.cops_positions = [(a["x"], a["y"]) for a in agents if a["type"] == "cop"]
        self.thieves_positions = [(a["x"], a["y"]) for a in agents if a["type"] == "thief"]
        self.cops_count = len(self.cops_positions)
        self.thieves_count = len(self.thieves_positions)

    # -- compilation (the counterpart of populate_space, map.py:119-128) -------------------
    def compile(self, wall_radius: float = DEFAULT_SPACE.wall_radius) -> CompiledMap:
        bbs, firsts, counts, plane_rows = [], [], [], []
        p0 = 0
        for ring in self.blocks:
            hull = convex_hull(ring)
            if len(hull) < 3:
                raise ValueError(f"degenerate block (hull has {len(hull)} vertices): {ring}")
            pl = _planes_for_hull(hull)
            xs = [v[0] for v in hull]
            ys = [v[1] for v in hull]
            bbs.append((min(xs) - wall_radius, min(ys) - wall_radius,
                        max(xs) + wall_radius, max(ys) + wall_radius))
            firsts.append(p0)
            counts.append(len(hull))
            plane_rows.append(pl)
            p0 += len(hull)
        ids = [f"cop_{i}" for i in range(self.cops_count)] + \
              [f"thief_{i}" for i in range(self.thieves_count)]
        starts = list(self.cops_positions) + list(self.thieves_positions)
        off, regs = [0], []
        for aid in ids:
            for r in self.agent_spawn_regions.get(aid, []) or []:
                assert all(k in r for k in "xywh"), \
                    "Invalid spawn region format. Must contain x, y, w, h."
                regs.append((r["x"], r["y"], r["w"], r["h"]))
            off.append(len(regs))
        return CompiledMap(
            name=self.name,
            window=(float(self.window_dimensions[0]), float(self.window_dimensions[1])),
            shape_bb=np.asarray(bbs, dtype=np.float64).reshape(-1, 4),
            shape_first=np.asarray(firsts, dtype=np.int32),
            shape_count=np.asarray(counts, dtype=np.int32),
            planes=np.concatenate(plane_rows, axis=0) if plane_rows else np.zeros((0, PLANE_STRIDE)),
            start_pos=np.asarray(starts, dtype=np.float64).reshape(-1, 2),
            region_off=np.asarray(off, dtype=np.int32),
            regions=np.asarray(regs, dtype=np.float64).reshape(-1, 4),
            n_cops=self.cops_count,
            n_thieves=self.thieves_count,
            spec=dict(self.spec, wall_radius=wall_radius),
        )


# --------------------------------------------------------------------------------------
# Bundled maps.  The geometry/roster DATA of the five reference maps ships under maps_data/ in a
# compact schema ("rings": flat vertex lists after rect expansion; generated by
# tools/import_reference_maps.py).  User-supplied maps in the reference's own schema load
# through the same Map class.  Presets carry the build-side overrides SURVEY.md section 0.2
# lists for the BASELINE configurations.
# --------------------------------------------------------------------------------------
MAPS_DIR = Path(__file__).resolve().parent / "maps_data"

# labyrinth.json: 30x20 canvas units in a 1280x720 window -> scale 1280/30, 720/20.
LABYRINTH_SCALE = (1280.0 / 30.0, 720.0 / 20.0)


def bundled_map_path(name: str) -> Path:
    p = MAPS_DIR / f"{name}.cmap.json"
    if not p.exists():
        raise FileNotFoundError(f"no bundled map named {name!r} in {MAPS_DIR}")
    return p


def load_preset(name: str, n_cops: Optional[int] = None, n_thieves: Optional[int] = None) -> Map:
    """Bundled map with the documented overrides applied (see maps_data/presets.json)."""
    with open(MAPS_DIR / "presets.json") as f:
        presets = json.load(f)
    if name not in presets:
        raise KeyError(f"unknown preset {name!r}; have {sorted(presets)}")
    ps = presets[name]
    kwargs = {}
    if ps.get("scale"):
        kwargs["scale"] = tuple(ps["scale"])
    if n_cops is None and n_thieves is None and "default_roster" not in ps:
        return Map(bundled_map_path(ps["file"]), **kwargs)  # the file's own roster
    cops = ps.get("cops") or []
    thieves = ps.get("thieves") or []
    d_nc, d_nt = ps.get("default_roster", (2, 1))
    nc = d_nc if n_cops is None else n_cops
    nt = d_nt if n_thieves is None else n_thieves
    if nc > len(cops) or nt > len(thieves):
        raise ValueError(f"preset {name!r} has start positions for {len(cops)} cops / "
                         f"{len(thieves)} thieves only")
    kwargs["roster"] = ["cop"] * nc + ["thief"] * nt
    kwargs["start_positions"] = [tuple(p) for p in cops[:nc]] + [tuple(p) for p in thieves[:nt]]
    if ps.get("spawn_regions"):
        kwargs["spawn_regions"] = dict(ps["spawn_regions"])
    return Map(bundled_map_path(ps["file"]), **kwargs)
