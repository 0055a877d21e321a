"""The spatial-hash ray grid built at cat_create must list, for (cell of the origin, ray index), every wall that can matter to
cpSpaceSegmentQueryFirst: a wall whose bb the thin segment enters (cpBBSegmentQuery finite: it is visited) AND whose rounded hull
the ray can touch (segment within wall radius + ray radius of the hull: only then can the visit return a hit).  With
CAT_GRID_HULLS=0 the table keeps every wall of the first kind.  The third rule, occlusion, takes out walls that ARE visitable and
hittable for some origins but can never change the result: that is checked on the results (last test).  Brute-force checks on the
host copy of the tables (no GPU)."""
import ctypes as C

import numpy as np
import pytest

from as_cops_and_thieves_amd import _native as nat
from as_cops_and_thieves_amd import tables
from as_cops_and_thieves_amd.config import C_FIELDS_F64, C_FIELDS_I32, SimConfig
from as_cops_and_thieves_amd.maps import load_preset


def _grid(cmap, cfg, cell):
    L = nat.lib()
    c = nat.CatConfig()
    for n in C_FIELDS_I32 + C_FIELDS_F64:
        setattr(c, n, getattr(cfg, n))
    dx, dy = tables.ray_table(cfg.sensor)
    lut = np.zeros(32768, np.float32)
    t = nat.CatTables(dx.ctypes.data, dy.ctypes.data, lut.ctypes.data, lut.ctypes.data)
    blob = cmap.to_blob()
    h = C.c_void_p()
    assert L.cat_grid_build_host(C.byref(c), C.byref(t), blob, len(blob), float(cell), C.byref(h)) == 0
    return L, h, dx, dy


def _exact_gate(bb, ax, ay, bx, by):
    """[CP cpBBSegmentQuery] != INFINITY, vectorised over walls."""
    dx, dy = bx - ax, by - ay
    with np.errstate(divide="ignore", invalid="ignore"):
        tx1, tx2 = (bb[:, 0] - ax) / dx, (bb[:, 2] - ax) / dx
        ty1, ty2 = (bb[:, 1] - ay) / dy, (bb[:, 3] - ay) / dy
    tmin = np.full(len(bb), -np.inf); tmax = np.full(len(bb), np.inf)
    okx = np.ones(len(bb), bool); oky = np.ones(len(bb), bool)
    if dx == 0.0:
        okx = ~((ax < bb[:, 0]) | (bb[:, 2] < ax))
    else:
        tmin = np.maximum(tmin, np.minimum(tx1, tx2)); tmax = np.minimum(tmax, np.maximum(tx1, tx2))
    if dy == 0.0:
        oky = ~((ay < bb[:, 1]) | (bb[:, 3] < ay))
    else:
        tmin = np.maximum(tmin, np.minimum(ty1, ty2)); tmax = np.minimum(tmax, np.maximum(ty1, ty2))
    return okx & oky & (tmin <= tmax) & (0.0 <= tmax) & (tmin <= 1.0)


def _seg_hull_distance(hull, ax, ay, bx, by):
    """Exact distance between the segment a-b and the convex polygon `hull` ([n, 2], any orientation): 0 if they intersect."""
    a, b = np.array([ax, ay]), np.array([bx, by])

    def inside(q):
        e = np.roll(hull, -1, 0) - hull
        c = e[:, 0] * (q[1] - hull[:, 1]) - e[:, 1] * (q[0] - hull[:, 0])
        return bool(np.all(c >= 0) or np.all(c <= 0))

    def seg_seg(p1, p2, q1, q2):
        def pt_seg(q, s1, s2):
            d = s2 - s1
            t = 0.0 if not d.any() else min(1.0, max(0.0, float(np.dot(q - s1, d) / np.dot(d, d))))
            return float(np.hypot(*(q - (s1 + t * d))))

        def cross(u, v):
            return u[0] * v[1] - u[1] * v[0]
        d1, d2 = p2 - p1, q2 - q1
        den = cross(d1, d2)
        if den != 0.0:
            t, u = cross(q1 - p1, d2) / den, cross(q1 - p1, d1) / den
            if 0.0 <= t <= 1.0 and 0.0 <= u <= 1.0:
                return 0.0
        return min(pt_seg(p1, q1, q2), pt_seg(p2, q1, q2), pt_seg(q1, p1, p2), pt_seg(q2, p1, p2))
    if inside(a) or inside(b):
        return 0.0
    return min(seg_seg(a, b, hull[i], hull[(i + 1) % len(hull)]) for i in range(len(hull)))


@pytest.mark.parametrize("name,rays,cell,hulls", [("labyrinth", 64, 32, 1), ("agh-map", 90, 32, 1), ("agh-map", 64, 8, 1), ("squarinth", 64, 16, 1),
                                                  ("lbirinth", 90, 48, 1), ("agh-map", 64, 16, 0)])
def test_ray_grid_lists_every_wall_that_can_be_visited_and_hit(name, rays, cell, hulls, monkeypatch):
    monkeypatch.setenv("CAT_GRID_HULLS", str(hulls))
    monkeypatch.setenv("CAT_GRID_OCCLUSION", "0")      # rules 1 and 2 of build_grids; rule 3 (occlusion) is checked through the query results below
    cmap = load_preset(name).compile()
    cfg = SimConfig(n_rays=rays)
    L, h, rdx, rdy = _grid(cmap, cfg, cell)
    hull_of = [cmap.planes[f:f + n, 2:4] for f, n in zip(cmap.shape_first, cmap.shape_count)]
    rsum = cfg.wall_radius + cfg.ray_radius
    rng = np.random.default_rng(0)
    lo = cmap.shape_bb[:, :2].min(0) - 450; hi = cmap.shape_bb[:, 2:].max(0) + 450
    out = (C.c_int * 256)()
    listed = exact = 0
    for trial in range(400):
        if trial % 4 == 0:      # origins on cell borders / wall bb corners: the worst cases for a conservative table
            s = rng.integers(cmap.n_shapes)
            ax, ay = cmap.shape_bb[s, rng.choice([0, 2])] + rng.choice([-1e-9, 0, 1e-9]), cmap.shape_bb[s, rng.choice([1, 3])]
            if trial % 8 == 0:
                ax, ay = np.floor(ax / cell) * cell, np.floor(ay / cell) * cell
        else:
            ax, ay = rng.uniform(lo, hi)
        for k in range(rays):
            bx, by = ax + rdx[k], ay + rdy[k]
            want = np.nonzero(_exact_gate(cmap.shape_bb, ax, ay, bx, by))[0]
            n = L.cat_grid_lookup_host(h, float(ax), float(ay), k, out, 256)
            got = list(out[:n])
            assert got == sorted(got), "ids ascending (index order of the sequential visit)"
            if hulls:
                if trial % 16 not in (0, 1, 2, 3, 5):      # the exact hull distance is slow in Python: a third of the origins
                    continue
                want = [s_ for s_ in want if s_ in got or _seg_hull_distance(hull_of[s_], ax, ay, bx, by) <= rsum + 1e-9]
            assert set(want) <= set(got), (name, ax, ay, k, sorted(set(want) - set(got)))
            listed += n; exact += len(want)
    assert listed <= 3.0 * max(exact, 1) + 400 * rays * 0.5      # and it is reasonably tight
    print(f"{name}: listed/exact = {listed / max(exact, 1):.2f}, table {L.cat_grid_bytes_host(h) / 1e6:.2f} MB")
    L.cat_grid_free_host(h)


def test_contact_grid_lists_every_wall_whose_bb_is_within_the_ray_radius():
    """The contact rows serve one rule, [CP cpShapeSegmentQuery]'s "start point within the query radius of the shape -> alpha 0"
    (agent_setup): every wall whose bb comes within the ray radius of the origin must be listed for the origin's cell."""
    cmap = load_preset("agh-map").compile()
    cfg = SimConfig(n_rays=64)
    L, h, *_ = _grid(cmap, cfg, 8)
    rng = np.random.default_rng(1)
    out = (C.c_int * 256)()
    bb = cmap.shape_bb
    longest = 0
    for trial in range(6000):
        if trial % 3 == 0:      # right at a wall's bb
            s_ = rng.integers(cmap.n_shapes)
            x = bb[s_, rng.choice([0, 2])] + rng.choice([-1.0, -1.0 + 1e-9, -1e-9, 0.0, 1e-9, 1.0 - 1e-9, 1.0])
            y = rng.uniform(bb[s_, 1] - 1, bb[s_, 3] + 1)
        else:
            x, y = rng.uniform([-50, -50], [1350, 850])
        n = L.cat_grid_lookup_host(h, float(x), float(y), -1, out, 256)
        got = set(out[:n])
        r = cfg.ray_radius + 1e-6      # the margin agent_setup tests the origin against
        want = np.nonzero((bb[:, 0] - r <= x) & (x <= bb[:, 2] + r) & (bb[:, 1] - r <= y) & (y <= bb[:, 3] + r))[0]
        assert set(want) <= got
        longest = max(longest, n)
    assert longest <= 7      # one packed row: no cell of the densest map needs the CSR continuation
    L.cat_grid_free_host(h)


@pytest.mark.parametrize("name,rays,cell,gate", [("agh-map", 64, 8, 1), ("agh-map", 90, 16, 1), ("labyrinth", 64, 8, 1), ("lbirinth", 64, 8, 1),
                                                 ("agh-map", 64, 8, 0), ("grandbyrinth", 64, 8, 1)])
def test_segment_query_over_the_listed_walls_equals_the_query_over_all_walls(name, rays, cell, gate):
    """The table's three rules (visit, hit, occlusion -- build_grids) may drop walls, never change a result: for any origin and ray
    the oracle's sequential wall query gives the same (wall, alpha bits, point) over the listed walls as over every wall."""
    from oracle.cat_oracle import OracleSim
    cmap = load_preset(name).compile()
    cfg = SimConfig(n_envs=1, n_rays=rays, bbtree_gate=gate)
    L, h, rdx, rdy = _grid(cmap, cfg, cell)
    orc = OracleSim(cfg, [cmap])
    rng = np.random.default_rng(7)
    lo = cmap.shape_bb[:, :2].min(0) - 60; hi = cmap.shape_bb[:, 2:].max(0) + 60
    out = (C.c_int * 256)()
    dropped = listed = 0
    for trial in range(1500):
        if trial % 5 == 0:      # origins right at walls (their bb edges, a hair inside / outside) and on cell borders
            s_ = rng.integers(cmap.n_shapes)
            ax = cmap.shape_bb[s_, rng.choice([0, 2])] + rng.choice([-2.0, -1e-9, 0.0, 1e-9, 2.0])
            ay = rng.uniform(cmap.shape_bb[s_, 1] - 3, cmap.shape_bb[s_, 3] + 3)
            if trial % 10 == 0:
                ax, ay = np.floor(ax / cell) * cell, np.floor(ay / cell) * cell
        else:
            ax, ay = rng.uniform(lo, hi)
        for k in rng.choice(rays, 16, replace=False):
            b = (ax + rdx[k], ay + rdy[k])
            n = L.cat_grid_lookup_host(h, float(ax), float(ay), int(k), out, 256)
            full = orc.segment_query(0, -1, (ax, ay), b, cfg.ray_radius, los=True)
            part = orc.segment_query(0, -1, (ax, ay), b, cfg.ray_radius, los=True, walls=list(out[:n]))
            assert (full[0], np.float64(full[1]).tobytes(), full[2]) == (part[0], np.float64(part[1]).tobytes(), part[2]), \
                (name, ax, ay, int(k), full, part, list(out[:n]))
            listed += n
    L.cat_grid_free_host(h)


def _random_polygon_map(tmp_path, seed, n_blocks, wall_radius=None):
    """Convex blocks of every kind the rules have to get right: slivers, slanted boxes, triangles, blocks that touch along an
    edge, overlap, or sit a hair (less than the ray radius) apart, and a ring of border walls."""
    import json
    from as_cops_and_thieves_amd.maps import Map
    rng = np.random.default_rng(seed)
    blocks = []
    for q in range(n_blocks):
        cx, cy = rng.uniform(60, 580), rng.uniform(60, 420)
        kind = q % 5
        if kind == 0:      # axis-aligned box, sometimes a sliver
            w, h_ = rng.uniform(2, 80), rng.uniform(2, 80)
            blocks.append({"type": "rect", "x": cx, "y": cy, "w": w, "h": h_})
            if q % 10 == 0:      # a neighbour sharing its right edge, and one 0.5 px away from its top edge
                blocks.append({"type": "rect", "x": cx + w, "y": cy, "w": rng.uniform(5, 40), "h": h_})
                blocks.append({"type": "rect", "x": cx, "y": cy + h_ + 0.5, "w": w, "h": rng.uniform(3, 20)})
        else:              # convex polygon with 3 - 7 vertices on an ellipse, rotated
            n = int(rng.integers(3, 8))
            a, b, rot = rng.uniform(3, 60), rng.uniform(3, 60), rng.uniform(0, np.pi)
            ang = np.sort(rng.uniform(0, 2 * np.pi, n))
            vs = [{"x": float(cx + a * np.cos(t) * np.cos(rot) - b * np.sin(t) * np.sin(rot)),
                   "y": float(cy + a * np.cos(t) * np.sin(rot) + b * np.sin(t) * np.cos(rot))} for t in ang]
            blocks.append({"type": "poly", "vs": vs})
    blocks += [{"type": "rect", "x": 10, "y": 10, "w": 5, "h": 460}, {"type": "rect", "x": 10, "y": 470, "w": 620, "h": 5},
               {"type": "rect", "x": 630, "y": 10, "w": 5, "h": 465}, {"type": "rect", "x": 10, "y": 10, "w": 620, "h": 5}]
    agents = [{"type": "cop", "x": 30, "y": 30}, {"type": "cop", "x": 60, "y": 30}, {"type": "thief", "x": 30, "y": 450}]
    f = tmp_path / f"random_{seed}.json"
    f.write_text(json.dumps({"window": {"w_px": 640, "h_px": 480}, "canvas": {"w": 640, "h": 480},
                             "objects": {"blocks": blocks}, "agents": agents}))
    return Map(f).compile() if wall_radius is None else Map(f).compile(wall_radius)


@pytest.mark.parametrize("seed,cell,gate,wall_radius", [(1, 4, 1, None), (2, 8, 1, None), (3, 4, 1, None), (4, 16, 1, None), (5, 4, 0, None),
                                                        (6, 6, 1, None),
                                                        (7, 4, 1, 0.0), (8, 8, 1, 0.0), (9, 4, 0, 0.0)])   # wall_radius 0: a wall's bb IS its hull's
def test_listed_walls_give_the_full_query_result_on_random_polygon_maps(tmp_path, seed, cell, gate, wall_radius):
    """wall_radius = 0 (SimConfig allows it): the occlusion rule's slack across the ray relies on the ROUNDED shape
    reaching past the hull's end vertices, which a zero radius does not give -- the rule then counts a hull only strictly inside."""
    from oracle.cat_oracle import OracleSim
    cmap = _random_polygon_map(tmp_path, seed, 30, wall_radius)
    cfg = SimConfig(n_envs=1, n_rays=64, bbtree_gate=gate, **({} if wall_radius is None else {"wall_radius": wall_radius}))
    L, h, rdx, rdy = _grid(cmap, cfg, cell)
    orc = OracleSim(cfg, [cmap])
    rng = np.random.default_rng(100 + seed)
    out = (C.c_int * 256)()
    shorter = 0
    for trial in range(2500):
        if trial % 4 == 0:      # right at a wall's bb, a hair inside / outside, or within the ray radius of it
            s_ = rng.integers(cmap.n_shapes)
            ax = cmap.shape_bb[s_, rng.choice([0, 2])] + rng.choice([-2.5, -1.0, -1e-9, 0.0, 1e-9, 1.0, 2.5])
            ay = rng.uniform(cmap.shape_bb[s_, 1] - 3, cmap.shape_bb[s_, 3] + 3)
            if trial % 8 == 0:
                ax, ay = np.floor(ax / cell) * cell, np.floor(ay / cell) * cell
        else:
            ax, ay = rng.uniform([0, 0], [640, 480])
        for k in rng.choice(64, 12, replace=False):
            b = (ax + rdx[k], ay + rdy[k])
            n = L.cat_grid_lookup_host(h, float(ax), float(ay), int(k), out, 256)
            full = orc.segment_query(0, -1, (ax, ay), b, cfg.ray_radius, los=True)
            part = orc.segment_query(0, -1, (ax, ay), b, cfg.ray_radius, los=True, walls=list(out[:n]))
            assert (full[0], np.float64(full[1]).tobytes(), full[2]) == (part[0], np.float64(part[1]).tobytes(), part[2]), \
                (seed, ax, ay, int(k), full, part, list(out[:n]))
    L.cat_grid_free_host(h)
