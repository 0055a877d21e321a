"""Host logic: map loader / hull / blob, constants, presets — against golden facts extracted from
the reference's own files (tests/golden/*.json, made by tools/make_golden.py)."""
import json
import math
import struct
from pathlib import Path

import numpy as np
import pytest
from scipy.spatial import ConvexHull

from as_cops_and_thieves_amd import constants, maps
from as_cops_and_thieves_amd.config import SimConfig
from as_cops_and_thieves_amd.maps import Map, bundled_map_path, convex_hull, load_preset, parse_block

GOLDEN = Path(__file__).parent / "golden"


def test_constants_equal_reference_pyproject_and_literals():
    g = json.loads((GOLDEN / "reference_constants.json").read_text())
    p = constants.PhysicalParams()
    for k, v in g["physical_params"].items():
        assert getattr(p, k) == v, k
    s = constants.SensorParams()
    assert s.num_rays == g["literals"]["num_rays"]["value"]
    assert s.ray_length == g["literals"]["ray_length"]["value"]
    assert s.ray_radius == g["literals"]["ray_radius"]["value"]
    assert constants.SpaceParams().wall_radius == g["literals"]["wall_radius"]["value"]
    assert {t.name: t.value for t in constants.ObjectType} == g["object_types"]
    cfg = SimConfig()
    assert cfg.dt == 1.0 / g["literals"]["simple_env_time_step_denominator"]["value"]
    assert cfg.max_step_count == g["literals"]["max_step_count"]["value"]


def test_load_physical_params_from_pyproject(tmp_path):
    f = tmp_path / "pyproject.toml"
    f.write_text("[tool.physical-params]\nunit_velocity = 7.5\nmax_speed = 99.0\nunit_mass = 1.0\nunit_size = 5.0\n"
                 "pymunk_cop_category = 42\npymunk_thief_category = 2137\ntermination_radius = 21.0\n")
    p = constants.load_physical_params(f)
    assert (p.unit_velocity, p.max_speed, p.termination_radius) == (7.5, 99.0, 21.0)


def test_bias_coef_is_chipmunk_formula():
    # cpSpaceStep: biasCoef = 1 - pow(collisionBias, dt); collisionBias = pow(1 - 0.1, 60)
    assert abs(SimConfig(dt=1 / 60).bias_coef - 0.1) < 1e-15
    assert abs(SimConfig(dt=1 / 15).bias_coef - 0.3439) < 1e-12


def test_parse_block_rules_match_reference_map_py():
    assert parse_block({"x": 3, "y": 7}) == [(3, 7), (4, 7), (4, 8), (3, 8), (3, 7)]          # w/h default 1
    assert parse_block({"type": "rect", "x": 600, "y": 650, "w": -205, "h": 5})[1] == (395.0, 650.0)  # negative w
    assert parse_block({"x": 1, "y": 2, "w": None, "h": 3}) == [(1, 2), (2, 2), (2, 5), (1, 5), (1, 2)]
    tri = parse_block({"type": "poly", "vs": [{"x": 10, "y": 10}, {"x": 20, "y": 10}, {"x": 15, "y": 5}]})
    assert tri[0] == tri[-1] and len(tri) == 4                                               # ring closed
    with pytest.raises(ValueError):
        parse_block({"type": "rect", "y": 2})
    with pytest.raises(ValueError):
        parse_block({"type": "poly"})
    with pytest.raises(ValueError):
        parse_block({"type": "circle", "x": 1, "y": 1})


def test_convex_hull_is_ccw_from_lexmin_and_matches_qhull():
    rng = np.random.default_rng(0)
    for trial in range(200):
        n = int(rng.integers(3, 30))
        pts = rng.integers(0, 40, (n, 2)).astype(float) if trial % 2 else rng.uniform(0, 100, (n, 2))
        uniq = np.unique(pts, axis=0)
        if len(uniq) < 3 or np.linalg.matrix_rank(uniq[1:] - uniq[0]) < 2:
            continue
        hull = convex_hull([tuple(p) for p in pts] + [tuple(pts[0])])      # closed ring with a duplicate
        want = {tuple(uniq[i]) for i in ConvexHull(uniq).vertices}
        assert set(hull) == want
        assert hull[0] == min(hull)                                        # lexicographic minimum first
        k = len(hull)
        for i in range(k):                                                 # strictly convex, CCW
            a, b, c = hull[i], hull[(i + 1) % k], hull[(i + 2) % k]
            assert (b[0] - a[0]) * (c[1] - b[1]) - (b[1] - a[1]) * (c[0] - b[0]) > 0


def test_map_structural_facts_match_reference_files():
    facts = json.loads((GOLDEN / "map_facts.json").read_text())["maps"]
    for name, f in facts.items():
        if f["has_agents"]:
            m = Map(bundled_map_path(name))
            assert [("cop" if i < m.cops_count else "thief") for i in range(m.cops_count + m.thieves_count)] == \
                sorted(f["roster"])
        else:
            with pytest.raises(KeyError):                                  # labyrinth.json: map.py:75
                Map(bundled_map_path(name))
            m = load_preset(name)
        c = m.compile()
        assert tuple(m.window_dimensions) == tuple(f["window"])
        assert c.n_shapes == f["shapes"]
        assert c.n_planes == f["hull_edges_total"]
        assert int(c.shape_count.max()) == f["hull_edges_max"]
        dropped = sum(1 for ring, cnt in zip(m.blocks, c.shape_count) if cnt < len(set(ring)))
        assert dropped == f["shapes_dropping_vertices"]


def test_plane_records_follow_chipmunk_setverts():
    c = load_preset("squarinth").compile()
    f, n = int(c.shape_first[0]), int(c.shape_count[0])
    pl = c.planes[f:f + n]
    verts = pl[:, 2:4]
    for i in range(n):
        a, b = verts[i - 1], verts[i]
        e = b - a
        want_n = np.array([e[1], -e[0]]) / math.hypot(*e)                  # normalize(rperp(b - a)): outward
        assert np.allclose(pl[i, 0:2], want_n, atol=1e-15)
        assert pl[i, 4] == b[0] * pl[i, 0] + b[1] * pl[i, 1]
        assert pl[i, 5] <= pl[i, 6]                                        # dtMin <= dtMax along the edge
        centre = verts.mean(0)
        assert (centre - b) @ pl[i, 0:2] < 0                               # normal points away from the interior
    assert np.allclose(c.shape_bb[0], [verts[:, 0].min() - 1, verts[:, 1].min() - 1,
                                       verts[:, 0].max() + 1, verts[:, 1].max() + 1])


def test_blob_layout_roundtrip():
    c = load_preset("lbirinth").compile()
    blob = c.to_blob()
    hdr = struct.unpack("<16i", blob[:64])
    assert hdr[0] == maps.BLOB_MAGIC and hdr[2] == c.n_shapes and hdr[3] == c.n_planes and hdr[4] == c.n_agents
    nf = 2 + 4 * c.n_shapes + 8 * c.n_planes + 2 * c.n_agents + 4 * c.regions.shape[0]
    f64 = np.frombuffer(blob, "<f8", nf, 64)
    assert np.array_equal(f64[2:2 + 4 * c.n_shapes].reshape(-1, 4), c.shape_bb)
    i32 = np.frombuffer(blob, "<i4", -1, 64 + 8 * nf)
    assert np.array_equal(i32[:c.n_shapes], c.shape_first)
    assert len(blob) == 64 + 8 * nf + 4 * (2 * c.n_shapes + c.n_agents + 1)


def test_presets_and_roster_overrides():
    m = load_preset("grandbyrinth", 3, 2)
    assert (m.cops_count, m.thieves_count) == (3, 2)
    m = load_preset("squarinth", 1, 1)
    assert (m.cops_count, m.thieves_count) == (1, 1) and "cop_0" in m.agent_spawn_regions
    lab = load_preset("labyrinth")
    assert (lab.cops_count, lab.thieves_count) == (2, 1)
    xs = [x for ring in lab.blocks for x, _ in ring]
    assert max(xs) == pytest.approx(1280.0)                                # canvas 30 -> window 1280
    with pytest.raises(ValueError):
        load_preset("agh-map", 3, 2)
    # squarinth agent order in the file is cop, thief, cop: ids are per-type counters, cops first
    sq = Map(bundled_map_path("squarinth"))
    assert sq.cops_positions == [(350, 350), (300, 300)] and sq.thieves_positions == [(450, 330)]
    assert set(sq.agent_spawn_regions) == {"cop_0", "cop_1", "thief_0"} and len(sq.agent_spawn_regions["thief_0"]) == 4


def _star_junction(tmp_path, arms):
    """`arms` thin walls meeting at one point: an agent circle at the hub overlaps the bounding box of every one of them."""
    import json, math
    hub = (300.0, 300.0)
    blocks = []
    for q in range(arms):
        a = 2 * math.pi * q / arms
        dx, dy = math.cos(a), math.sin(a)
        p0 = (hub[0] + 3 * dx, hub[1] + 3 * dy)
        p1 = (hub[0] + 120 * dx, hub[1] + 120 * dy)
        blocks.append({"type": "poly", "vs": [{"x": p0[0] - dy, "y": p0[1] + dx}, {"x": p1[0] - dy, "y": p1[1] + dx},
                                              {"x": p1[0] + dy, "y": p1[1] - dx}, {"x": p0[0] + dy, "y": p0[1] - dx}]})
    agents = [{"type": "cop", "x": 50, "y": 50}, {"type": "cop", "x": 80, "y": 50}, {"type": "thief", "x": 50, "y": 550}]
    f = tmp_path / f"star{arms}.json"
    f.write_text(json.dumps({"window": {"w_px": 640, "h_px": 640}, "canvas": {"w": 640, "h": 640}, "objects": {"blocks": blocks},
                             "agents": agents}))
    return Map(f).compile()


def test_contact_cache_is_sized_against_the_map_at_create(tmp_path):
    """CAT_WALL_CACHE = 8 cached wall contacts per agent.  The bound held against it -- the most wall bbs one agent circle's bb can
    overlap at once -- is computed from the map at create time (library: cat_map_wall_bb_depth_host / cat_create; oracle:
    cato_create), so a 9-wall star junction is refused up front instead of dropping a contact at run time.  The five maps need
    at most 5."""
    from as_cops_and_thieves_amd import _native as nat
    from as_cops_and_thieves_amd.config import SimConfig
    from oracle.cat_oracle import OracleSim
    import ctypes as C
    lib = C.CDLL(str(nat.LIB_PATH))                     # host-only entry point: no torch / device needed
    lib.cat_map_wall_bb_depth_host.argtypes = [C.c_void_p, C.c_size_t, C.c_double]
    depth = lambda cm: lib.cat_map_wall_bb_depth_host(cm.to_blob(), len(cm.to_blob()), 5.0)
    assert {n: depth(load_preset(n).compile()) for n in ("agh-map", "grandbyrinth", "labyrinth", "lbirinth", "squarinth")} == \
        {"agh-map": 5, "grandbyrinth": 2, "labyrinth": 3, "lbirinth": 2, "squarinth": 2}
    ok, bad = _star_junction(tmp_path, 8), _star_junction(tmp_path, 9)
    assert depth(ok) == 8 and depth(bad) == 9
    OracleSim(SimConfig(n_envs=2, n_rays=16), [ok])
    with pytest.raises(RuntimeError, match="9 walls at once"):
        OracleSim(SimConfig(n_envs=2, n_rays=16), [bad])
