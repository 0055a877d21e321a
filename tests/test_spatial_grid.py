"""The spatial-hash ray grid built at cat_create must be a SUPERSET of the exact BBTree gate: for any
origin and ray, every wall whose bb the thin segment enters (cpBBSegmentQuery finite) is listed for
(cell of the origin, ray index).  Brute-force check on the host copy of the tables (no GPU)."""
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


@pytest.mark.parametrize("name,rays,cell", [("labyrinth", 64, 32), ("agh-map", 90, 32), ("squarinth", 64, 16), ("lbirinth", 90, 48)])
def test_ray_grid_is_superset_of_exact_gate(name, rays, cell):
    cmap = load_preset(name).compile()
    cfg = SimConfig(n_rays=rays)
    L, h, rdx, rdy = _grid(cmap, cfg, cell)
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
            assert set(want) <= set(got), (name, ax, ay, k, sorted(set(want) - set(got)))
            listed += n; exact += len(want)
    assert listed <= 3.0 * max(exact, 1) + 400 * rays * 0.5      # and it is reasonably tight
    print(f"{name}: listed/exact = {listed / max(exact, 1):.2f}, table {L.cat_grid_bytes_host(h) / 1e6:.2f} MB")
    L.cat_grid_free_host(h)


def test_contact_grid_lists_every_wall_within_agent_radius():
    cmap = load_preset("agh-map").compile()
    cfg = SimConfig(n_rays=64)
    L, h, *_ = _grid(cmap, cfg, 32)
    rng = np.random.default_rng(1)
    out = (C.c_int * 256)()
    bb = cmap.shape_bb
    for _ in range(4000):
        x, y = rng.uniform([-50, -50], [1350, 850])
        n = L.cat_grid_lookup_host(h, float(x), float(y), -1, out, 256)
        got = set(out[:n])
        r = cfg.agent_radius
        want = np.nonzero((bb[:, 0] <= x + r) & (x - r <= bb[:, 2]) & (bb[:, 1] <= y + r) & (y - r <= bb[:, 3]))[0]
        assert set(want) <= got
    L.cat_grid_free_host(h)
