#!/usr/bin/env python3
"""CPU diagnostic: what the ray fan of a RUNNING batch looks like per 64-ray chunk -- candidate walls per ray from the spatial-hash
table (host copy), exact gate survivors, rays with no candidate, longest list in a chunk (= iterations of the packing loop).
Positions come from the oracle after a burn-in with random actions.  usage: tools/fan_stats.py [map] [envs]"""
import ctypes as C
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from as_cops_and_thieves_amd import _native as nat, tables          # noqa: E402
from as_cops_and_thieves_amd.config import C_FIELDS_F64, C_FIELDS_I32, SimConfig   # noqa: E402
from as_cops_and_thieves_amd.maps import load_preset              # noqa: E402
from oracle import cat_oracle                                      # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "labyrinth"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 256
R = 64
cmap = load_preset(name).compile()
cfg = SimConfig(n_envs=N, n_rays=R, max_step_count=400, seed=0)
cat_oracle.lib().cato_set_threads(8)
sim = cat_oracle.OracleSim(cfg, [cmap])
sim.reset()
for t in range(300):
    out = sim.step(sim.random_actions(t))
    sim.reset(mask=out["terminated"].copy())
pos = sim.get_state()["pos"]
L = nat.lib()
c = nat.CatConfig()
for n in C_FIELDS_I32 + C_FIELDS_F64:
    setattr(c, n, getattr(cfg, n))
dx, dy = tables.ray_table(cfg.sensor)
lut = np.zeros(32768, np.float32)
t = nat.CatTables(dx.ctypes.data, dy.ctypes.data, lut.ctypes.data, lut.ctypes.data)
blob = cmap.to_blob()
h = C.c_void_p()
assert L.cat_grid_build_host(C.byref(c), C.byref(t), blob, len(blob), 16.0, C.byref(h)) == 0
buf = (C.c_int * 256)()
bb = cmap.shape_bb


def tbb(ax, ay, bx, by, b):
    ddx, ddy = bx - ax, by - ay
    tmin, tmax = -np.inf, np.inf
    for a0, d, lo, hi in ((ax, ddx, b[0], b[2]), (ay, ddy, b[1], b[3])):
        if d == 0:
            if a0 < lo or hi < a0:
                return np.inf
        else:
            t1, t2 = (lo - a0) / d, (hi - a0) / d
            tmin, tmax = max(tmin, min(t1, t2)), min(tmax, max(t1, t2))
    return max(tmin, 0.0) if (tmin <= tmax and 0 <= tmax and tmin <= 1) else np.inf


cnts, gates, maxc, zero, pairs = [], [], [], [], []
hit_type = sim.out["obs_type"]
for e in range(min(N, 128)):
    for i in range(3):
        ax, ay = pos[e, i]
        cc, gg = [], []
        for k in range(R):
            n = L.cat_grid_lookup_host(h, float(ax), float(ay), k, buf, 256)
            cc.append(n)
            gg.append(sum(1 for q in range(n) if np.isfinite(tbb(ax, ay, ax + dx[k], ay + dy[k], bb[buf[q]]))))
        cnts.append(np.mean(cc)); gates.append(np.sum(gg)); maxc.append(max(cc)); zero.append(sum(1 for x in cc if x == 0)); pairs.append(sum(cc))
print(f"{name}: per 64-ray chunk over {len(cnts)} chunks: candidates/ray {np.mean(cnts):.2f}, (ray, wall) pairs {np.mean(pairs):.1f} "
      f"(max {max(pairs)}), pairs passing the exact gate {np.mean(gates):.1f}, rays without a candidate {np.mean(zero):.1f}, "
      f"longest list in the chunk {np.mean(maxc):.2f} (max {max(maxc)}); hist of longest list {np.bincount(maxc).tolist()}")
for role, sl in (("cops", slice(0, 2)), ("thief", slice(2, 3))):
    print(role, "rays hitting something:", float((hit_type[:, sl] != 4).mean()))

# nearest-first estimate: stage A = one query per ray with a gate-passing candidate; stage B >= candidates whose t_bb lies below the
# ray's FINAL alpha (the first candidate's own alpha is >= that), minus the first one
A_cnt, B_cnt, all_cnt, chunks = 0, 0, 0, 0
for e in range(min(N, 64)):
    for i in range(3):
        ax, ay = pos[e, i]
        chunks += 1
        for k in range(R):
            n = L.cat_grid_lookup_host(h, float(ax), float(ay), k, buf, 256)
            ts = sorted(tb for tb in (tbb(ax, ay, ax + dx[k], ay + dy[k], bb[buf[q]]) for q in range(n)) if np.isfinite(tb))
            if not ts:
                continue
            sh, alpha, _ = sim.segment_query(e, i, (ax, ay), (ax + dx[k], ay + dy[k]), 1.0)
            a_fin = alpha if sh >= 0 else 1.0
            A_cnt += 1
            B_cnt += max(0, sum(1 for tb in ts if tb < a_fin) - 1)
            all_cnt += len(ts)
print(f"nearest-first: per chunk stage A {A_cnt / chunks:.1f} queries, stage B >= {B_cnt / chunks:.1f}, against {all_cnt / chunks:.1f} gate-passing pairs")
