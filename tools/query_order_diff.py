#!/usr/bin/env python3
"""CPU diagnostic: how many observations depend on the ORDER in which a segment query visits the shapes whose bb it enters
(DESIGN D2).  The oracle in index order (the product's) against two other orders, same seeds, same actions, the same state on both sides every tick,
every ray compared: nearest-bb-first, and the walls by Chipmunk's own descent of its static BBTree as oracle/cat_oracle.c restates it
(cpBBTreeInsert in index order, SubtreeSegmentQuery).  And, whatever tree Chipmunk builds, the BOUND: the queries whose result can depend on the
visiting order at all (the candidate of the smallest alpha is gated out under some order, or the two smallest alphas tie -- order_dependent in the
oracle).  usage: tools/query_order_diff.py [envs] [ticks]"""
import ctypes
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from as_cops_and_thieves_amd.config import SimConfig              # noqa: E402
from as_cops_and_thieves_amd.maps import load_preset              # noqa: E402
from oracle import cat_oracle                                      # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
T = int(sys.argv[2]) if len(sys.argv) > 2 else 200
L = cat_oracle.lib()
L.cato_set_threads(8)
for name in ("labyrinth", "agh-map", "squarinth", "lbirinth", "grandbyrinth"):
    cmap = load_preset(name).compile()
    cfg = SimConfig(n_envs=N, n_rays=64, max_step_count=100, seed=5)
    line = [f"{name}:"]
    for mode, what in ((0, "nearest-bb-first"), (2, "Chipmunk's tree descent (restated)")):
        a, b = cat_oracle.OracleSim(cfg, [cmap]), cat_oracle.OracleSim(cfg, [cmap])
        L.cato_set_index_order(1); a.reset(); b.reset()
        rays = diff_shape = diff_obs = 0
        if mode == 0:
            L.cato_count_order_dependence(1)
        for t in range(T):
            acts = a.random_actions(t)
            before = a.get_state()                       # same state on both sides every tick: only the query order differs
            L.cato_set_index_order(1); oa = {k: v.copy() for k, v in a.step(acts).items()}
            L.cato_set_index_order(mode)
            b.set_state(**before)
            ob = b.step(acts)
            L.cato_set_index_order(1); a.reset(mask=oa["terminated"].copy())
            rays += oa["obs_type"].size
            diff_shape += int((oa["hit_shape"] != ob["hit_shape"]).sum())
            diff_obs += int(((oa["obs_type"] != ob["obs_type"]) | (oa["obs_distance"] != ob["obs_distance"])).sum())
        L.cato_set_index_order(1)
        if mode == 0:
            dep = (ctypes.c_longlong * 4)()
            L.cato_order_dependence(dep)
            L.cato_count_order_dependence(0)
            line.append(f"{rays} rays;  order-dependent at all (any tree): {dep[1] + dep[2]} of {dep[0]} queries = {100.0 * (dep[1] + dep[2]) / max(dep[0], 1):.4f} % "
                        f"(walls {dep[1]}, agents {dep[2]}; {dep[3]} of them ties of the two smallest alphas);")
        line.append(f" index order against {what}: winning shape differs on {diff_shape} ({100.0 * diff_shape / rays:.4f} %), observation (class or f16 distance) on {diff_obs} ({100.0 * diff_obs / rays:.4f} %);")
    print(" ".join(line))
