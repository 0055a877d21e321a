#!/usr/bin/env python3
"""GPU box: long bit-exact rollouts against the oracle on every bundled map (and three random polygon maps), outputs and the whole
f64 state compared every 10 ticks and after every reset -- rare paths (grazing rays, arbiter ageing, respawns all over the map)
accumulate.  The oracle visits every wall; the kernels only the walls the candidate table lists.
usage: python tools/soak_parity.py [ticks] [envs]"""
import sys, time, json, tempfile
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import numpy as np
from as_cops_and_thieves_amd.config import SimConfig
from as_cops_and_thieves_amd.maps import Map, load_preset
from tests.test_gpu_parity import _run
from tests.test_spatial_grid import _random_polygon_map

ticks = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
envs = int(sys.argv[2]) if len(sys.argv) > 2 else 128
cases = [(n, load_preset(n).compile(), 64) for n in ("labyrinth", "agh-map", "lbirinth", "squarinth", "grandbyrinth")]
cases.append(("agh-map, 90 rays", load_preset("agh-map").compile(), 90))
tmp = Path(tempfile.mkdtemp())
for seed in (21, 22, 23):
    _random_polygon_map(tmp, seed, 14)
    f = tmp / f"random_{seed}.json"
    data = json.loads(f.read_text())
    for a in data["agents"]:
        a["spawn_region"] = {"x": 20, "y": 20, "w": 600, "h": 440}
    f.write_text(json.dumps(data))
    cases.append((f"random polygons (seed {seed})", Map(f).compile(), 64))
for name, m, rays in cases:
    cfg = SimConfig(n_envs=envs, n_rays=rays, max_step_count=90, seed=41)
    t0 = time.time()
    st = _run(cfg, [m], np.zeros(envs, np.int32), ticks=ticks, rng=np.random.default_rng(3), check_every=10, auto_reset=True)
    print(f"{name:28s} {envs} envs x {ticks} ticks, {rays} rays: bit-exact; {st['done']} episodes ended ({st['captured']} captures), "
          f"{st['contacts']} cached contacts seen at the checks   [{time.time() - t0:.0f} s]", flush=True)
