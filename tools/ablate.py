#!/usr/bin/env python3
"""Diagnostic: time tick_kernel with subsets of the outputs disabled (NULL output pointers)."""
import sys, ctypes as C
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import torch
from as_cops_and_thieves_amd import _native as nat
from as_cops_and_thieves_amd.config import SimConfig
from as_cops_and_thieves_amd.maps import load_preset
from as_cops_and_thieves_amd.sim import CatSim
name = sys.argv[1] if len(sys.argv) > 1 else "labyrinth"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
R = int(sys.argv[3]) if len(sys.argv) > 3 else 64
cmap = load_preset(name).compile()
def run(disable, label, steps=300):
    sim = CatSim(SimConfig(n_envs=N, n_rays=R, seed=0), [cmap])
    sim.reset()
    for k in disable:
        setattr(sim._out_struct, k, None)
    acts = [sim.random_actions(t) for t in range(64)]
    for t in range(50):
        sim.step(acts[t % 64]); sim.reset_done()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    tot = 0.0
    for t in range(steps):
        e0.record(); sim.step(acts[t % 64]); e1.record(); sim.reset_done()
        torch.cuda.synchronize(); tot += e0.elapsed_time(e1)
    print(f"{label:40s} tick {1e3*tot/steps:8.1f} us")
    sim.close()
run([], "all outputs")
run(["obs_distance", "obs_type"], "no obs stores")
run(["shared_distance", "shared_type", "team_positions"], "no shared stores")
run(list(nat.OUT_FIELDS), "no output stores at all")
