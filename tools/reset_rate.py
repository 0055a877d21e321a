#!/usr/bin/env python3
"""Diagnostic: how many env slots end their episode per tick on the running batch (random actions), per map -- the launches
of a batch whose episodes end at different ticks each contain a few slots that run Space.step + the spawn sampling in their serial
front (tick_kernel, auto-reset).  Usage: python tools/reset_rate.py [envs] [map ...]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from as_cops_and_thieves_amd.config import SimConfig
from as_cops_and_thieves_amd.maps import load_preset
from as_cops_and_thieves_amd.sim import CatSim

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
names = sys.argv[2:] or ["labyrinth", "lbirinth", "squarinth", "grandbyrinth", "agh-map"]
for name in names:
    sim = CatSim(SimConfig(n_envs=N, n_rays=64, seed=0), [load_preset(name).compile()])
    sim.reset()
    ends, caps, ticks_with = 0, 0, 0
    T0, T = 600, 400
    for t in range(T0 + T):
        sim.step_fused(None, tick=t)
        if t >= T0:
            term = sim.out["terminated"].bool()
            n = int(term.sum())
            ends += n; ticks_with += n > 0
            caps += int((term & (sim.out["truncated"] == 0)).sum())
    print(f"{name:14s} {N} envs: {ends / T:8.2f} episodes end per tick ({caps / T:.2f} by capture), {ticks_with} of {T} ticks have at least one")
