#!/usr/bin/env python3
"""Diagnostic: the same 4096 envs stepped as 1, 2 or 4 sub-batches on separate HIP streams (one handle each)."""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import torch
from as_cops_and_thieves_amd.config import SimConfig
from as_cops_and_thieves_amd.maps import load_preset
from as_cops_and_thieves_amd.sim import CatSim
N = 4096
cmap = load_preset("labyrinth").compile()
for parts in (1, 2, 4):
    n = N // parts
    streams = [torch.cuda.Stream() for _ in range(parts)]
    sims = []
    for i in range(parts):
        with torch.cuda.stream(streams[i]):
            s = CatSim(SimConfig(n_envs=n, n_rays=64, seed=0, max_step_count=400, env_id_offset=i * n), [cmap]); s.reset(); sims.append(s)
    torch.cuda.synchronize()
    def run(steps, t0):
        for k in range(steps):
            for i in range(parts):
                with torch.cuda.stream(streams[i]):
                    sims[i].step_fused(None, tick=t0 + k, auto_reset=True)
    run(100, 0); torch.cuda.synchronize()
    t = time.perf_counter(); run(1000, 100); torch.cuda.synchronize(); el = time.perf_counter() - t
    print(f"{parts} stream(s) x {n} envs: {1e6 * el / 1000:.1f} us per step of {N} envs -> {N * 1000 / el / 1e6:.1f} M env-steps/s", flush=True)
    for s in sims: s.close()
