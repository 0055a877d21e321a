#!/usr/bin/env python3
"""Resident rollout (cat_rollout_fused, T ticks per launch) against one launch per tick (cat_step_fused) on the bench's workloads.
usage: tools/rollout_bench.py [T ...]   (CAT_RB_WORKLOADS=lab,agh,3v2,mixed,inside,r90 selects)"""
import os, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import bench

W = {"lab": dict(map="labyrinth", cops=2, thieves=1, envs=4096), "agh": dict(map="agh-map", cops=2, thieves=1, envs=4096),
     "3v2": dict(map="grandbyrinth", cops=3, thieves=2, envs=8192), "mixed": dict(map="mixed", cops=2, thieves=1, envs=16384),
     "inside": dict(map="labyrinth-inside", cops=2, thieves=1, envs=4096), "r90": dict(map="labyrinth", cops=2, thieves=1, envs=4096, rays=90),
     "lab32k": dict(map="labyrinth", cops=2, thieves=1, envs=32768)}
Ts = [int(a) for a in sys.argv[1:]] or [16, 64]
names = os.environ.get("CAT_RB_WORKLOADS", "lab,agh,3v2,mixed").split(",")
dev = torch.device("cuda", 0)
for name in names:
    w = W[name]
    sim, cfg, cmap = bench.build_sim(w["map"], w["cops"], w["thieves"], w["envs"], w.get("rays", 64), 0, dev)
    sim.reset()
    burn = int(os.environ.get("CAT_RB_BURN", "600"))
    for t in range(burn):
        sim.step_fused(None, tick=t, auto_reset=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    K = 200
    e0.record()
    for k in range(K):
        sim.step_fused(None, tick=burn + k, auto_reset=True)
    e1.record(); torch.cuda.synchronize()
    one = e0.elapsed_time(e1) / K * 1e3
    line = f"{name:7s} envs {cfg.n_envs:6d}  one launch per tick {one:7.2f} us/tick ({cfg.n_envs / one:6.1f} M env-steps/s)"
    tick = burn + K
    for T in Ts:
        sim.rollout_fused(T, None, tick=tick, auto_reset=True); tick += T
        torch.cuda.synchronize()
        reps = max(2, 512 // T)
        e0.record()
        for _ in range(reps):
            sim.rollout_fused(T, None, tick=tick, auto_reset=True); tick += T
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / (reps * T) * 1e3
        line += f" | T={T}: {us:7.2f} us/tick ({cfg.n_envs / us:6.1f} M)"
    print(line, flush=True)
    assert sim.device_errors() == 0
    sim.close()
