#!/usr/bin/env python3
"""Diagnostic: per-phase shader-cycle shares of tick_kernel (build with -DCAT_PHASE_TIMING into
libcat_sim_timing.so; never the shipped library).  With -DCAT_EVENT_COUNTS on top the build also counts the shape-query rounds,
the classification sweeps and the exact face / corner tests (and its cycle marks are then distorted by the counting).  Usage: python tools/phase_timing.py [map] [envs] [rays]"""
import ctypes as C, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import torch
from as_cops_and_thieves_amd import _native as nat
import os
nat.LIB_PATH = Path(os.environ.get("CAT_TIMING_LIB", str(nat.PKG / "libcat_sim_timing.so"))).resolve()
from as_cops_and_thieves_amd.config import SimConfig
from as_cops_and_thieves_amd.maps import load_preset
from as_cops_and_thieves_amd.sim import CatSim
name = sys.argv[1] if len(sys.argv) > 1 else "labyrinth"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
R = int(sys.argv[3]) if len(sys.argv) > 3 else 64
cmap = load_preset(name).compile()
sim = CatSim(SimConfig(n_envs=N, n_rays=R, seed=0), [cmap])
sim.reset()
for t in range(100):
    sim.step(sim.random_actions(t)); sim.reset_done()
torch.cuda.synchronize()
L = nat.lib()
buf = (C.c_ulonglong * 24)()
L.cat_debug_phase_cycles(buf, 1)
T = 50
for t in range(100, 100 + T):
    sim.step(sim.random_actions(t)); sim.reset_done()
torch.cuda.synchronize()
L.cat_debug_phase_cycles(buf, 1)
# a mark closes the span since the previous mark of the same wave, so with shared work units the spans also contain
# the claim/scan steps that precede them; "tail" = everything after a wave's last unit (scan, waits, write-back)
names = {3: "write-back: stores of the wave's previous write-back still in flight", 23: "write-back (first of the wave): older memory operations in flight", 21: "write-back: acquire fence", 22: "write-back: slot area + flags", 0: "stage_map", 1: "state -> LDS", 2: "termination+actions", 4: "agent setup (cells, cones)",
         5: "packing (gate)", 20: "unit claim + chunk prologue (row fetch)", 16: "after last unit: scan / wait for open units", 17: "write-back: rewards (LUT)", 18: "write-back: state record", 19: "write-back: shared obs + output stores", 6: "dense items (hull query)", 7: "per-ray resolve", 8: "hit point + f16",
         9: "before physics unit", 10: "physics (rest)", 11: "kernel end", 12: "phys: integrate",
         13: "phys: wall broadphase+narrow", 14: "phys: pairs", 15: "phys: aging"}
tot = sum(buf)
for i in range(24):
    if buf[i]:
        print(f"{str(names.get(i, i)):46s} {buf[i] / T / N:10.0f} cycles/wave  {100.0 * buf[i] / tot:5.1f}%")
print(f"{'total':22s} {tot / T / N:10.0f} cycles/wave")
cnt = (C.c_ulonglong * 8)()
if hasattr(L, "cat_debug_counts"):
    L.cat_debug_counts(cnt)
if any(cnt):
    c = [x / T / N for x in cnt]
    print("per env-step: shape-query rounds %.1f (items %.0f, %.1f lanes per round); classification iterations %.1f (lanes %.0f = %.1f per iteration);"
          % (c[0], c[1], c[1] / max(c[0], 1e-9), c[2], c[3], c[3] / max(c[2], 1e-9)))
    print("              exact face iterations %.1f (tests %.0f = %.1f lanes each, %.2f per item); exact corner iterations %.1f (tests %.0f = %.1f lanes each, %.2f per item)"
          % (c[4], c[5], c[5] / max(c[4], 1e-9), c[5] / max(c[1], 1e-9), c[6], c[7], c[7] / max(c[6], 1e-9), c[7] / max(c[1], 1e-9)))
