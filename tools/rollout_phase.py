#!/usr/bin/env python3
"""Diagnostic: per-phase shader-cycle shares of rollout_kernel (a -DCAT_PHASE_TIMING build, CAT_TIMING_LIB=...; never the shipped
library).  usage: CAT_TIMING_LIB=build/var/timing.so python tools/rollout_phase.py [map] [envs] [T]   (T = 0: step_kernel, one tick per launch)"""
import ctypes as C, os, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import torch
from as_cops_and_thieves_amd import _native as nat
nat.LIB_PATH = Path(os.environ.get("CAT_TIMING_LIB", str(ROOT / "build/var/timing.so"))).resolve()
import bench
name = sys.argv[1] if len(sys.argv) > 1 else "labyrinth"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
T = int(sys.argv[3]) if len(sys.argv) > 3 else 64
sim, cfg, cmap = bench.build_sim(name, 2, 1, N, 64, 0, torch.device("cuda", 0))
sim.reset()
for t in range(600):
    sim.step_fused(None, tick=t, auto_reset=True)
torch.cuda.synchronize()
L = nat.lib()
buf = (C.c_ulonglong * 24)()
L.cat_debug_phase_cycles(buf, 1)
reps = 4
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
if T == 0:   # T = 0: the one-tick path (step_kernel through cat_step_fused), 64 launches
    T = 64; reps = 1
    for r in range(T):
        sim.step_fused(None, tick=600 + r, auto_reset=True)
else:
    for r in range(reps):
        sim.rollout_fused(T, None, tick=600 + r * T, auto_reset=True)
e1.record(); torch.cuda.synchronize()
L.cat_debug_phase_cycles(buf, 1)
names = {0: "prologue: stage_map", 1: "prologue: state -> LDS", 2: "front: termination + actions", 4: "front: agent setup", 3: "front: publish",
         21: "IDLE: no open unit (sleep)", 22: "scan + claim", 20: "fan: prologue (row fetch, sorting)", 5: "fan: packing (gate)", 6: "fan: shape queries",
         7: "fan: per-ray walk", 8: "fan: hit point + f16", 9: "before physics", 10: "physics (rest)", 12: "phys: integrate", 13: "phys: walls",
         14: "phys: pairs", 15: "phys: aging", 23: "unit done (release + count)", 16: "write-back: acquire", 17: "write-back: rewards (LUT)",
         18: "write-back: counters / state record", 19: "write-back: shared obs + output stores", 11: "kernel end"}
tot = sum(buf)
steps = reps * T * N
for i in sorted(names, key=lambda k: list(names).index(k)):
    if buf[i]:
        print(f"{names[i]:46s} {buf[i] / steps:10.0f} cycles/env-step  {100.0 * buf[i] / tot:5.1f}%")
print(f"{'total':46s} {tot / steps:10.0f} cycles/env-step;  {e0.elapsed_time(e1) * 1e3 / (reps * T):.2f} us/tick in this (instrumented) build")
cnt = (C.c_ulonglong * 8)()
if hasattr(L, "cat_debug_counts"):
    L.cat_debug_counts(cnt)
if any(cnt):   # a -DCAT_EVENT_COUNTS build (its cycle marks are distorted by the counting)
    c = [x / steps for x in cnt]
    print("per env-step: shape-query rounds %.2f (items %.1f, %.1f lanes per round); classification iterations %.2f (lanes %.1f = %.1f per iteration);"
          % (c[0], c[1], c[1] / max(c[0], 1e-9), c[2], c[3], c[3] / max(c[2], 1e-9)))
    print("              exact face iterations %.2f (tests %.1f = %.1f lanes each, %.2f per item); exact corner iterations %.2f (tests %.1f = %.1f lanes each, %.2f per item)"
          % (c[4], c[5], c[5] / max(c[4], 1e-9), c[5] / max(c[1], 1e-9), c[6], c[7], c[7] / max(c[6], 1e-9), c[7] / max(c[1], 1e-9)))
