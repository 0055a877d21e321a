#!/usr/bin/env python3
"""Diagnostic: duration of every wave of ONE tick_kernel launch (build with -DCAT_WAVE_SPREAD into
libcat_sim_spread.so; never the shipped library).  Usage: python tools/wave_spread.py [map] [envs]
Finding it was written for (labyrinth, 4096 envs = four co-resident waves per SIMD): the durations fall on four
plateaus by block index quarter (63k / 72k / 85k / 99k cycles) -- the SIMD arbitrates oldest-first, so each
later "layer" of workgroups only gets the issue slots the earlier ones leave, and the launch ends with the
slowest wave of the last layer (~120k), while a wave alone needs ~65k."""
import ctypes as C, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import torch
from as_cops_and_thieves_amd import _native as nat
nat.LIB_PATH = nat.PKG / "libcat_sim_spread.so"
from as_cops_and_thieves_amd.config import SimConfig
from as_cops_and_thieves_amd.maps import load_preset
from as_cops_and_thieves_amd.sim import CatSim
name = sys.argv[1] if len(sys.argv) > 1 else "labyrinth"; N = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
sim = CatSim(SimConfig(n_envs=N, n_rays=64, seed=0), [load_preset(name).compile()])
sim.reset()
for t in range(100): sim.step_fused(None, t, auto_reset=True)
torch.cuda.synchronize()
import numpy as np
L = nat.lib(); buf = (C.c_ulonglong * (2 * N))()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); sim.step_fused(None, 100, auto_reset=False); e1.record()
torch.cuda.synchronize(); L.cat_debug_spread(buf, N)
print("event-timed tick:", round(1e3 * e0.elapsed_time(e1), 1), "us")
t = np.array(buf, dtype=np.uint64).reshape(N, 2).astype(np.int64)
d = t[:, 1] - t[:, 0]
print(f"{name} N={N}: wave cycles min {d.min()} p10 {np.percentile(d,10):.0f} median {np.median(d):.0f} mean {d.mean():.0f} p90 {np.percentile(d,90):.0f} p99 {np.percentile(d,99):.0f} max {d.max()}")
st = sim.get_state()
ws = st["wall_shape"].cpu().numpy().reshape(N, -1); pa = st["pair_age"].cpu().numpy().reshape(N, -1)
nc = (ws >= 0).sum(1) + (pa >= 0).sum(1)
print("contacts per env: mean", nc.mean(), "max", nc.max())
for c in range(0, nc.max() + 1):
    m = nc == c
    if m.sum() > 10: print(f"  contacts {c}: n={m.sum()} mean dur {d[m].mean():.0f}")
print("corr(dur, contacts) =", np.corrcoef(d, nc)[0, 1])
db = d.reshape(-1, 4)
print("block-level: mean of block means sd", db.mean(1).std(), " within-block sd", db.std(1).mean(), " overall sd", d.std())
# start time relative to earliest (assume one clock domain per ~XCD: use raw start order)
order = np.argsort(t[:, 0]); 
print("corr(dur, env index) =", np.corrcoef(d, np.arange(N))[0, 1])
o = sim.out
od = o["obs_distance"].cpu().numpy().view(np.float16).astype(np.float32).reshape(N, -1)
print("corr(dur, mean obs distance) =", np.corrcoef(d, od.mean(1))[0, 1])
print("mean duration by env-index sixteenth:", [int(d[i * N // 16:(i + 1) * N // 16].mean()) for i in range(16)])
