#!/usr/bin/env python3
"""Diagnostic: would a one-tick launch end earlier if the envs were DEALT to the workgroups by their measured work?

A one-tick launch runs one workgroup per CU (16 env slots each) and ends with its slowest workgroup.  This probe measures every env's work in one
launch of a -DCAT_WAVE_SPREAD build in the UNIT form (front + unit + write-back durations per slot), permutes the envs' state records so that the
workgroups' sums are even (sorted by work, dealt in snake order), and times the launches that follow -- against the same state dealt at random and
as it lay.  The work of an env changes slowly from tick to tick (agents move a few pixels), so what was measured at tick t holds for the next ticks.
The timed sim is a second one of the same library (its default scheduler: the pooled kernel on the light maps), fed the permuted state.
usage: CAT_SPREAD_LIB=build/var/spread.so python tools/balance_probe.py [map] [envs] [cops] [thieves]"""
import ctypes as C, os, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import numpy as np
import torch
from as_cops_and_thieves_amd import _native as nat
nat.LIB_PATH = Path(os.environ.get("CAT_SPREAD_LIB", str(ROOT / "build/var/spread.so"))).resolve()
import bench
name = sys.argv[1] if len(sys.argv) > 1 else "labyrinth"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
nc = int(sys.argv[3]) if len(sys.argv) > 3 else 2
nt = int(sys.argv[4]) if len(sys.argv) > 4 else 1
dev = torch.device("cuda", 0)
os.environ["CAT_POOL"] = "0"
sim_u, cfg, cmap = bench.build_sim(name, nc, nt, N, 64, 0, dev)      # unit form: every slot's fan is its own units, so their durations are the env's work
del os.environ["CAT_POOL"]
sim_p, _, _ = bench.build_sim(name, nc, nt, N, 64, 0, dev)           # the library's own choice: what the bench times
print(f"{name} {nc}v{nt} x{N}: measuring sim {sim_u.one_tick_kernel}, timed sim {sim_p.one_tick_kernel}")
sim_u.reset()
for t in range(600):
    sim_u.step_fused(None, tick=t, auto_reset=True)
torch.cuda.synchronize()
L = nat.lib()
L.cat_debug_spread.argtypes = [C.c_void_p, C.c_int]
L.cat_debug_slot_times.argtypes = [C.c_void_p, C.c_int]
W = 16


def measure_work(sim, tick):
    sim.step_fused(None, tick=tick, auto_reset=True)
    torch.cuda.synchronize()
    sb = (C.c_ulonglong * (16 * N))()
    L.cat_debug_slot_times(sb, N)
    s = np.array(sb, dtype=np.uint64).reshape(N, 16).astype(np.int64) * 0.01
    work = (s[:, 1] - s[:, 0]) + (s[:, 13] - s[:, 12])
    for u in range(5):
        d = s[:, 3 + 2 * u] - s[:, 2 + 2 * u]
        work = work + np.where((s[:, 3 + 2 * u] > 0) & (d > 0) & (d < 1000), d, 0.0)
    return work


def timed(sim, state, tick, launches=20, reps=3):
    sim.set_state(**state)
    out = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for b in range(launches):
            sim.step_fused(None, tick=tick + b, auto_reset=True)
        e1.record(); torch.cuda.synchronize()
        out.append(1e3 * e0.elapsed_time(e1) / launches)
        sim.set_state(**state)              # every repetition starts from the same state
    return out


tick = 600
sim_p.reset()
rng = np.random.default_rng(0)
for rnd in range(3):
    work = measure_work(sim_u, tick); tick += 1
    st = sim_u.get_state()
    nb = N // W
    order = np.argsort(-work)                       # heaviest first
    perm = np.empty(N, dtype=np.int64)              # perm[16 * w + s] = env whose record goes to slot s of workgroup w
    for j, e in enumerate(order):
        r, c = divmod(j, nb)
        w = c if r % 2 == 0 else nb - 1 - c
        perm[W * w + r] = e
    wsum = work.reshape(nb, W).sum(1)
    wsum_b = work[perm].reshape(nb, W).sum(1)
    print(f"round {rnd}: env work us  min {work.min():.1f} median {np.median(work):.1f} p90 {np.percentile(work, 90):.1f} max {work.max():.1f};  workgroup sums / 16: "
          f"as it lies median {np.median(wsum) / W:.2f} max {wsum.max() / W:.2f};  dealt median {np.median(wsum_b) / W:.2f} max {wsum_b.max() / W:.2f}")
    ident = {k: v.clone() for k, v in st.items()}
    pt = torch.as_tensor(perm, device=dev)
    dealt = {k: (v[pt].clone() if v.dim() > 0 and v.shape[0] == N else v.clone()) for k, v in st.items()}
    rp = torch.as_tensor(rng.permutation(N), device=dev)
    shuf = {k: (v[rp].clone() if v.dim() > 0 and v.shape[0] == N else v.clone()) for k, v in st.items()}
    for nm, state in (("as it lies", ident), ("random deal", shuf), ("dealt by work", dealt), ("as it lies", ident), ("dealt by work", dealt)):
        print(f"   {nm:14s} us per launch (3 x 20 launches from the same state): " + "  ".join(f"{x:.2f}" for x in timed(sim_p, state, tick)))
    for t in range(40):                              # move on: another state for the next round
        sim_u.step_fused(None, tick=tick, auto_reset=True); tick += 1
