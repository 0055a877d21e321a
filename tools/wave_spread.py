#!/usr/bin/env python3
"""Diagnostic: timeline of every wave of ONE step_kernel launch (build with -DCAT_WAVE_SPREAD, CAT_SPREAD_LIB=build/var/spread.so;
never the shipped library): the 100 MHz realtime counter (one domain for the whole device) at the wave's start, after the staging
barrier, after its own slot's front is published and at its exit from the scheduler, all relative to the launch's first wave start.
usage: CAT_SPREAD_LIB=build/var/spread.so python tools/wave_spread.py [map] [envs] [cops] [thieves]"""
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
sim, cfg, cmap = bench.build_sim(name, nc, nt, N, 64, 0, torch.device("cuda", 0))
sim.reset()
for t in range(600):
    sim.step_fused(None, tick=t, auto_reset=True)
torch.cuda.synchronize()
L = nat.lib()
L.cat_debug_spread.argtypes = [C.c_void_p, C.c_int]
buf = (C.c_ulonglong * (8 * N))()
burst = int(os.environ.get("CAT_SPREAD_BURST", "20"))   # launches issued back to back; the LAST one's timeline is what is read
tick = 600
for rep in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for b in range(burst):
        sim.step_fused(None, tick=tick, auto_reset=True); tick += 1
    e1.record()
    torch.cuda.synchronize()
    L.cat_debug_spread(buf, N)
    t = np.array(buf, dtype=np.uint64).reshape(N, 8).astype(np.int64)
    rt = (t[:, :4] - t[:, 0].min()) * 0.01   # us
    print(f"--- {name} {nc}v{nt} x{N}, burst {rep} of {burst} launches: {1e3 * e0.elapsed_time(e1) / burst:.1f} us per launch by events; the last launch:")
    for i, nm in enumerate(("wave start", "after the staging barrier", "own front published", "scheduler exit")):
        v = rt[:, i]
        print(f"  {nm:28s} min {v.min():6.2f}  p10 {np.percentile(v, 10):6.2f}  median {np.median(v):6.2f}  p90 {np.percentile(v, 90):6.2f}  p99 {np.percentile(v, 99):6.2f}  max {v.max():6.2f} us")
    for i, nm in ((6, "parameter burst arrived"), (7, "env id + descriptor arrived")):
        v = (t[:, i] - t[:, 0].min()) * 0.01
        print(f"  {nm:28s} min {v.min():6.2f}  p10 {np.percentile(v, 10):6.2f}  median {np.median(v):6.2f}  p90 {np.percentile(v, 90):6.2f}  p99 {np.percentile(v, 99):6.2f}  max {v.max():6.2f} us;  after the wave's own start: median {np.median((t[:, i] - t[:, 0]) * 0.01):.2f}")
    dur = t[:, 5] - t[:, 4]
    print(f"  wave lifetime (shader clock)  min {dur.min()}  median {int(np.median(dur))}  p90 {int(np.percentile(dur, 90))}  max {dur.max()} cycles;  {np.median(dur) / max(np.median(rt[:, 3] - rt[:, 0]), 1e-9) / 1e3:.2f} GHz")
    W = 16
    blk = rt.reshape(-1, W, 4)
    bs, be = blk[:, :, 0].min(1), blk[:, :, 3].max(1)
    print(f"  per workgroup: first start min {bs.min():.2f} median {np.median(bs):.2f} max {bs.max():.2f};  last exit min {be.min():.2f} median {np.median(be):.2f} max {be.max():.2f};  span median {np.median(be - bs):.2f} max {(be - bs).max():.2f} us")
    print(f"  within a workgroup: spread of wave starts median {np.median(blk[:, :, 0].max(1) - bs):.2f} us; spread of exits median {np.median(be - blk[:, :, 3].min(1)):.2f} us")
    # per env slot: front, units, write-back
    if hasattr(L, "cat_debug_slot_times"):
        L.cat_debug_slot_times.argtypes = [C.c_void_p, C.c_int]
        sb = (C.c_ulonglong * (16 * N))()
        L.cat_debug_slot_times(sb, N)
        s = (np.array(sb, dtype=np.uint64).reshape(N, 16).astype(np.int64) - t[:, 0].min()) * 0.01
        A = nc + nt
        def q(v, nm):
            print(f"  {nm:34s} min {v.min():6.2f}  p10 {np.percentile(v, 10):6.2f}  median {np.median(v):6.2f}  mean {v.mean():6.2f}  p90 {np.percentile(v, 90):6.2f}  p99 {np.percentile(v, 99):6.2f}  max {v.max():6.2f}")
        q(s[:, 1] - s[:, 0], "front duration")
        nu = 0
        while nu < 5 and (s[:, 3 + 2 * nu] > 0).all() and (s[:, 3 + 2 * nu] > s[:, 2 + 2 * nu]).all(): nu += 1
        for u in range(nu):
            q(s[:, 2 + 2 * u] - s[:, 1], f"unit {u}: wait publish -> start")
            q(s[:, 3 + 2 * u] - s[:, 2 + 2 * u], f"unit {u}: duration")
        if nu < 5:   # the next unit exists on some slots only (Space.step: not where the front ran it for an auto-reset)
            ok = (s[:, 3 + 2 * nu] > 0) & (s[:, 3 + 2 * nu] >= s[:, 2 + 2 * nu]) & (s[:, 2 + 2 * nu] > s[:, 1])
            if ok.any():
                q((s[:, 2 + 2 * nu] - s[:, 1])[ok], f"unit {nu} (on {ok.mean():.3f} of the slots): wait")
                q((s[:, 3 + 2 * nu] - s[:, 2 + 2 * nu])[ok], f"unit {nu}: duration")
                q((s[:, 3 + 2 * nu])[ok], f"unit {nu}: ends at")
                q((s[:, 12] - s[:, 3 + 2 * nu])[ok], f"unit {nu} end -> write-back start")
        last_end = np.max(s[:, [3 + 2 * u for u in range(nu)]], axis=1)
        q(s[:, 12] - last_end, "last unit end -> write-back start")
        q(s[:, 13] - s[:, 12], "write-back duration")
        q(s[:, 13], "slot finished at")
        wg = s.reshape(-1, W, 16)
        fin = wg[:, :, 13]
        am = fin.argmax(1)
        crit = wg[np.arange(len(am)), am]   # the slot that finishes last in each workgroup
        print("  the LAST slot of each workgroup (medians over workgroups): front start %.2f, published %.2f, " % (np.median(crit[:, 0]), np.median(crit[:, 1])) +
              ", ".join("unit %d %.2f -> %.2f" % (u, np.median(crit[:, 2 + 2 * u]), np.median(crit[:, 3 + 2 * u])) for u in range(nu)) +
              ", write-back %.2f -> %.2f" % (np.median(crit[:, 12]), np.median(crit[:, 13])))
        # the same for the workgroup that ends the launch
        b = fin.max(1).argmax()
        c = wg[b, am[b]]
        print("  the slot that ends the launch: front start %.2f, published %.2f, " % (c[0], c[1]) + ", ".join("unit %d %.2f -> %.2f" % (u, c[2 + 2 * u], c[3 + 2 * u]) for u in range(min(nu + 1, 5))) + ", write-back %.2f -> %.2f" % (c[12], c[13]))
        print("  that workgroup's slots finish at:", np.round(np.sort(fin[b]), 1))
        # total work of a workgroup (sum of its slots' phase durations as measured, i.e. under the contention they ran in) against when it ended
        work = (s[:, 1] - s[:, 0]) + (s[:, 13] - s[:, 12])
        for u in range(nu): work = work + (s[:, 3 + 2 * u] - s[:, 2 + 2 * u])
        wsum = work.reshape(-1, W).sum(1) / W
        bar = np.median(rt[:, 1])
        q(wsum + bar, "workgroup: barrier + work / 16")
        q(fin.max(1), "workgroup: last slot finished at")
        q(fin.max(1) - (wsum + bar), "workgroup: end - (barrier + work/16)")
        print("  correlation(work, end) over workgroups: %.2f" % np.corrcoef(wsum, fin.max(1))[0, 1])
