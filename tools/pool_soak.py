#!/usr/bin/env python3
"""GPU box: the pooled and the unit scheduler of the light-maps fan against each other at full batch size over a long run: N envs x TICKS ticks of
the bench workload (400-tick episodes, in-kernel actions and auto-reset) through (a) one-tick launches of the pooled kernels, (b) resident launches of the
pooled kernels (T ticks each), (c) one-tick launches of the unit form -- one process each (CAT_POOL is read at cat_create); prints a SHA-256 of the final
state and of the last tick's outputs, the device error word, and the rate.  usage: tools/pool_soak.py [map] [envs] [ticks] [T]"""
import hashlib, os, subprocess, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
name = sys.argv[1] if len(sys.argv) > 1 else "labyrinth"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
TICKS = int(sys.argv[3]) if len(sys.argv) > 3 else 2048
T = int(sys.argv[4]) if len(sys.argv) > 4 else 256
if len(sys.argv) > 5:   # child: mode
    import torch
    import bench
    mode = sys.argv[5]
    sim, cfg, cmap = bench.build_sim(name, 2, 1, N, 64, 0, torch.device("cuda", 0))
    sim.reset()
    torch.cuda.synchronize(); t0 = time.time()
    if mode == "resident":
        for r in range(TICKS // T):
            rows = sim.rollout_fused(T, None, tick=r * T, auto_reset=True)
        out = {k: v[-1] for k, v in rows.items()}
    else:
        for t in range(TICKS):
            out = sim.step_fused(None, tick=t, auto_reset=True)
    torch.cuda.synchronize(); dt = time.time() - t0
    h = hashlib.sha256()
    st = sim.get_state()
    for k in sorted(st): h.update(st[k].cpu().numpy().tobytes())
    ho = hashlib.sha256()
    for k in sorted(out):
        if k != "hit_shape": ho.update(out[k].cpu().numpy().tobytes())
    print(f"{mode:9s} kernel {sim.one_tick_kernel if mode != 'resident' else sim.rollout_kernel:22s} state {h.hexdigest()[:16]} outputs {ho.hexdigest()[:16]} "
          f"device errors {sim.device_errors()}  {N * TICKS / dt / 1e6:.1f} M env-steps/s")
    sys.exit(0)
print(f"{name} x{N}, {TICKS} ticks ({N * TICKS / 1e6:.1f} M env-steps per run), resident launches of {T} ticks")
for mode, pool in (("one-tick", "1"), ("resident", "1"), ("one-tick", "0"), ("resident", "0")):
    env = dict(os.environ, CAT_POOL=pool)
    r = subprocess.run([sys.executable, __file__, name, str(N), str(TICKS), str(T), mode], env=env, capture_output=True, text=True, timeout=600)
    print(f"CAT_POOL={pool}", (r.stdout.strip().splitlines() or [r.stderr[-300:]])[-1])
