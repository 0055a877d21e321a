#!/usr/bin/env python3
"""GPU box: the schedulers of one workload against each other at full batch size over a long run: N envs x TICKS ticks of the bench workload (400-tick
episodes, in-kernel actions and auto-reset) through one-tick launches and through resident launches (T ticks each), under each scheduler choice -- one
process each (the choices are read at cat_create); prints a SHA-256 of the final state and of the last tick's outputs, the device error word, and the rate.
Light maps: CAT_POOL = 1 / 0 (pooled / unit form of the group fan; for 3v2 CAT_POOL=1 also brings the ring in).  "mixed" (the five maps): CAT_SPLIT = 1 / 0
(two parts on two streams / one part on the chunk form).  All eight (four) hashes of a workload must be equal.
usage: tools/pool_soak.py [map] [envs] [ticks] [T] [cops] [thieves]"""
import hashlib, os, subprocess, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
name = sys.argv[1] if len(sys.argv) > 1 else "labyrinth"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
TICKS = int(sys.argv[3]) if len(sys.argv) > 3 else 2048
T = int(sys.argv[4]) if len(sys.argv) > 4 else 256
NC = int(sys.argv[5]) if len(sys.argv) > 5 else 2
NT = int(sys.argv[6]) if len(sys.argv) > 6 else 1
if len(sys.argv) > 7:   # child: mode
    import torch
    import bench
    mode = sys.argv[7]
    sim, cfg, cmap = bench.build_sim(name, NC, NT, N, 64, 0, torch.device("cuda", 0))
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
    print(f"{mode:9s} kernel {sim.one_tick_kernel if mode != 'resident' else sim.rollout_kernel:36s} state {h.hexdigest()[:16]} outputs {ho.hexdigest()[:16]} "
          f"device errors {sim.device_errors()}  {N * TICKS / dt / 1e6:.1f} M env-steps/s")
    sys.exit(0)
print(f"{name} {NC}v{NT} x{N}, {TICKS} ticks ({N * TICKS / 1e6:.1f} M env-steps per run), resident launches of {T} ticks")
switches = [("CAT_SPLIT", "1"), ("CAT_SPLIT", "0")] if name == "mixed" else [("CAT_POOL", "1"), ("CAT_POOL", "0")]
for key, val in switches:
    for mode in ("one-tick", "resident"):
        env = dict(os.environ, **{key: val})
        r = subprocess.run([sys.executable, __file__, name, str(N), str(TICKS), str(T), str(NC), str(NT), mode], env=env, capture_output=True, text=True, timeout=900)
        print(f"{key}={val}", (r.stdout.strip().splitlines() or [r.stderr[-300:]])[-1])
