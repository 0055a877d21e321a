#!/usr/bin/env python3
"""Diagnostic: env-steps/s of the MAPPO trainer on the device env -- rollout collection (env tick +
the stacked policies and critics per tick, one HIP graph) and the PPO update (HIP-graph minibatch steps) -- beside the
bare env rate of bench.py.  Usage: python tools/train_throughput.py [envs] [rollouts] [map] [horizon] [cops] [thieves]"""
import sys, time, faulthandler
import os
faulthandler.dump_traceback_later(int(os.environ.get("CAT_WATCHDOG_S", "100")), exit=True)    # a hang shows where
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from as_cops_and_thieves_amd import VecCopsEnv, load_preset
from as_cops_and_thieves_amd.selfplay.mappo import MAPPOTrainer, TrainerConfig

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
K = int(sys.argv[2]) if len(sys.argv) > 2 else 10
name = sys.argv[3] if len(sys.argv) > 3 else "labyrinth"
H = int(sys.argv[4]) if len(sys.argv) > 4 else 16     # ticks per rollout (BPTT window stays 16)
roster = (int(sys.argv[5]), int(sys.argv[6])) if len(sys.argv) > 6 else (None, None)
env = VecCopsEnv(load_preset(name, *roster), num_envs=N, num_rays=int(os.environ.get("CAT_RAYS", "64")), max_step_count=400)
tr = MAPPOTrainer(env, None, TrainerConfig(horizon=H), seed=0)      # CFG_AGENT for both roles, as the reference's driver
for _ in range(3):                                         # eager warm-up, then the graph captures
    tr.collect(); tr.update()
    torch.cuda.synchronize(); print('warm-up pass done', flush=True)
torch.cuda.synchronize()
tc = tu = 0.0
for _ in range(K):
    t0 = time.perf_counter(); tr.collect(); torch.cuda.synchronize(); t1 = time.perf_counter()
    tr.update(); torch.cuda.synchronize(); t2 = time.perf_counter()
    tc += t1 - t0; tu += t2 - t1
steps = K * tr.tcfg.horizon * N
graphs = {r: bool(rl._graphs) for r, rl in tr.roles.items()}
print(f"{N} envs, horizon {tr.tcfg.horizon}, {name}: collect {steps / tc / 1e6:.2f} M env-steps/s ({1e3 * tc / K:.2f} ms), "
      f"update {steps / tu / 1e6:.2f} M env-steps/s ({1e3 * tu / K:.2f} ms), end to end {steps / (tc + tu) / 1e6:.2f} M env-steps/s; "
      f"update graphs {graphs}, stats {tr.read_stats()}")
env.check_errors()                                       # device-side error flags of the env core (bad actions, dropped contacts)
assert all(v == v and abs(v) < 1e6 for v in tr.read_stats().values()), "non-finite training statistics"
