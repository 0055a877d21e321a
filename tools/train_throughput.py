#!/usr/bin/env python3
"""Diagnostic: env-steps/s of the MAPPO trainer on the device env -- rollout collection (env
tick + 3 LSTM policies + 3 LSTM critics per tick) and the PPO update -- beside the bare env rate of bench.py.
Usage: python tools/train_throughput.py [envs] [rollouts]"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from as_cops_and_thieves_amd import VecCopsEnv, load_preset
from as_cops_and_thieves_amd.selfplay.mappo import MAPPOConfig, MAPPOTrainer

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
K = int(sys.argv[2]) if len(sys.argv) > 2 else 6
env = VecCopsEnv(load_preset("labyrinth"), num_envs=N, num_rays=64, max_step_count=400)
tr = MAPPOTrainer(env, MAPPOConfig(), seed=0)
tr.update(tr.collect())   # warm-up (allocator, autotuning)
torch.cuda.synchronize()
tc = tu = 0.0
for _ in range(K):
    t0 = time.perf_counter(); ro = tr.collect(); torch.cuda.synchronize(); t1 = time.perf_counter()
    tr.update(ro); torch.cuda.synchronize(); t2 = time.perf_counter()
    tc += t1 - t0; tu += t2 - t1
steps = K * tr.cfg.horizon * N
print(f"{N} envs, horizon {tr.cfg.horizon}: collect {steps / tc / 1e6:.2f} M env-steps/s, update {steps / tu / 1e6:.2f} M env-steps/s, "
      f"end to end {steps / (tc + tu) / 1e6:.2f} M env-steps/s")
