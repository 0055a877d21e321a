#!/usr/bin/env python3
"""Diagnostic: which aten ops the PPO minibatch step spends its GPU time in (torch.profiler over eager steps).
Usage: python tools/profile_update_step.py [envs] [horizon]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from torch.profiler import profile, ProfilerActivity
from as_cops_and_thieves_amd import VecCopsEnv, load_preset
from as_cops_and_thieves_amd.selfplay.mappo import MAPPOTrainer, TrainerConfig

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
H = int(sys.argv[2]) if len(sys.argv) > 2 else 16
env = VecCopsEnv(load_preset("labyrinth"), num_envs=N, num_rays=64, max_step_count=400)
tr = MAPPOTrainer(env, None, TrainerConfig(horizon=H, graph_rollout=False, graph_update=False), seed=0)
for _ in range(2):
    tr.collect(); tr.update()
tr.collect()
torch.cuda.synchronize()
rl = next(iter(tr.roles.values()))
rl.start = tr._start_buf
rl.idx.copy_(torch.randperm(rl.N, device=rl.device)[:rl.B])
for _ in range(2):
    rl._step_forward_backward(); rl._step_apply()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    for _ in range(3):
        rl._step_forward_backward(); rl._step_apply()
    torch.cuda.synchronize()
rows = [e for e in prof.key_averages() if e.key.startswith("aten::") or e.key.startswith("_")]
rows.sort(key=lambda e: -e.count)
print("aten / custom ops by call count over 3 steps:")
for e in rows[:45]:
    print(f"{e.count:5d} x  {e.key:45s} cpu {e.self_cpu_time_total / 1e3:8.2f} ms   device {e.self_device_time_total / 1e3:8.2f} ms")
print(prof.key_averages().table(sort_by="self_cuda_time_total", row_limit=30, max_name_column_width=60))
