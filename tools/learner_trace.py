#!/usr/bin/env python3
"""For `rocprofv3 --kernel-trace --stats`: the learner leg of bench.py at one setting -- K rounds of rollout collection (env ticks + the six stacked
networks per tick, one replayed HIP graph) [+ the PPO update of CFG_AGENT] on labyrinth 2v1 x4096, after 3 warm-up rounds (graph capture).
usage: learner_trace.py collect|full [horizon] [rays] [rounds]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from as_cops_and_thieves_amd import VecCopsEnv, load_preset
from as_cops_and_thieves_amd.selfplay.mappo import MAPPOTrainer, TrainerConfig
mode = sys.argv[1] if len(sys.argv) > 1 else "full"
H = int(sys.argv[2]) if len(sys.argv) > 2 else 128
R = int(sys.argv[3]) if len(sys.argv) > 3 else 64
K = int(sys.argv[4]) if len(sys.argv) > 4 else 5
env = VecCopsEnv(load_preset("labyrinth"), num_envs=4096, num_rays=R, max_step_count=400)
tr = MAPPOTrainer(env, None, TrainerConfig(horizon=H), seed=0)
for _ in range(3):
    tr.collect()
    if mode == "full":
        tr.update()
torch.cuda.synchronize()
for _ in range(K):
    tr.collect()
    if mode == "full":
        tr.update()
torch.cuda.synchronize()
print(f"{mode}: {K} rounds of horizon {H}, {R} rays, after 3 warm-up rounds")
