#!/usr/bin/env python3
"""Diagnostic: rollout collection only (graph replays), for `rocprofv3 --kernel-trace --stats`.  usage: collect_only.py [envs] [rollouts]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from as_cops_and_thieves_amd import VecCopsEnv, load_preset
from as_cops_and_thieves_amd.selfplay.mappo import MAPPOTrainer, TrainerConfig
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
K = int(sys.argv[2]) if len(sys.argv) > 2 else 50
env = VecCopsEnv(load_preset("labyrinth"), num_envs=N, num_rays=64, max_step_count=400)
tr = MAPPOTrainer(env, None, TrainerConfig(horizon=16), seed=0)
for _ in range(3):
    tr.collect()
torch.cuda.synchronize()
for _ in range(K):
    tr.collect()
torch.cuda.synchronize()
