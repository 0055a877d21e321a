#!/usr/bin/env python3
"""Diagnostic: can the whole learner pipeline (device env observations -> packing -> stacked conv/LSTM networks ->
HIP-graph PPO update) learn a FUNCTION OF THE RAY OBSERVATIONS quickly?  The env's rewards are replaced by a
contextual-bandit signal: +1 when the agent's action is the impulse that points at its nearest non-empty ray
(quadrant of argmin distance), else 0.  The game's own objective needs tens of millions of env-steps before anything
moves (tools/learn_curve.py); this probe isolates "does PPO on these networks learn" from "is the game hard".
Usage: python tools/learn_probe.py [envs] [updates] [raw|norm] [lr]"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from as_cops_and_thieves_amd import VecCopsEnv, load_preset
from as_cops_and_thieves_amd.selfplay.mappo import MAPPOTrainer, RoleConfig, TrainerConfig
from as_cops_and_thieves_amd.selfplay.probe import NearestRayRewardEnv

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
U = int(sys.argv[2]) if len(sys.argv) > 2 else 150
norm = (sys.argv[3] == "norm") if len(sys.argv) > 3 else False
lr = float(sys.argv[4]) if len(sys.argv) > 4 else 3e-4
env = NearestRayRewardEnv(VecCopsEnv(load_preset("squarinth"), num_envs=N, num_rays=64, max_step_count=400, seed=1))
rc = RoleConfig(random_timesteps=0, learning_starts=0, learning_rate=lr, entropy_loss_scale=0.01)
tr = MAPPOTrainer(env, {"cop": rc, "thief": rc}, TrainerConfig(horizon=16, policy_freeze_duration=0, opponent_freeze_duration=0, normalize_inputs=norm), seed=0)
t0 = time.time()
for u in range(U + 1):
    tr.collect()
    if u % max(1, U // 10) == 0:
        acc = {a: float(rl.buf["rew"][g].mean()) for rl in tr.roles.values() for g, a in enumerate(rl.agents)}
        print(f"update {u:4d} ({time.time() - t0:5.1f} s): fraction of actions pointing at the nearest ray {acc}", flush=True)
    tr.update()
