#!/usr/bin/env python3
"""Diagnostic: the native learner over configurations other than the headline one (each 3 rollouts + updates on the GPU):
per-role configs (two learners, G = 2 and G = 1), 1v1, 3v2, a five-map batch.  usage: config_sweep.py <case>"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from as_cops_and_thieves_amd import VecCopsEnv, load_preset
from as_cops_and_thieves_amd.selfplay.mappo import CFG_AGENT_COP, CFG_AGENT_THIEF, MAPPOTrainer, RoleConfig, TrainerConfig
import dataclasses
case = sys.argv[1]
fast = dict(random_timesteps=0, learning_starts=0)
if case == "roles":
    env = VecCopsEnv(load_preset("labyrinth"), 4096, num_rays=64, max_step_count=400)
    cfg = {"cop": dataclasses.replace(CFG_AGENT_COP, **fast), "thief": dataclasses.replace(CFG_AGENT_THIEF, **fast)}
elif case == "1v1":
    env = VecCopsEnv(load_preset("squarinth", 1, 1), 4096, num_rays=64, max_step_count=400)
    cfg = None
elif case == "3v2":
    env = VecCopsEnv(load_preset("grandbyrinth", 3, 2), 4096, num_rays=64, max_step_count=400)
    cfg = None
elif case == "mixed":
    maps = [load_preset(n) for n in ("labyrinth", "squarinth", "lbirinth", "grandbyrinth", "agh-map")]
    env = VecCopsEnv(maps, 4000, num_rays=64, max_step_count=400, slot_map_ids=[i % 5 for i in range(4000)])
    cfg = None
tr = MAPPOTrainer(env, cfg, TrainerConfig(horizon=16, policy_freeze_duration=0, opponent_freeze_duration=0), seed=0)
for _ in range(4):
    tr.collect(); tr.update()
torch.cuda.synchronize()
ok = all(bool(torch.isfinite(rl.fp.master).all()) and bool(rl._graphs) for rl in tr.roles.values())
print(case, "learners", {k: rl.G for k, rl in tr.roles.items()}, "finite+graphs", ok, {k: round(v, 6) for k, v in list(tr.read_stats().items())[:3]})
env.check_errors() if hasattr(env, "check_errors") else None
