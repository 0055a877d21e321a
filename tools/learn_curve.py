#!/usr/bin/env python3
"""Diagnostic: does the learner learn?  Cops (trained) against uniformly random thieves on a map; prints the cop win
rate of sampled-action evaluation episodes every few updates.  Usage: python tools/learn_curve.py [map] [envs] [updates] [lr] [max_step_count] [entropy_scale] [norm|raw] [horizon]"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from as_cops_and_thieves_amd import VecCopsEnv, load_preset
from as_cops_and_thieves_amd.selfplay.mappo import MAPPOTrainer, RoleConfig, TrainerConfig
from as_cops_and_thieves_amd.selfplay.self_play import evaluate_agents, mean_reward_per_tick

name = sys.argv[1] if len(sys.argv) > 1 else "squarinth"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
U = int(sys.argv[3]) if len(sys.argv) > 3 else 200
lr = float(sys.argv[4]) if len(sys.argv) > 4 else 1e-4
msc = int(sys.argv[5]) if len(sys.argv) > 5 else 400
ent = float(sys.argv[6]) if len(sys.argv) > 6 else 0.02
norm = (sys.argv[7] == "norm") if len(sys.argv) > 7 else False
horizon = int(sys.argv[8]) if len(sys.argv) > 8 else 16
rc = RoleConfig(random_timesteps=0, learning_starts=0, learning_rate=lr, entropy_loss_scale=ent)
if name == "arena":   # an open 600 x 600 arena with one small block: cops and thieves spawn 50..200 px apart, so the
    import json, tempfile                       # cops' shaping reward (1.5 exp(-d/50) while a thief is in sight) is dense
    from as_cops_and_thieves_amd.maps import Map
    reg = {"x": 150, "y": 150, "w": 300, "h": 300}
    data = {"window": {"w_px": 600, "h_px": 600}, "canvas": {"w": 600, "h": 600},
            "objects": {"blocks": [{"type": "rect", "x": 20, "y": 20, "w": 30, "h": 30}]},
            "agents": [{"type": "cop", "x": 250, "y": 300, "spawn_region": reg}, {"type": "cop", "x": 350, "y": 300, "spawn_region": reg},
                       {"type": "thief", "x": 300, "y": 200, "spawn_region": reg}]}
    f = Path(tempfile.mkdtemp()) / "arena.json"
    f.write_text(json.dumps(data))
    the_map = Map(f)
else:
    the_map = load_preset(name)
env = VecCopsEnv(the_map, num_envs=N, num_rays=64, max_step_count=msc, seed=1)
ev = VecCopsEnv(the_map, num_envs=512, num_rays=64, max_step_count=msc, seed=99)
tc = TrainerConfig(policy_freeze_duration=0, opponent_freeze_duration=0, random_action_roles=("thief",), normalize_inputs=norm, horizon=horizon)
tr = MAPPOTrainer(env, {"cop": rc, "thief": rc}, tc, seed=0)
tr.set_frozen(role="thief", policy=True, value=True)
evr = MAPPOTrainer(ev, {"cop": rc, "thief": rc}, TrainerConfig(horizon=16, graph_rollout=False, graph_update=False, normalize_inputs=norm), seed=1)
t0 = time.time()
for u in range(U + 1):
    if u % max(1, U // 10) == 0:
        evr.load_state_dict(tr.state_dict(), optimizer=False)
        c, t = evaluate_agents(ev, evr, 512, random_roles=("thief",))
        r = mean_reward_per_tick(ev, evr, msc, random_roles=("thief",))
        print(f"update {u:4d} ({u * horizon * N / 1e6:6.1f} M env-steps, {time.time() - t0:5.1f} s): cop win rate {c:.3f} thief {t:.3f}  "
              f"mean cop reward/tick over full evaluation episodes {0.5 * (r['cop_0'] + r['cop_1']):+.4f}  value_loss {tr.read_stats().get('cop_0/value_loss')}", flush=True)
    tr.collect(); tr.update()
