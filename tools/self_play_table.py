#!/usr/bin/env python3
"""GPU: the reference's two-sided self-play protocol (src/self_play_driver.py, TrainingConfig: PFSP opponents from the archives) run
for many iterations with the reference's own agent settings -- CFG_AGENT (lr 1e-4, entropy 0.02, 4 epochs x 4 minibatches, KL 0.015),
RAW inputs, the freeze / random-action schedule of CFG_TRAINER -- and a table of how each newly trained role fares against the
archived opponents it is evaluated on.  The only departures: `--timesteps` ticks per iteration on `--envs` parallel envs (the
reference: 100 000 ticks of ONE env), and `--episodes` evaluation episodes per opponent instead of 5, so that a win rate has a
standard error of a few per cent.  Episodes are capped at 2000 ticks, what the reference's driver passes (self_play_driver.py:34).
--per-role: CFG_AGENT_COP / CFG_AGENT_THIEF (mappo_config.py:19-39, "if cops still struggle") instead of CFG_AGENT for everyone.
usage: tools/self_play_table.py [--iterations 30] [--envs 512] [--timesteps 30000] [--episodes 30] [--per-role] [--max-step-count 2000]"""
import argparse
import json
import re
import sys
import tempfile
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from as_cops_and_thieves_amd.selfplay.self_play import TrainingConfig, run_self_play   # noqa: E402
from as_cops_and_thieves_amd.selfplay.mappo import CFG_AGENT, CFG_AGENT_COP, CFG_AGENT_THIEF   # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--map", default="squarinth")
ap.add_argument("--iterations", type=int, default=30)
ap.add_argument("--envs", type=int, default=512)
ap.add_argument("--timesteps", type=int, default=30_000)
ap.add_argument("--episodes", type=int, default=30)
ap.add_argument("--max-step-count", type=int, default=2000)
ap.add_argument("--per-role", action="store_true")
args = ap.parse_args()
role_cfg = {"cop": CFG_AGENT_COP, "thief": CFG_AGENT_THIEF} if args.per_role else {"cop": CFG_AGENT, "thief": CFG_AGENT}
cfg_name = "CFG_AGENT_COP / CFG_AGENT_THIEF" if args.per_role else "CFG_AGENT"
out = Path(tempfile.mkdtemp(prefix="selfplay_"))
tc = TrainingConfig(training_timesteps_per_role_training=args.timesteps, n_trial_episodes=args.episodes)
lines = []
t0 = time.time()


def log(msg):
    lines.append(msg)
    if "iteration" in msg:
        print(f"{msg}   [{time.time() - t0:.0f} s]", flush=True)


res = run_self_play(args.map, args.envs, out, iterations=args.iterations, training=tc, num_rays=64, log=log, role_cfg=role_cfg,
                    max_step_count=args.max_step_count)
print(f"# {args.map}: {args.iterations} iterations x {args.timesteps} ticks x {args.envs} envs = "
      f"{args.iterations * args.timesteps * args.envs / 1e6:.0f} M env-steps in {time.time() - t0:.0f} s; {cfg_name}, raw inputs, PFSP; "
      f"max_step_count {args.max_step_count}; {args.episodes} evaluation episodes per archived opponent")
print("# iteration | new cops vs archived thieves: opponents beaten / evaluated | new thieves vs archived cops: beaten / evaluated")
for h in res["iterations"]:
    ev = h["evaluations"]
    c = [not won for won in ev["cop"].values()]
    t = [not won for won in ev["thief"].values()]
    print(f"{h['iteration']:9d} | {sum(c):2d} / {len(c):2d} | {sum(t):2d} / {len(t):2d}")
# per-episode rates as the evaluation logged them: "<role> vs <file>: cop 0.xx thief 0.yy"
rates = {"cop": {}, "thief": {}}
it = -1
for m in lines:
    if "iteration" in m and "saved" in m:
        it = int(m.split("iteration ")[1].split(":")[0])
    else:
        mm = re.search(r"\]\s+(cop|thief) vs \S+: cop ([0-9.]+) thief ([0-9.]+)", m)
        if mm:
            rates[mm.group(1)].setdefault(it + 1, []).append(float(mm.group(2) if mm.group(1) == "cop" else mm.group(3)))
print("# mean episode win rate of the newly trained role over the opponents it met (cop rate for cops, thief rate for thieves)")
for k in sorted(set(rates["cop"]) | set(rates["thief"])):
    c, t = rates["cop"].get(k, []), rates["thief"].get(k, [])
    print(f"{k:9d} | cops {sum(c) / max(len(c), 1):.2f} over {len(c)} opponents | thieves {sum(t) / max(len(t), 1):.2f} over {len(t)} opponents")
print("# the evaluation log, line by line")
for m in lines:
    if " vs " in m:
        print(m)
for role, d in (("cop", "cops"), ("thief", "thieves")):
    wr = json.loads((out / d / "win_rates.json").read_text())
    print(f"# {d}/win_rates.json (archived {role} policies, as opponents): " + ", ".join(f"{k.split('_')[-1].split('.')[0]}: {v.get('wins', 0)}/{v.get('games', 0)}" for k, v in sorted(wr.items(), key=lambda kv: int(kv[0].split('_')[-1].split('.')[0]))) + "  (evaluations the archived policy WON as the opponent / evaluations)")
