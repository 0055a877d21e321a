#!/usr/bin/env python3
"""Self-play loop on the batched env: the structure of reference ``src/self_play_driver.py:83-117``
+ ``src/training/orchestration.py:100-249`` (train both roles, save one checkpoint per role into
its archive, evaluate against archived opponents, keep PFSP win-rates) with the env tick on the
GPU.  Evaluation follows ``src/utils/eval_pfsp_agents.py:25-49``: greedy actions, an episode is a
cop win iff ``winner == "cop"``.

    python -m as_cops_and_thieves_amd.selfplay.self_play --map squarinth --envs 1024 --iterations 3
"""
from __future__ import annotations

import argparse
import random
from pathlib import Path
from typing import Dict

import torch

from ..environments import VecCopsEnv
from ..maps import load_preset
from . import archive
from .mappo import MAPPOConfig, MAPPOTrainer


@torch.no_grad()
def evaluate(trainer: MAPPOTrainer, episodes: int) -> Dict[str, int]:
    """Greedy rollouts until ``episodes`` episodes have finished; returns win counts per role."""
    env, N = trainer.env, trainer.N
    obs, _ = env.reset()
    starts = torch.ones(N, dtype=torch.bool, device=trainer.device)
    p_state = {a: trainer.policies[a].initial_state(N, trainer.device) for a in trainer.agents}
    wins = {"cop": 0, "thief": 0}
    finished = 0
    while finished < episodes:
        actions = {}
        for a in trainer.agents:
            with trainer._autocast():
                logits, p_state[a] = trainer.policies[a](trainer._policy_in(obs, a).unsqueeze(1), p_state[a], starts.view(N, 1))
            actions[a] = logits[:, 0].float().argmax(-1).to(torch.int32)
        obs, _, terms, _, infos = env.step(actions)
        w = infos["winner"]
        wins["cop"] += int((w == 0).sum())
        wins["thief"] += int((w == 1).sum())
        starts = terms[trainer.agents[0]].clone()
        finished = wins["cop"] + wins["thief"]
    return wins


def run_self_play(map_name: str, num_envs: int, iterations: int, rollouts_per_iteration: int, out_dir: Path,
                  num_rays: int = 64, strategy: str = "pfsp", win_rate_buffer: int = 20, eval_episodes: int = 64,
                  seed: int = 0, device=None) -> Dict[str, float]:
    out_dir = Path(out_dir)
    arch = {"cop": out_dir / "cops", "thief": out_dir / "thieves"}
    env = VecCopsEnv(load_preset(map_name), num_envs, num_rays=num_rays, seed=seed, device=device)
    trainer = MAPPOTrainer(env, MAPPOConfig(), seed=seed)
    rng = random.Random(seed)
    stats: Dict[str, float] = {}
    for it in range(iterations):
        # opponent sampling: the role trained against an archived opponent is chosen alternately
        learner, opponent = ("cop", "thief") if it % 2 == 0 else ("thief", "cop")
        opp_file = archive.sample_policy_from_archive(arch[opponent], opponent, strategy, rng=rng)
        if opp_file is not None:
            trainer.load_role_state_dict(torch.load(opp_file, map_location=trainer.device))
            trainer.cfg.frozen_roles = (opponent,)
        else:
            trainer.cfg.frozen_roles = ()
        stats = trainer.train(rollouts_per_iteration)
        for role in ("cop", "thief"):
            ck = out_dir / f"joint_iter_{it}_{role}.pt"
            out_dir.mkdir(parents=True, exist_ok=True)
            torch.save(trainer.role_state_dict(role), ck)
            archive.add_policy_to_archive(str(ck), arch[role], it, role)
        wins = evaluate(trainer, eval_episodes)
        total = max(1, wins["cop"] + wins["thief"])
        if opp_file is not None:   # PFSP bookkeeping: outcomes of the archived opponent vs the current learner
            name = Path(opp_file).name
            for _ in range(wins[opponent]):
                archive.update_policy_win_rate(arch[opponent], name, True, win_rate_buffer)
            for _ in range(wins[learner]):
                archive.update_policy_win_rate(arch[opponent], name, False, win_rate_buffer)
        stats.update({"iteration": it, "cop_win_rate": wins["cop"] / total})
        print(f"[self-play] iter {it}: learner={learner} opponent={Path(opp_file).name if opp_file else None} "
              f"cop wins {wins['cop']}/{total}")
    env.close()
    return stats


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--map", default="squarinth")
    ap.add_argument("--envs", type=int, default=1024)
    ap.add_argument("--rays", type=int, default=64)
    ap.add_argument("--iterations", type=int, default=3)
    ap.add_argument("--rollouts", type=int, default=8)
    ap.add_argument("--out", type=Path, default=Path("selfplay_out"))
    ap.add_argument("--strategy", default="pfsp", choices=["latest", "random", "pfsp"])
    args = ap.parse_args()
    run_self_play(args.map, args.envs, args.iterations, args.rollouts, args.out, num_rays=args.rays, strategy=args.strategy)


if __name__ == "__main__":
    main()
