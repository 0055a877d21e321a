#!/usr/bin/env python3
"""Headless counterpart of reference ``src/driver.py``: build the env from a map file, reset once,
then step with random ``Discrete(4)`` actions per live agent (driver.py:65-69).  The reference
wraps this in a pygame keyboard loop (GUI, out of scope); ``--steps`` replaces the key presses.
Like the reference it never resets after the episode ends: once ``env.agents`` is empty the
action dict is empty and the env idles.

    python -m as_cops_and_thieves_amd.driver <mapfile-or-preset> [--steps 500]
"""
from __future__ import annotations

import argparse
from pathlib import Path

from .environments import SimpleEnv
from .maps import Map, load_preset


def configure_argparser():
    parser = argparse.ArgumentParser(description="Cops and Robbers Game (headless, MI355X env core)")
    parser.add_argument("mapfile", type=str, help="map JSON (reference schema) or a bundled preset name")
    parser.add_argument("-r", "--render-mode", type=str, choices=["human", "rgb_array"], default="rgb_array")
    parser.add_argument("-i", "--map-image", type=Path, default=None)
    parser.add_argument("--steps", type=int, default=500)
    parser.add_argument("--seed", type=int, default=None)
    return parser.parse_args()


def main() -> None:
    args = configure_argparser()
    map = Map(args.mapfile) if Path(args.mapfile).exists() else load_preset(args.mapfile)
    env = SimpleEnv(map=map, map_image=args.map_image, render_mode=args.render_mode)
    observations, infos = env.reset(seed=args.seed)
    ended_at = None
    for t in range(args.steps):
        actions = {agent: env.action_space(agent).sample() for agent in env.agents}
        observations, rewards, terminations, _, infos = env.step(actions)
        if ended_at is None and terminations and any(terminations.values()):
            ended_at = t + 1
            print(f"episode ended at step {ended_at}: winner = {next(iter(infos.values()))['winner']}")
    print(f"ran {args.steps} steps; env.step_count = {env.step_count}; live agents = {env.agents}")
    env.close()


if __name__ == "__main__":
    main()
