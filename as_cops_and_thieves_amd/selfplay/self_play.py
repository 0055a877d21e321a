#!/usr/bin/env python3
"""Self-play on the batched env: the reference's ``src/self_play_driver.py:83-117`` loop over
``training/orchestration.py:100-249`` (``_orchestrate_simultaneous_training_iteration``) and
``utils/agent_learning_utils.py:172-380`` (``train_simultaneously_and_evaluate`` / ``evaluate_agent``), with
``TrainingConfig`` of ``src/configs/training_config.py:3-12``.

Per iteration, as in the reference:

1. both roles continue from the latest checkpoint of their archive (model weights only -- the reference builds a
   fresh ``MAPPO`` and copies policy + value weights in, ``orchestration.py:121-211``; Adam starts fresh);
2. one ``trainer.train()``: all roles train simultaneously under the timestep schedule (``mappo.MAPPOTrainer.train``);
3. the newly trained cops are evaluated against up to 5 DISTINCT archived thief policies sampled by the configured
   strategy (PFSP), ``n_trial_episodes`` episodes each; per opponent ONE outcome -- "did the archived opponent win
   more episodes than the learner" -- is booked into that opponent's entry of the thief archive's ``win_rates.json``;
   then the same for the thieves against archived cops (``agent_learning_utils.py:233-380``);
4. the joint checkpoint ``joint_iter_{i}_full_agent.pt`` (every model, the optimiser state, the trainer position) is
   saved and copied into BOTH archives as ``{role}_iter_{i}.pt`` (``orchestration.py:225-245``).

Conscious deviation (quirk Q16): the reference's ``evaluate_agent`` loads each sampled opponent INTO the
agent it has just trained (``eval_agent = learned_agent`` is an alias, ``agent_learning_utils.py:253``), so the
checkpoint saved in step 4 holds the last sampled ARCHIVED policies of both roles instead of the trained ones.  Here
the opponents are loaded into a separate evaluation copy and the trained weights are what gets saved.

    python -m as_cops_and_thieves_amd.selfplay.self_play --map squarinth --envs 1024 --iterations 3 --timesteps 2000

**N GPUs of one node** (BASELINE configs[2]: 32768 envs sharded 8x, gradient all-reduce over xGMI):

    python -m as_cops_and_thieves_amd.selfplay.self_play --gpus 8 --map agh-map --envs 32768 ...

The command starts N fresh rank processes through ``torch.distributed.run`` as a CHILD process (never an exec; the parent has not
touched the GPU).  Rank r simulates the global env ids ``shard_envs(envs, r, N)`` (no data-path collective: SURVEY 8e), the ranks
open one RCCL group, and every optimiser step all-reduces ONE buffer -- the flat gradients | the KL statistics of all stacked
networks (``mappo.RoleLearner.minibatch_step``) -- so the replicas stay bit-identical and take the same early-stop decisions.
Rank 0 alone evaluates, writes checkpoints, archives and ``win_rates.json``; the others wait at a barrier and read the files.
"""
from __future__ import annotations

import argparse
import dataclasses
import inspect
import os
import random
import sys
from pathlib import Path
from typing import Dict, Optional, Tuple

import torch

from ..environments import VecCopsEnv
from ..maps import load_preset
from . import archive
from .mappo import CFG_AGENT, MAPPOTrainer, RoleConfig, TrainerConfig, _sample


@dataclasses.dataclass
class TrainingConfig:
    """``src/configs/training_config.py:3-12``."""
    num_self_play_iterations: int = 40
    training_timesteps_per_role_training: int = 100_000
    archive_save_interval: int = 1
    policy_sample_strategy: str = "pfsp"
    win_rate_buffer_size: int = 20
    n_trial_episodes: int = 5
    cop_role_prefix: str = "cop"
    thief_role_prefix: str = "thief"
    num_opponents_to_evaluate: int = 5        # evaluate_agent(num_additional_opponents_to_evaluate=5)


@torch.no_grad()
def evaluate_agents(env, runner: MAPPOTrainer, n_episodes: int, random_roles: Tuple[str, ...] = ()) -> Tuple[float, float]:
    """``src/utils/eval_pfsp_agents.py:7-59``: ``n_episodes`` episodes with every model frozen, actions sampled from
    the policies (skrl ``policy.act``), an episode ends at its first termination and is a win of ``infos["winner"]``.
    Returns (cop wins / n, thief wins / n).  The batched form plays the episodes in parallel, one per env slot
    (``env.num_envs >= n_episodes``; the first episode of the first ``n_episodes`` slots counts).  ``random_roles``:
    these roles act uniformly at random instead (a fixed yardstick opponent; not part of the reference protocol)."""
    N = env.num_envs
    assert N >= n_episodes and N == runner.N
    obs, _ = env.reset()
    starts = torch.ones(N, dtype=torch.bool, device=runner.device)
    state = {r: rl.policy.initial_state(N) for r, rl in runner.roles.items()}
    winner = torch.full((N,), -1, dtype=torch.int8, device=runner.device)
    open_ = torch.ones(N, dtype=torch.bool, device=runner.device)
    open_[n_episodes:] = False
    actions = torch.zeros(N, len(runner.agents), dtype=torch.int32, device=runner.device)
    for _ in range(env.max_step_count + 2):
        keep = (~starts).view(1, N)
        for r, rl in runner.roles.items():
            pin = torch.stack([runner_pack(obs[a]) for a in rl.agents])
            if runner.tcfg.normalize_inputs:
                pin = pin * runner._pin_scale
            logits, state[r] = rl.policy.forward(pin.unsqueeze(1), state[r], keep)
            act = _sample(torch.log_softmax(logits[:, 0].float(), dim=-1))
            rnd = [ar in random_roles for ar in rl.agent_roles]
            if any(rnd):                                # a uniformly random opponent (not part of the reference protocol)
                rows = torch.tensor(rnd, device=runner.device).view(rl.G, 1)
                act = torch.where(rows, torch.randint(0, 4, (rl.G, N), device=runner.device), act)
            actions.index_copy_(1, rl.index_t, act.t().to(torch.int32))
        obs, _, terms, _, infos = env.step(actions)
        done = terms[runner.agents[0]]
        first = open_ & done
        winner = torch.where(first, infos["winner"].to(torch.int8), winner)
        open_ = open_ & ~done
        starts = done.clone()
        if not bool(open_.any()):                       # one host sync per tick, on the evaluation path only
            break
    w = winner[:n_episodes]
    return float((w == 0).sum()) / n_episodes, float((w == 1).sum()) / n_episodes


@torch.no_grad()
def mean_reward_per_tick(env, runner: MAPPOTrainer, ticks: int, random_roles: Tuple[str, ...] = ()) -> Dict[str, float]:
    """Diagnostic (not part of the reference protocol): reset ``env``, act for ``ticks`` ticks with sampled actions
    (``random_roles`` uniformly at random) and return every agent's mean reward per tick -- the quantity PPO maximises,
    measured from the same starting conditions whenever it is called."""
    N = env.num_envs
    assert N == runner.N
    obs, _ = env.reset()
    starts = torch.ones(N, dtype=torch.bool, device=runner.device)
    state = {r: rl.policy.initial_state(N) for r, rl in runner.roles.items()}
    actions = torch.zeros(N, len(runner.agents), dtype=torch.int32, device=runner.device)
    total = {a: torch.zeros((), device=runner.device) for a in runner.agents}
    for _ in range(ticks):
        keep = (~starts).view(1, N)
        for r, rl in runner.roles.items():
            pin = torch.stack([runner_pack(obs[a]) for a in rl.agents])
            if runner.tcfg.normalize_inputs:
                pin = pin * runner._pin_scale
            logits, state[r] = rl.policy.forward(pin.unsqueeze(1), state[r], keep)
            act = _sample(torch.log_softmax(logits[:, 0].float(), dim=-1))
            rnd = [ar in random_roles for ar in rl.agent_roles]
            if any(rnd):
                act = torch.where(torch.tensor(rnd, device=runner.device).view(rl.G, 1), torch.randint(0, 4, (rl.G, N), device=runner.device), act)
            actions.index_copy_(1, rl.index_t, act.t().to(torch.int32))
        obs, rewards, terms, _, _ = env.step(actions)
        for a in runner.agents:
            total[a] += rewards[a].float().mean()
        starts = terms[runner.agents[0]].clone()
    return {a: float(v) / ticks for a, v in total.items()}


def runner_pack(obs_agent):
    from .. import packing
    return packing.pack_policy_input(obs_agent)


def evaluate_agent(eval_env, evaluator: MAPPOTrainer, learned: MAPPOTrainer, learned_role: str, opponent_role: str,
                   opponent_archive: Path, tc: TrainingConfig, rng: random.Random, log=print) -> Dict[str, bool]:
    """``agent_learning_utils.py:233-380``: the newly trained ``learned_role`` against up to
    ``tc.num_opponents_to_evaluate`` distinct archived ``opponent_role`` policies.  Returns {opponent file: opponent won}."""
    results: Dict[str, bool] = {}
    evaluator.load_state_dict(learned.state_dict(), roles=[learned_role], optimizer=False)
    seen = set()
    for i in range(tc.num_opponents_to_evaluate):
        path = None
        for strategy in (tc.policy_sample_strategy, "random"):          # 20 tries for a new one, then 20 uniformly
            for _ in range(20):
                cand = archive.sample_policy_from_archive(opponent_archive, opponent_role, strategy, rng=rng)
                if cand is None:
                    break
                if Path(cand).name not in seen:
                    path = cand
                    break
            if path is not None:
                break
        if path is None:
            log(f"[self-play] no new distinct {opponent_role} opponent for evaluation round {i + 1}/{tc.num_opponents_to_evaluate}")
            break
        name = Path(path).name
        seen.add(name)
        evaluator.load_state_dict(torch.load(path, map_location=evaluator.device, weights_only=True), roles=[opponent_role],
                                  optimizer=False)                       # copy_role_models: policy + value weights
        cop_rate, thief_rate = evaluate_agents(eval_env, evaluator, tc.n_trial_episodes)
        opponent_won = (thief_rate > cop_rate) if learned_role == tc.cop_role_prefix else (cop_rate > thief_rate)
        archive.update_policy_win_rate(opponent_archive, name, opponent_won, tc.win_rate_buffer_size)
        results[name] = opponent_won
        log(f"[self-play]   {learned_role} vs {name}: cop {cop_rate:.2f} thief {thief_rate:.2f} -> opponent {'won' if opponent_won else 'lost'}")
    return results


def run_self_play(map_name: str, num_envs: int, out_dir: Path, iterations: Optional[int] = None,
                  training: Optional[TrainingConfig] = None, trainer_cfg: Optional[TrainerConfig] = None,
                  role_cfg: Optional[Dict[str, RoleConfig]] = None, num_rays: int = 64, n_cops: Optional[int] = None,
                  n_thieves: Optional[int] = None, max_step_count: int = 2000, eval_envs: Optional[int] = None,
                  seed: int = 0, device=None, resume: bool = True, log=print, env_factory=None) -> Dict[str, object]:
    """The self-play loop.  ``resume``: continue after the highest iteration found in the archives ("latest").
    ``max_step_count``: 2000, what the reference's driver passes (``self_play_driver.py:34``).
    ``env_factory(num_envs, seed[, env_id_offset])``: build the envs some other way (the CPU tests pass a stand-in env with the
    same surface).

    With an initialised ``torch.distributed`` group of W > 1 ranks this is ONE data-parallel job: ``num_envs`` is the TOTAL,
    rank r simulates ``shard_envs(num_envs, r, W)``; the trainer all-reduces its gradient | KL buffer every optimiser step (all
    ranks hold identical parameters at all times); rank 0 alone evaluates and writes files, the others wait and read them."""
    import torch.distributed as dist
    from ..sharding import shard_envs
    multi = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
    rank, world = (dist.get_rank(), dist.get_world_size()) if multi else (0, 1)
    chief = rank == 0

    def sync(ok: bool = True, what: str = ""):
        """File hand-over between rank 0 and the others: a MIN all-reduce of an ok flag instead of a bare barrier, so that a failure
        of rank 0 while it evaluates or writes reaches every rank at once (they raise too) instead of leaving them in a collective
        until the backend's watchdog fires.  The group is opened with a generous timeout (``init_ranks``): rank 0's evaluation
        (2000-tick episodes against several archived opponents) runs while the others wait here."""
        if multi:
            flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device="cuda" if dist.get_backend() == "nccl" else "cpu")
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) == 0 and ok:
                raise RuntimeError(f"self-play: rank 0 failed {what or 'in its rank-0-only section'}; this rank ({rank}) stops with it")
    tc = training or TrainingConfig()
    iterations = tc.num_self_play_iterations if iterations is None else iterations
    # TrainerConfig's 128-tick rollouts: with 16-tick rollouts the cops' win rate against random thieves stays at its untrained
    # 10 % for 262 M env-steps, with 128 it rises (profiles/r02_learning_curves.txt); the reference collects 4096 ticks per update
    trainer_cfg = trainer_cfg or TrainerConfig(timesteps=tc.training_timesteps_per_role_training)
    out_dir = Path(out_dir)
    arch = {tc.cop_role_prefix: out_dir / "cops", tc.thief_role_prefix: out_dir / "thieves"}
    if chief:
        for p in arch.values():
            p.mkdir(parents=True, exist_ok=True)
    sync()
    n_eval = eval_envs or tc.n_trial_episodes
    if env_factory is None:
        preset = load_preset(map_name, n_cops, n_thieves)
        env_factory = lambda n, s, off=0: VecCopsEnv(preset, n, num_rays=num_rays, max_step_count=max_step_count, seed=s, device=device,
                                                     env_id_offset=off)
    n_local, offset = shard_envs(num_envs, rank, world)
    if multi:
        # checked on EVERY rank from the same numbers, so that all of them refuse together (a rank that raised alone would leave the others in the
        # trainer's first all-reduce): the smallest shard must still fill every role's minibatches
        smallest = min(shard_envs(num_envs, r, world)[0] for r in range(world))
        windows = max(1, trainer_cfg.horizon // min(trainer_cfg.bptt, trainer_cfg.horizon))
        need = max(c.mini_batches for c in (role_cfg or {"cop": CFG_AGENT, "thief": CFG_AGENT}).values())
        if smallest * windows < need:
            raise ValueError(f"{num_envs} envs over {world} ranks leave a rank {smallest} env(s) = {smallest * windows} training sequences per update, "
                             f"fewer than the {need} minibatches of the role configuration")
    takes_offset = len(inspect.signature(env_factory).parameters) >= 3
    if multi and not takes_offset:
        raise TypeError("a data-parallel run needs env_factory(num_envs, seed, env_id_offset): the ranks must simulate different envs")
    env = env_factory(n_local, seed, offset) if takes_offset else env_factory(n_local, seed)
    eval_env = env_factory(n_eval, seed + 7919)          # every rank builds one (the evaluator's shapes); only rank 0 plays on it
    role_cfg = role_cfg or {"cop": CFG_AGENT, "thief": CFG_AGENT}        # self_play_driver.py passes CFG_AGENT
    trainer = MAPPOTrainer(env, role_cfg, trainer_cfg, seed=seed)
    evaluator = MAPPOTrainer(eval_env, role_cfg, dataclasses.replace(trainer_cfg, graph_rollout=False, graph_update=False),
                             seed=seed + 1)
    rng = random.Random(seed)
    start = 0
    if resume:
        latest = [archive.get_latest_policy_from_archive(arch[r], r) for r in arch]
        its = [int(Path(p).stem.split("_")[-1]) for p in latest if p]
        start = max(its) + 1 if its else 0
    history = []
    for it in range(start, start + iterations):
        # ---- 1. continue from the latest archived checkpoint of each role (orchestration.py:146-211)
        for role in arch:
            ck = archive.sample_policy_from_archive(arch[role], role, "latest")
            if ck:
                trainer.load_state_dict(torch.load(ck, map_location=trainer.device, weights_only=True), roles=[role], optimizer=False)
        trainer.reset_optimizers()
        trainer.reset_episodes()
        # ---- 2. simultaneous training (agent_learning_utils.py:172-197)
        stats = trainer.train(trainer_cfg.timesteps)
        cop, thief = tc.cop_role_prefix, tc.thief_role_prefix
        ev = {cop: {}, thief: {}}
        chief_error = None
        if chief:
            try:
                # ---- 3. evaluation against archived opponents (:199-228)
                ev = {cop: evaluate_agent(eval_env, evaluator, trainer, cop, thief, arch[thief], tc, rng, log),
                      thief: evaluate_agent(eval_env, evaluator, trainer, thief, cop, arch[cop], tc, rng, log)}
                # ---- 4. joint checkpoint into both archives (orchestration.py:225-245)
                ck = out_dir / f"joint_iter_{it}_full_agent.pt"
                torch.save(trainer.state_dict(), ck)
                if it % tc.archive_save_interval == 0 or it == start + iterations - 1:
                    for role in arch:
                        archive.add_policy_to_archive(str(ck), arch[role], it, role)
                log(f"[self-play] iteration {it}: saved {ck.name}; evaluated {len(ev[cop])} thief and {len(ev[thief])} cop opponents"
                    + (f"; {world} ranks x {n_local} envs" if multi else ""))
            except Exception as exc:   # noqa: BLE001 -- handed to the other ranks below, then re-raised here
                if not multi:
                    raise
                chief_error = exc
        sync(ok=chief_error is None, what=f"while evaluating / saving iteration {it}")   # the other ranks read this iteration's archive entries in step 1 of the next
        if chief_error is not None:
            raise chief_error
        history.append({"iteration": it, "evaluations": ev, "stats": stats})
    digest = trainer.param_digest()
    env.close()
    eval_env.close()
    return {"iterations": history, "param_digest": digest, "archives": {r: str(p) for r, p in arch.items()}, "rank": rank, "world": world,
            "envs_local": n_local, "env_id_offset": offset}


def launch_ranks(n: int, argv) -> int:
    """``--gpus N`` typed as a plain command: start N fresh rank processes of this module, one per GPU, through
    ``torch.distributed.run`` as a CHILD process (the bench.py pattern: never an exec, and this parent has not initialised the GPU)."""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK", "MASTER_PORT"):
        env.pop(k, None)
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), "-m", "as_cops_and_thieves_amd.selfplay.self_play", *argv]
    return subprocess.run(cmd, env=env).returncode


def group_timeout():
    """Collective timeout of the job's process group: the non-chief ranks wait in ``sync`` while rank 0 evaluates (2000-tick episodes
    against ``num_opponents_to_evaluate`` archived opponents per role) and writes checkpoints -- well beyond the backends' default of
    10 minutes on a slow disk or a long evaluation.  ``CAT_SELFPLAY_TIMEOUT_S`` (default two hours)."""
    import datetime
    return datetime.timedelta(seconds=float(os.environ.get("CAT_SELFPLAY_TIMEOUT_S", "7200")))


def init_ranks(gpus: int) -> str:
    """Inside a rank process (WORLD_SIZE set by the launcher): one GPU per rank and the RCCL group (``nccl`` IS RCCL on ROCm).
    ``CAT_SELFPLAY_REHEARSE=1``: the flow on a box with fewer GPUs than ranks -- ranks share devices, gloo carries the all-reduce
    (RCCL refuses two ranks on one device).  Returns the backend in use."""
    import torch.distributed as dist
    world, local = int(os.environ["WORLD_SIZE"]), int(os.environ.get("LOCAL_RANK", "0"))
    if gpus != world:
        raise SystemExit(f"--gpus {gpus} inside a {world}-rank group: run the plain command, it starts its own ranks")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if os.environ.get("CAT_SELFPLAY_REHEARSE") == "1":
        torch.cuda.set_device(local % max(1, torch.cuda.device_count()))
        dist.init_process_group("gloo", timeout=group_timeout())
        return "gloo (CAT_SELFPLAY_REHEARSE=1: ranks share devices)"
    torch.cuda.set_device(local)
    dist.init_process_group("nccl", device_id=torch.device("cuda", local), timeout=group_timeout())
    probe = torch.ones(1, device=torch.device("cuda", local))
    dist.all_reduce(probe)
    if int(probe.item()) != world:
        raise RuntimeError(f"RCCL all-reduce of ones over {world} ranks returned {probe.item()}")
    return "nccl (RCCL)"


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--map", default="squarinth")
    ap.add_argument("--envs", type=int, default=1024)
    ap.add_argument("--rays", type=int, default=64)
    ap.add_argument("--iterations", type=int, default=3)
    ap.add_argument("--timesteps", type=int, default=100_000, help="env ticks per iteration (reference: 100 000)")
    ap.add_argument("--out", type=Path, default=Path("lstm_policy_archive_self_play_new"))
    ap.add_argument("--strategy", default="pfsp", choices=["latest", "random", "pfsp"])
    ap.add_argument("--eval-envs", type=int, default=None)
    ap.add_argument("--max-step-count", type=int, default=2000, help="episode cap (self_play_driver.py:34 passes 2000)")
    ap.add_argument("--horizon", type=int, default=None, help="rollout ticks per update (TrainerConfig default: 128)")
    ap.add_argument("--cops", type=int, default=None)
    ap.add_argument("--thieves", type=int, default=None)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--per-role-configs", action="store_true", help="CFG_AGENT_COP / CFG_AGENT_THIEF (mappo_config.py:19-39) instead of CFG_AGENT for both")
    ap.add_argument("--random-timesteps", type=int, default=None, help="override of the role configs' random_timesteps (mappo_config.py:9: 10000)")
    ap.add_argument("--learning-starts", type=int, default=None, help="override of learning_starts (mappo_config.py:10: 15000)")
    ap.add_argument("--freeze-duration", type=int, default=None, help="override of CFG_TRAINER's policy / opponent freeze durations (15000)")
    ap.add_argument("--non-recurrent", action="store_true", help="the reference's non-recurrent Policy / Value pair (policy_net.py, value_net.py; "
                    "model_utils.py:45-77) instead of the LSTM pair its drivers use")
    ap.add_argument("--gpus", type=int, default=1, help="data-parallel ranks, one per GPU: --envs is the TOTAL, sharded across them")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))          # plain command: start the ranks ourselves
    backend = None
    if args.gpus > 1:
        backend = init_ranks(args.gpus)
    tc = TrainingConfig(policy_sample_strategy=args.strategy, training_timesteps_per_role_training=args.timesteps)
    over = {"horizon": args.horizon} if args.horizon else {}
    if args.non_recurrent:
        over["recurrent"] = False
    if args.freeze_duration is not None:
        over.update(policy_freeze_duration=args.freeze_duration, opponent_freeze_duration=args.freeze_duration)
    tcfg = TrainerConfig(timesteps=args.timesteps, **over)
    from .mappo import CFG_AGENT_COP, CFG_AGENT_THIEF
    role_cfg = {"cop": CFG_AGENT_COP, "thief": CFG_AGENT_THIEF} if args.per_role_configs else {"cop": CFG_AGENT, "thief": CFG_AGENT}
    sched = {k: v for k, v in (("random_timesteps", args.random_timesteps), ("learning_starts", args.learning_starts)) if v is not None}
    role_cfg = {r: dataclasses.replace(c, **sched) for r, c in role_cfg.items()}
    rank = int(os.environ.get("RANK", "0"))
    res = run_self_play(args.map, args.envs, args.out, iterations=args.iterations, training=tc, trainer_cfg=tcfg, role_cfg=role_cfg,
                        num_rays=args.rays, n_cops=args.cops, n_thieves=args.thieves, max_step_count=args.max_step_count,
                        eval_envs=args.eval_envs, seed=args.seed, log=print if rank == 0 else (lambda *a, **k: None))
    if backend:
        import torch.distributed as dist
        print(f"[self-play] rank {res['rank']}/{res['world']}: {res['envs_local']} envs from global id {res['env_id_offset']}, all-reduce over "
              f"{backend}, parameters {res['param_digest'][:16]}", flush=True)
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
