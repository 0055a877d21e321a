"""The data-parallel self-play command's loop on CPU: two gloo ranks run ``run_self_play`` on disjoint env shards of ONE job and one
shared output directory.  Rank 0 alone may evaluate and write (checkpoints, archives, win_rates.json) -- rank 1's writers are
replaced by functions that raise --, both ranks must hold bit-identical parameters afterwards, and a second call resumes from the
files rank 0 wrote."""
import os
import socket
import sys
from pathlib import Path

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parents[1]


def _worker(rank, world, port, out_dir, q):
    sys.path.insert(0, str(ROOT))
    import warnings
    warnings.filterwarnings("ignore")
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from as_cops_and_thieves_amd.maps import load_preset
    from as_cops_and_thieves_amd.selfplay import archive, self_play
    from as_cops_and_thieves_amd.selfplay.mappo import RoleConfig, TrainerConfig
    from tests.fake_env import OracleVecEnv
    cmap = load_preset("squarinth").compile()
    seen = []

    def factory(n, s, off=0):
        seen.append((n, s, off))
        return OracleVecEnv(cmap, n, num_rays=16, max_step_count=12, seed=s, env_id_offset=off)
    if rank != 0:      # nothing but rank 0 may touch the output directory or play evaluation episodes
        def forbidden(*a, **k):
            raise AssertionError("a rank other than 0 evaluated or wrote a file")
        archive.add_policy_to_archive = forbidden
        archive.update_policy_win_rate = forbidden
        self_play.evaluate_agents = forbidden
        torch.save = forbidden
    rc = RoleConfig(learning_epochs=1, mini_batches=2, random_timesteps=4, learning_starts=8, kl_threshold=0.0)
    tc = TrainerConfig(horizon=4, timesteps=16, policy_freeze_duration=8, opponent_freeze_duration=8)
    kw = dict(training=self_play.TrainingConfig(n_trial_episodes=3, num_opponents_to_evaluate=2), trainer_cfg=tc,
              role_cfg={"cop": rc, "thief": rc}, env_factory=factory, log=lambda *a: None)
    total = 10 if world <= 4 else 20          # every rank needs two training sequences for the two minibatches of `rc`
    res = self_play.run_self_play("squarinth", total, out_dir, iterations=3, **kw)
    res2 = self_play.run_self_play("squarinth", total, out_dir, iterations=1, **kw)        # resumes after iteration 2
    q.put((rank, res["param_digest"], res["envs_local"], res["env_id_offset"], [h["iteration"] for h in res["iterations"]],
           [h["iteration"] for h in res2["iterations"]], res2["param_digest"], seen[0],
           sum(len(v) for h in res["iterations"] for v in h["evaluations"].values())))
    dist.barrier()
    dist.destroy_process_group()


def _run_job(world, out_dir):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, out_dir, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=600) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0                                         # every rank ended by itself: no process left behind
    return got


def test_two_ranks_share_one_self_play_job(tmp_path):
    r0, r1 = _run_job(2, tmp_path)
    assert r0[1] == r1[1] and r0[6] == r1[6] and r0[1] != r0[6]       # identical replicas after each call; training moved them
    assert (r0[2], r0[3]) == (5, 0) and (r1[2], r1[3]) == (5, 5)      # 10 envs: two disjoint contiguous shards
    assert r0[7] == (5, 0, 0) and r1[7] == (5, 0, 5)                  # env_factory(n_local, seed, env_id_offset)
    assert r0[4] == r1[4] == [0, 1, 2] and r0[5] == r1[5] == [3]      # both ranks resumed from rank 0's files
    assert r0[8] > 0 and r1[8] == 0                                   # only rank 0 evaluated
    for role, d in (("cop", "cops"), ("thief", "thieves")):
        assert sorted(p.name for p in (tmp_path / d).glob("*.pt")) == [f"{role}_iter_{i}.pt" for i in range(4)]
    assert sorted(p.name for p in tmp_path.glob("joint_iter_*")) == [f"joint_iter_{i}_full_agent.pt" for i in range(4)]
    assert (tmp_path / "thieves" / "win_rates.json").exists()


def test_eight_ranks_share_one_self_play_job(tmp_path):
    """The rank count of the 8-GPU node (BASELINE configs[2]), on CPU: 20 envs over eight gloo ranks = ragged shards (3, 3, 3, 3, 2, 2, 2, 2) at disjoint
    contiguous offsets; every optimiser step all-reduces the gradient | KL buffer over all eight, so all eight end with bit-identical parameters after
    each call; rank 0 alone evaluated and wrote; every rank resumed from rank 0's files; all eight processes exit by themselves."""
    got = _run_job(8, tmp_path)
    assert len(got) == 8 and [g[0] for g in got] == list(range(8))
    assert len({g[1] for g in got}) == 1 and len({g[6] for g in got}) == 1 and got[0][1] != got[0][6]
    assert [(g[2], g[3]) for g in got] == [(3, 0), (3, 3), (3, 6), (3, 9), (2, 12), (2, 14), (2, 16), (2, 18)]
    assert all(g[7] == (g[2], 0, g[3]) for g in got)
    assert all(g[4] == [0, 1, 2] and g[5] == [3] for g in got)
    assert got[0][8] > 0 and all(g[8] == 0 for g in got[1:])
    for role, d in (("cop", "cops"), ("thief", "thieves")):
        assert sorted(p.name for p in (tmp_path / d).glob("*.pt")) == [f"{role}_iter_{i}.pt" for i in range(4)]


def _refusing_worker(rank, world, port, q):
    sys.path.insert(0, str(ROOT))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from as_cops_and_thieves_amd.selfplay import self_play
    from as_cops_and_thieves_amd.selfplay.mappo import RoleConfig, TrainerConfig
    rc = RoleConfig(learning_epochs=1, mini_batches=2)
    try:
        self_play.run_self_play("squarinth", 3, "/nonexistent", iterations=1, trainer_cfg=TrainerConfig(horizon=4, timesteps=8), role_cfg={"cop": rc, "thief": rc},
                                env_factory=lambda n, s, off=0: None, log=lambda *a: None)
        q.put((rank, "no error"))
    except ValueError as exc:
        q.put((rank, str(exc)))
    dist.destroy_process_group()


def test_a_shard_too_small_for_the_minibatches_is_refused_by_every_rank_together():
    """3 envs over 2 ranks with two minibatches per update: rank 1's single env cannot fill them.  Every rank computes every rank's shard and raises
    the same error before anything is built -- a rank failing alone would leave the other waiting in the trainer's first all-reduce."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_refusing_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all("fewer than the 2 minibatches" in msg for _, msg in got), got
