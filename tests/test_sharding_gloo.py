"""N>1 path on CPU: world_size-2 gloo processes.  Each rank owns a contiguous block of global env
ids and needs no data-path collective: the union of the ranks' (oracle-simulated) shards equals
one big batch bit for bit, and the benchmark's MAX-over-ranks timing reduction works."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parents[1]


def test_shard_envs_partitions_exactly():
    from as_cops_and_thieves_amd.sharding import shard_envs
    for total, world in ((32768, 8), (4096, 1), (37, 5), (7, 8)):
        parts = [shard_envs(total, r, world) for r in range(world)]
        assert sum(n for n, _ in parts) == total
        off = 0
        for n, o in parts:
            assert o == off
            off += n
    with pytest.raises(ValueError):
        shard_envs(8, 3, 2)


def _worker(rank, world, port, total, ticks, q):
    sys.path.insert(0, str(ROOT))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from as_cops_and_thieves_amd.config import SimConfig
    from as_cops_and_thieves_amd.maps import load_preset
    from as_cops_and_thieves_amd.sharding import max_over_ranks, shard_envs, sum_over_ranks
    from oracle.cat_oracle import OracleSim
    n, off = shard_envs(total, rank, world)
    cmap = load_preset("squarinth").compile()
    sim = OracleSim(SimConfig(n_envs=n, n_rays=16, seed=4, max_step_count=15, env_id_offset=off), [cmap])
    sim.reset()
    for t in range(ticks):
        out = sim.step(sim.random_actions(t))
        sim.reset(mask=out["terminated"].copy())
    st = sim.get_state()
    slow = max_over_ranks(1.0 + rank)            # pretend rank r took (1 + r) seconds
    steps = sum_over_ranks(float(n * ticks))
    q.put((rank, off, st["pos"].copy(), out["obs_distance"].copy(), slow, steps))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("total,world", [(10, 2), (21, 8)])   # 8: the rank count of the node the driver's SCALE run uses (ragged shards of 3 and 2 envs)
def test_rank_shards_equal_one_batch(total, world):
    ticks = 40
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, ticks, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=120) for _ in range(world)])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    from as_cops_and_thieves_amd.config import SimConfig
    from as_cops_and_thieves_amd.maps import load_preset
    from oracle.cat_oracle import OracleSim
    sim = OracleSim(SimConfig(n_envs=total, n_rays=16, seed=4, max_step_count=15), [load_preset("squarinth").compile()])
    sim.reset()
    for t in range(ticks):
        out = sim.step(sim.random_actions(t))
        sim.reset(mask=out["terminated"].copy())
    pos = np.concatenate([g[2] for g in got]); obs = np.concatenate([g[3] for g in got])
    assert np.array_equal(pos, sim.get_state()["pos"]) and np.array_equal(obs, out["obs_distance"])
    assert all(g[4] == float(world) for g in got)          # MAX over ranks (rank r reported 1 + r)
    assert all(g[5] == total * ticks for g in got)  # whole-job env-steps


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _adv_worker(rank, world, port, q):
    sys.path.insert(0, str(ROOT))
    import torch
    import torch.distributed as dist
    from as_cops_and_thieves_amd.selfplay.mappo import advantage_moments
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    gen = torch.Generator().manual_seed(7)
    full = torch.randn(3, 16, 64, generator=gen) * torch.tensor([1.0, 5.0, 0.1]).view(3, 1, 1) + torch.tensor([0.0, 2.0, -1.0]).view(3, 1, 1)
    shard = full[:, :, rank * 32:(rank + 1) * 32].contiguous()           # env shards: the N axis
    mean, std = advantage_moments(shard)
    q.put((rank, mean.flatten().tolist(), std.flatten().tolist(), full.mean(dim=(1, 2)).tolist(), full.std(dim=(1, 2)).tolist()))
    dist.barrier()
    dist.destroy_process_group()


def test_data_parallel_ranks_normalise_advantages_with_the_whole_batch_statistics():
    """with env shards on several ranks the advantage mean / std must be those of the WHOLE batch (what one process with
    all envs computes), not rank-local ones: two gloo ranks, half the envs each."""
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_adv_worker, args=(r, 2, port, q)) for r in range(2)]
    for p_ in procs:
        p_.start()
    res = [q.get(timeout=240) for _ in range(2)]
    for p_ in procs:
        p_.join(timeout=60)
        assert p_.exitcode == 0
    for rank, mean, std, want_mean, want_std in res:
        assert np.allclose(mean, want_mean, rtol=1e-6, atol=1e-7) and np.allclose(std, want_std, rtol=1e-6), rank
    assert res[0][1] == res[1][1] and res[0][2] == res[1][2]            # every rank uses the same numbers
