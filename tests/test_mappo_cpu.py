"""MAPPO trainer host logic on CPU (oracle-backed env stand-in): GAE scan, update mechanics,
role freezing, per-role checkpoints, 2-rank gloo gradient all-reduce."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from as_cops_and_thieves_amd.maps import load_preset
from as_cops_and_thieves_amd.selfplay.mappo import MAPPOConfig, MAPPOTrainer, compute_gae
from as_cops_and_thieves_amd.selfplay.models import LSTMPolicy, LSTMValue, conv_out_len
from tests.fake_env import OracleVecEnv

ROOT = Path(__file__).resolve().parents[1]


def test_conv_sizes_follow_ray_count():
    assert conv_out_len(90) == 13                       # reference: Linear(32 * 13, 256) at 90 rays (Q10)
    assert conv_out_len(64) == 9
    p, v = LSTMPolicy(64), LSTMValue(64)
    x = torch.randn(3, 5, 2 * 64)
    logits, st = p(x, p.initial_state(3, "cpu"))
    assert logits.shape == (3, 5, 4) and st[0].shape == (1, 3, 128)
    vals, st = v(torch.randn(3, 5, 4 * 64), v.initial_state(3, "cpu"))
    assert vals.shape == (3, 5) and st[0].shape == (2, 3, 128)
    n_pol = sum(q.numel() for q in LSTMPolicy(90).parameters())
    assert 300_000 < n_pol < 380_000                    # LSTMPolicy ~ 340 k at R = 90


def test_recurrent_state_resets_at_episode_starts():
    torch.manual_seed(0)
    p = LSTMPolicy(16)
    x = torch.randn(2, 6, 32)
    starts = torch.zeros(2, 6, dtype=torch.bool)
    starts[0, 3] = True
    full, _ = p(x, p.initial_state(2, "cpu"), starts)
    tail, _ = p(x[:1, 3:], p.initial_state(1, "cpu"))   # env 0 restarted at t = 3: same as a fresh sequence
    assert torch.allclose(full[0, 3:], tail[0], atol=1e-6)
    cont, _ = p(x, p.initial_state(2, "cpu"))
    assert torch.allclose(full[1], cont[1], atol=1e-6) and not torch.allclose(full[0, 3:], cont[0, 3:])


def test_gae_matches_textbook_recursion():
    rng = np.random.default_rng(0)
    T, N, g, l = 12, 5, 0.99, 0.95
    r, v = rng.normal(size=(T, N)), rng.normal(size=(T, N))
    d = rng.random((T, N)) < 0.2
    last = rng.normal(size=N)
    adv, ret = compute_gae(torch.tensor(r), torch.tensor(v), torch.tensor(d), torch.tensor(last), g, l)
    want = np.zeros((T, N))
    for n in range(N):
        run = 0.0
        for t in reversed(range(T)):
            nv = last[n] if t == T - 1 else v[t + 1, n]
            nd = 0.0 if d[t, n] else 1.0
            delta = r[t, n] + g * nv * nd - v[t, n]
            run = delta + g * l * nd * run
            want[t, n] = run
    assert np.allclose(adv.numpy(), want) and np.allclose(ret.numpy(), want + v)


def _env(n=8, seed=1, off=0):
    return OracleVecEnv(load_preset("squarinth").compile(), n, num_rays=16, max_step_count=12, seed=seed, env_id_offset=off)


def test_trainer_updates_parameters_and_respects_freeze():
    tr = MAPPOTrainer(_env(), MAPPOConfig(horizon=8, learning_epochs=2, mini_batches=2, frozen_roles=("thief",), kl_threshold=0.0))
    before = {a: [q.clone() for q in tr.policies[a].parameters()] for a in tr.agents}
    vbefore = [q.clone() for q in tr.values["thief_0"].parameters()]
    stats = tr.train(2)
    assert all(np.isfinite(v) for v in stats.values())
    changed = lambda a: any(not torch.equal(x, y) for x, y in zip(before[a], tr.policies[a].parameters()))
    assert changed("cop_0") and changed("cop_1") and not changed("thief_0")          # frozen policy, critic still learns
    assert any(not torch.equal(x, y) for x, y in zip(vbefore, tr.values["thief_0"].parameters()))
    sd = tr.role_state_dict("cop")
    assert set(sd) == {"cop_0", "cop_1"}
    tr2 = MAPPOTrainer(_env(), MAPPOConfig(horizon=8))
    tr2.load_role_state_dict(sd)
    assert all(torch.equal(x, y) for x, y in zip(tr.policies["cop_1"].parameters(), tr2.policies["cop_1"].parameters()))


def test_random_timesteps_and_learning_starts():
    tr = MAPPOTrainer(_env(), MAPPOConfig(horizon=4, random_timesteps=100, learning_starts=100))
    before = [q.clone() for q in tr.policies["cop_0"].parameters()]
    tr.train(3)
    assert tr.timestep == 12 and all(torch.equal(x, y) for x, y in zip(before, tr.policies["cop_0"].parameters()))


def _ddp_worker(rank, world, port, q):
    sys.path.insert(0, str(ROOT))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    tr = MAPPOTrainer(_env(4, seed=3, off=4 * rank), MAPPOConfig(horizon=6, learning_epochs=1, mini_batches=1, kl_threshold=0.0), seed=0)
    tr.train(2)
    flat = torch.cat([p.detach().reshape(-1) for p in tr.policies["cop_0"].parameters()])
    q.put((rank, flat.numpy().copy()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gradient_allreduce_keeps_replicas_identical():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_ddp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=300) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # different env shards (different data), same seed for the nets: after all-reduced updates the replicas agree
    assert np.array_equal(got[0], got[1])


@pytest.mark.parametrize("layers,channels", [(1, 2), (2, 4)])
def test_manual_recurrence_equals_nn_lstm(layers, channels):
    """The trunk evaluates the LSTM recurrence itself from nn.LSTM's parameters; it must match nn.LSTM (fp32)."""
    from as_cops_and_thieves_amd.selfplay.models import _Trunk
    torch.manual_seed(3)
    tr = _Trunk(channels, 16, 24, layers)
    x = torch.randn(5, 7, channels * 16)
    h0, c0 = torch.randn(layers, 5, 24), torch.randn(layers, 5, 24)
    out, (h, c) = tr(x, (h0, c0))
    f = tr.features(x.reshape(35, channels, 16)).reshape(5, 7, 256)
    want, (hw, cw) = tr.lstm(f, (h0, c0))
    assert torch.allclose(out, want, atol=1e-5) and torch.allclose(h, hw, atol=1e-5) and torch.allclose(c, cw, atol=1e-5)
    # gradients flow to the nn.LSTM parameters
    out.sum().backward()
    assert all(q.grad is not None and torch.isfinite(q.grad).all() for q in tr.lstm.parameters())


def test_conv_as_gemm_equals_conv1d():
    from as_cops_and_thieves_amd.selfplay.models import _Trunk, conv1d_as_gemm
    torch.manual_seed(4)
    tr = _Trunk(4, 64, 16, 1)
    x = torch.randn(6, 4, 64)
    assert torch.allclose(conv1d_as_gemm(x, tr.features[0]), tr.features[0](x), atol=1e-5)
    y = torch.relu(tr.features[0](x))
    assert torch.allclose(conv1d_as_gemm(y, tr.features[2]), tr.features[2](y), atol=1e-5)
    # and the trunk's feature path as a whole
    z = torch.relu(conv1d_as_gemm(torch.relu(conv1d_as_gemm(x, tr.features[0])), tr.features[2]))
    assert torch.allclose(torch.tanh(tr.features[5](z.flatten(1))), tr.features(x), atol=1e-5)
