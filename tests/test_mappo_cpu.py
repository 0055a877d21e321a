"""MAPPO trainer host logic on CPU (oracle-backed env stand-in): stacked networks vs the reference-shaped
per-agent modules, GAE scan, the timestep schedule, masked Adam vs torch.optim.Adam, per-minibatch KL early
stop, checkpoints, and the 2-rank gloo path (different env shards, identical parameters afterwards)."""
import os
import socket
import sys
import warnings
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from as_cops_and_thieves_amd.maps import load_preset
from as_cops_and_thieves_amd.selfplay.mappo import (CFG_AGENT, CFG_AGENT_COP, CFG_AGENT_THIEF, MAPPOTrainer, RoleConfig,
                                                     TrainerConfig, compute_gae)
from as_cops_and_thieves_amd.selfplay.models import LSTMPolicy, LSTMValue, conv_out_len
from as_cops_and_thieves_amd.selfplay.stacked import (FlatParams, StackedNet, agent_state_dict, init_from_modules,
                                                       role_param_shapes)
from tests.fake_env import OracleVecEnv

ROOT = Path(__file__).resolve().parents[1]
warnings.filterwarnings("ignore", message="grad and param do not obey the gradient layout contract")


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


def test_reference_hyperparameters():
    """src/configs/mappo_config.py:5-50 (the driver uses CFG_AGENT for every agent)."""
    for c in (CFG_AGENT, CFG_AGENT_COP):
        assert (c.learning_epochs, c.mini_batches, c.learning_rate, c.ratio_clip, c.entropy_loss_scale) == (4, 4, 1e-4, 0.15, 0.02)
    t = CFG_AGENT_THIEF
    assert (t.learning_epochs, t.mini_batches, t.learning_rate, t.ratio_clip, t.entropy_loss_scale) == (3, 8, 3e-4, 0.2, 0.01)
    for c in (CFG_AGENT, CFG_AGENT_COP, CFG_AGENT_THIEF):
        assert (c.random_timesteps, c.learning_starts, c.kl_threshold, c.value_loss_scale, c.grad_norm_clip) == (10_000, 15_000, 0.015, 0.5, 0.5)
    tc = TrainerConfig()
    assert (tc.timesteps, tc.opponent_freeze_duration, tc.policy_freeze_duration) == (100_000, 15_000, 15_000)
    assert (tc.horizon, tc.bptt) == (128, 16)      # build-side: 8 BPTT windows of the reference's sequence length per rollout


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


@pytest.mark.parametrize("kind,Mod,C", [("policy", LSTMPolicy, 2), ("value", LSTMValue, 4)])
def test_stacked_networks_equal_the_per_agent_modules(kind, Mod, C):
    """Forward values and every parameter gradient of the role-stacked evaluation (batched GEMMs, one LSTM autograd
    node with a hand-written backward) against the reference-shaped module of each agent, incl. episode restarts
    inside the BPTT window and a non-zero incoming state."""
    torch.manual_seed(0)
    R, G, T, B = 16, 2, 5, 3
    fp = FlatParams(role_param_shapes(R), G, "cpu", torch.float32)
    init_from_modules(fp, R, [11, 12])
    net = StackedNet(kind, R, fp)
    x = torch.randn(G, T, B, C * R)
    starts = torch.zeros(T, B, dtype=torch.bool)
    starts[2, 1] = starts[0, 0] = True
    st = net.initial_state(B)
    st = (torch.randn_like(st[0]), torch.randn_like(st[1]))
    out, (h, c) = net.forward(x, st, (~starts).float())
    fp.grad.zero_()
    ((out ** 2).sum() + 0.3 * h.sum() + (c ** 2).sum()).backward()
    for g in range(G):
        m = Mod(R)
        m.load_state_dict(agent_state_dict(fp, g)[kind])            # same parameter names as the reference modules
        o, (hm, cm) = m(x[g].transpose(0, 1), (st[0][:, g], st[1][:, g]), starts.t())
        o = o.unsqueeze(-1) if kind == "value" else o
        assert torch.allclose(o.transpose(0, 1), out[g], atol=1e-6) and torch.allclose(hm, h[:, g], atol=1e-6)
        ((o ** 2).sum() + 0.3 * hm.sum() + (cm ** 2).sum()).backward()
        for n, q in m.named_parameters():
            off, k, shp = fp.offsets[f"{kind}.{n}"]
            got = fp.grad[g, off:off + k].view(shp)
            assert torch.allclose(got, q.grad, rtol=1e-4, atol=1e-6 * float(q.grad.abs().max() + 1)), n


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
    # the stacked form: a leading agent axis
    adv2, _ = compute_gae(torch.tensor(r)[None].repeat(2, 1, 1), torch.tensor(v)[None].repeat(2, 1, 1), torch.tensor(d),
                          torch.tensor(last)[None].repeat(2, 1), g, l)
    assert np.allclose(adv2[1].numpy(), want)


def _env(n=8, seed=1, off=0):
    return OracleVecEnv(load_preset("squarinth").compile(), n, num_rays=16, max_step_count=12, seed=seed, env_id_offset=off)


def _rc(**kw):
    base = dict(learning_epochs=2, mini_batches=2, random_timesteps=0, learning_starts=0, kl_threshold=0.0)
    base.update(kw)
    return RoleConfig(**base)


def test_timestep_schedule_random_phase_learning_starts_and_freeze_durations():
    """mappo_config.py:9-10,61-62 / README.md:58-154 at rollout granularity: no update before learning_starts, the
    policies stay frozen until policy_freeze_duration, the critics train from the first update on."""
    rc = _rc(random_timesteps=8, learning_starts=16)
    tr = MAPPOTrainer(_env(), {"cop": rc, "thief": rc}, TrainerConfig(horizon=8, timesteps=48, policy_freeze_duration=24,
                                                                     opponent_freeze_duration=24))
    assert list(tr.roles) == ["cop+thief"]          # same RoleConfig for both roles: one stacked learner of 3 agents
    rl = tr.roles["cop+thief"]
    assert rl.agents == ["cop_0", "cop_1", "thief_0"]
    p0 = rl.fp.master.clone()
    tr.train()
    assert tr.timestep == 48
    # rollouts start at t = 0 (random), 8, 16, 24, 32, 40; updates follow the rollouts ending at 16 .. 48: five of them,
    # each 2 epochs x 2 minibatches; the policies join at the rollout that contains t = 24: three updates
    assert float((rl.steps * rl.col_value).max()) == 20 and float((rl.steps * rl.col_policy).max()) == 12
    d = rl.fp.master - p0
    assert float((d * rl.col_policy).abs().max()) > 0 and float((d * rl.col_value).abs().max()) > 0
    tr2 = MAPPOTrainer(_env(), {"cop": rc, "thief": rc}, TrainerConfig(horizon=8, timesteps=24, policy_freeze_duration=1000,
                                                                      opponent_freeze_duration=1000))
    rl2, g = tr2.learner_of("thief_0")
    q0 = rl2.fp.master.clone()
    tr2.train()
    d2 = (rl2.fp.master - q0)[g]
    assert float((d2 * rl2.col_policy).abs().max()) == 0.0        # frozen for the whole call
    assert float((d2 * rl2.col_value).abs().max()) > 0


def test_random_and_learning_thresholds_act_per_role():
    """skrl reads random_timesteps / learning_starts from each agent's own cfg: with CFG_AGENT_COP / CFG_AGENT_THIEF-style
    per-role values the thief starts updating two rollouts before the cop does (one update = 1 epoch x 2 minibatches)."""
    cop, thief = _rc(random_timesteps=8, learning_starts=16), _rc(random_timesteps=0, learning_starts=8, learning_rate=2e-4)
    tr = MAPPOTrainer(_env(), {"cop": cop, "thief": thief}, TrainerConfig(horizon=4, timesteps=24, policy_freeze_duration=0,
                                                                         opponent_freeze_duration=0), seed=1)
    tr.train()
    # rollouts end at t = 4, 8, ..., 24: thief updates at 8..24 (5 rollouts), cop from 16 (t0 >= 8 and t >= 16: 3 rollouts)
    nb = cop.learning_epochs * cop.mini_batches
    assert float(tr.roles["thief"].steps.max()) == 5 * nb and float(tr.roles["cop"].steps.max()) == 3 * nb


def test_masked_adam_equals_torch_adam_and_frozen_entries_do_not_move():
    """The flat masked Adam against torch.optim.Adam on the same gradients (what skrl constructs), with the policy
    half frozen for the first steps: frozen entries keep their value, their moments and their step count."""
    tr = MAPPOTrainer(_env(), {"cop": _rc(learning_rate=1e-2, grad_norm_clip=1e9), "thief": _rc()}, TrainerConfig(horizon=4))
    assert sorted(tr.roles) == ["cop", "thief"]     # different RoleConfigs: one learner per role
    rl = tr.roles["cop"]
    gen = torch.Generator().manual_seed(0)
    pol = rl.col_policy.bool()
    for k in range(6):
        frozen = k < 2
        rl.set_frozen("cop", policy=frozen, value=False)
        g = torch.randn(rl.G, rl.fp.P, generator=gen) * rl.fp.column_mask("")     # padding columns carry no gradient
        rl.ar[:, :-1].copy_(g); rl.ar[:, -1] = 0.0
        rl.epoch_active.fill_(1.0)
        before = rl.fp.master.clone()
        rl._step_apply()
        if frozen:
            assert torch.equal(rl.fp.master[:, pol], before[:, pol])
    # the value half took 6 steps of plain Adam, the policy half 4 (steps 2..5): replay both with torch.optim.Adam
    gen = torch.Generator().manual_seed(0)
    grads = [torch.randn(rl.G, rl.fp.P, generator=gen) * rl.fp.column_mask("") for _ in range(6)]
    w_v = torch.nn.Parameter(torch.zeros(rl.G, int((~pol).sum())))
    w_p = torch.nn.Parameter(torch.zeros(rl.G, int(pol.sum())))
    tr0 = MAPPOTrainer(_env(), {"cop": _rc(learning_rate=1e-2), "thief": _rc()}, TrainerConfig(horizon=4))   # same seed: same initial weights
    w_v.data.copy_(tr0.roles["cop"].fp.master[:, ~pol]); w_p.data.copy_(tr0.roles["cop"].fp.master[:, pol])
    ov, op = torch.optim.Adam([w_v], lr=1e-2), torch.optim.Adam([w_p], lr=1e-2)
    for k, g in enumerate(grads):
        w_v.grad = g[:, ~pol].clone(); ov.step()
        if k >= 2:
            w_p.grad = g[:, pol].clone(); op.step()
    assert torch.allclose(rl.fp.master[:, ~pol], w_v.data, atol=1e-6)
    assert torch.allclose(rl.fp.master[:, pol], w_p.data, atol=1e-6)


def test_kl_early_stop_skips_the_rest_of_the_epoch_per_agent():
    """skrl's per-minibatch check: a minibatch whose KL exceeds the threshold is not applied and ends THAT agent's epoch;
    the next epoch starts again."""
    tr = MAPPOTrainer(_env(), {"cop": _rc(kl_threshold=0.5), "thief": _rc()}, TrainerConfig(horizon=4))
    rl = tr.roles["cop"]
    rl.epoch_active.fill_(1.0)
    for kl0, kl1, want in ((0.1, 0.9, (1.0, 0.0)), (0.1, 0.1, (1.0, 0.0)), (0.9, 0.1, (0.0, 0.0))):
        rl.ar.zero_(); rl.ar[:, :-1] = 1e-3
        rl.ar[0, -1], rl.ar[1, -1] = kl0, kl1
        s0 = rl.steps.clone()
        rl._step_apply()
        took = (rl.steps - s0).amax(dim=1)
        assert tuple(took.tolist()) == want              # agent 1 stopped at the first minibatch, agent 0 at the third
    rl.epoch_active.fill_(1.0)                            # next epoch
    rl.ar[:, -1] = 0.0
    s0 = rl.steps.clone()
    rl._step_apply()
    assert tuple((rl.steps - s0).amax(dim=1).tolist()) == (1.0, 1.0)


class _RefShapedPolicy(torch.nn.Module):
    """A module written with the attribute names and layer list of the reference's LSTMPolicy (lstm_policy_net.py:28-53), NOT
    imported from the product: what the reference's own class registers, minus skrl's mixins (which add no parameters)."""

    def __init__(self, L2):
        super().__init__()
        nn = torch.nn
        self.features_extractor = nn.Sequential(nn.Conv1d(2, 64, kernel_size=5, stride=2, padding=0), nn.ReLU(),
                                                nn.Conv1d(64, 32, kernel_size=5, stride=3, padding=0), nn.ReLU(), nn.Flatten(),
                                                nn.Linear(32 * L2, 256), nn.Tanh())
        self.lstm = nn.LSTM(input_size=256, hidden_size=128, num_layers=1, batch_first=True)
        self.policy_head = nn.Sequential(nn.Linear(128, 128), nn.ReLU(), nn.Linear(128, 64), nn.ReLU(), nn.Linear(64, 4))


class _RefShapedValue(torch.nn.Module):
    """lstm_value_net.py:46-75 likewise."""

    def __init__(self, L2):
        super().__init__()
        nn = torch.nn
        self.features_extractor = nn.Sequential(nn.Conv1d(4, 64, kernel_size=5, stride=2, padding=0), nn.ReLU(),
                                                nn.Conv1d(64, 32, kernel_size=5, stride=3, padding=0), nn.ReLU(), nn.Flatten(),
                                                nn.Linear(32 * L2, 256), nn.Tanh())
        self.lstm = nn.LSTM(input_size=256, hidden_size=128, num_layers=2, batch_first=True)
        self.value_head = nn.Sequential(nn.Linear(128, 256), nn.ReLU(), nn.Linear(256, 128), nn.ReLU(), nn.Linear(128, 64), nn.ReLU(),
                                        nn.Linear(64, 1))


def test_checkpoints_use_the_reference_module_names_and_resume_exactly(tmp_path):
    rc = _rc()
    tr = MAPPOTrainer(_env(), {"cop": rc, "thief": rc}, TrainerConfig(horizon=4, timesteps=8, policy_freeze_duration=0,
                                                                     opponent_freeze_duration=0), seed=3)
    tr.train()
    sd = tr.state_dict()
    torch.save(sd, tmp_path / "joint_iter_0_full_agent.pt")
    sd = torch.load(tmp_path / "joint_iter_0_full_agent.pt", weights_only=True)     # tensors, numbers, strings only
    # skrl's MAPPO.save layout: {agent: {"policy", "value", "optimizer"}} (+ the trainer position under a non-agent key)
    assert set(sd) == {"cop_0", "cop_1", "thief_0", "__cat__"} and all(set(sd[a]) == {"policy", "value", "optimizer"} for a in tr.agents)
    LSTMPolicy(16).load_state_dict(sd["cop_1"]["policy"])
    LSTMValue(16).load_state_dict(sd["thief_0"]["value"])
    # strict load into modules carrying the REFERENCE's attribute names, and their own Adam accepts the optimiser entry
    L2 = conv_out_len(16)
    ref_p, ref_v = _RefShapedPolicy(L2), _RefShapedValue(L2)
    ref_p.load_state_dict(sd["cop_1"]["policy"], strict=True)
    ref_v.load_state_dict(sd["cop_1"]["value"], strict=True)
    opt = torch.optim.Adam(list(ref_p.parameters()) + list(ref_v.parameters()), lr=1e-4)
    opt.load_state_dict(sd["cop_1"]["optimizer"])
    rl, g = tr.learner_of("cop_1")
    o, k, shp = rl.fp.offsets["value.value_head.6.weight"]
    assert torch.equal(opt.state[ref_v.value_head[6].weight]["exp_avg"], rl.m[g, o:o + k].view(shp))
    # ... and the way back: a checkpoint the reference would write (module state dicts + torch Adam state, no "__cat__")
    with torch.no_grad():
        for q in list(ref_p.parameters()) + list(ref_v.parameters()):
            q.add_(0.25)
    theirs = {a: {"policy": ref_p.state_dict(), "value": ref_v.state_dict(), "optimizer": opt.state_dict()} for a in tr.agents}
    torch.save(theirs, tmp_path / "from_reference.pt")
    tr3 = MAPPOTrainer(_env(), {"cop": rc, "thief": rc}, TrainerConfig(horizon=4), seed=5)
    tr3.load_state_dict(torch.load(tmp_path / "from_reference.pt", weights_only=True))
    assert torch.equal(tr3.agent_models("thief_0")["policy"]["lstm.weight_hh_l0"], ref_p.lstm.weight_hh_l0.detach())
    assert torch.equal(tr3.agent_models("cop_0")["value"]["value_head.0.bias"], ref_v.value_head[0].bias.detach())
    # resume into a trainer whose agents are stacked DIFFERENTLY (one learner per role): checkpoints are per agent
    tr2 = MAPPOTrainer(_env(), {"cop": rc, "thief": _rc(learning_rate=5e-4)}, TrainerConfig(horizon=4), seed=99)
    assert sorted(tr2.roles) == ["cop", "thief"]
    tr2.load_state_dict(sd)
    for a in tr.agents:
        (l1, g1), (l2, g2) = tr.learner_of(a), tr2.learner_of(a)
        assert torch.equal(l1.fp.master[g1], l2.fp.master[g2]) and torch.equal(l1.m[g1], l2.m[g2])
        assert torch.equal(l1.steps[g1], l2.steps[g2])
    assert tr2.timestep == tr.timestep
    # round-2 files ("cat-mappo-2": models / optimizers at the top, trunk.* / head.* keys) still load
    old = {"format": "cat-mappo-2", "timestep": 8, "num_rays": 16, "optimizers": {}, "models": {
        a: {kind: {(("trunk.features." + n[len("features_extractor."):]) if n.startswith("features_extractor.") else
                    ("trunk." + n) if n.startswith("lstm.") else "head." + n.split(".", 1)[1]): v for n, v in sd[a][kind].items()}
            for kind in ("policy", "value")} for a in tr.agents}}
    tr4 = MAPPOTrainer(_env(), {"cop": rc, "thief": rc}, TrainerConfig(horizon=4), seed=6)
    tr4.load_state_dict(old)
    assert all(torch.equal(tr4.learner_of(a)[0].fp.master[tr4.learner_of(a)[1]], tr.learner_of(a)[0].fp.master[tr.learner_of(a)[1]]) for a in tr.agents)


def _ddp_worker(rank, world, port, q, kl):
    sys.path.insert(0, str(ROOT))
    warnings.filterwarnings("ignore")
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    rc = _rc(learning_epochs=2, mini_batches=2, kl_threshold=kl, learning_rate=3e-3)
    tr = MAPPOTrainer(_env(4, seed=3, off=4 * rank), {"cop": rc, "thief": rc},
                      TrainerConfig(horizon=6, timesteps=18, policy_freeze_duration=0, opponent_freeze_duration=0), seed=0)
    tr.train()
    flat = torch.cat([rl.fp.master.reshape(-1) for rl in tr.roles.values()])
    steps = torch.cat([rl.steps.amax(dim=1) for rl in tr.roles.values()])
    q.put((rank, flat.numpy().copy(), steps.numpy().copy(), float(next(iter(tr.roles.values())).buf["pin"].double().sum())))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("kl", [0.0, 1e-7])
def test_two_rank_training_keeps_replicas_identical(kl):
    """Ranks own different env shards (different data) and the same initial weights; the gradients AND the KL
    statistics are all-reduced in one buffer, so both ranks take the same early-stop decisions (a tiny threshold makes
    them stop) and end with bit-identical parameters."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_ddp_worker, args=(r, 2, port, q, kl)) for r in range(2)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(2):
        r, flat, steps, rew = q.get(timeout=300)
        got[r] = (flat, steps, rew)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got[0][2] != got[1][2]                                   # the shards really differ
    assert np.array_equal(got[0][0], got[1][0]) and np.array_equal(got[0][1], got[1][1])
    if kl:
        assert got[0][1].max() < 3 * 2 * 2                          # some minibatches were skipped, identically on both ranks


def test_rollout_of_several_bptt_windows_trains_on_window_sequences():
    """horizon = W x bptt: the rollout is cut into W * N training sequences of ``bptt`` ticks, each with the recurrent state
    it started from; the number of optimiser steps per update does not change."""
    rc = _rc(learning_epochs=1, mini_batches=2)
    tr = MAPPOTrainer(_env(), {"cop": rc, "thief": rc}, TrainerConfig(horizon=8, bptt=4, policy_freeze_duration=0,
                                                                     opponent_freeze_duration=0), seed=1)
    rl = tr.roles["cop+thief"]
    assert (rl.W, rl.bptt, rl.B) == (2, 4, 8) and rl.tb["pin"].shape[:3] == (3, 4, 16)
    tr.collect(); tr.collect()
    h_at_window_1 = tuple(s.clone() for s in rl.p0w)                 # [W, layers, G, N, H]
    assert float(h_at_window_1[0][1].abs().max()) > 0                # window 1 starts from a non-zero state
    tr.update()
    N = rl.N
    for w in range(2):
        sl = slice(w * N, (w + 1) * N)
        assert torch.equal(rl.tb["pin"][:, :, sl], rl.buf["pin"][:, w * 4:(w + 1) * 4])
        assert torch.equal(rl.tb["act"][:, :, sl], rl.buf["act"][:, w * 4:(w + 1) * 4])
        assert torch.equal(rl.start[:, sl], tr._start_buf[w * 4:(w + 1) * 4])
        assert torch.equal(rl.p0[0][:, :, sl], h_at_window_1[0][w]) and torch.equal(rl.v0[1][:, :, sl], rl.v0w[1][w])
    assert float(rl.steps.max()) == 2                                # 1 epoch x 2 minibatches


def test_non_recurrent_pair_and_the_two_model_initialisers():
    """src/models/policy_net.py:9-45, value_net.py:8-34 and src/utils/model_utils.py:45-121: the non-recurrent Policy / Value with
    the reference's attribute names (features_extractor + net; net), fed with the packed layouts of packing.py."""
    from as_cops_and_thieves_amd import packing
    from as_cops_and_thieves_amd.selfplay.models import (Policy, Value, initialize_lstm_models_for_mappo, initialize_models_for_mappo,
                                                         state_width)
    env = _env()
    obs, _ = env.reset()
    state = env.state()
    agents, R, nc = env.possible_agents, 16, env.nc
    models = initialize_models_for_mappo(agents, R, nc)
    assert set(models) == set(agents) and all(set(m) == {"policy", "value"} for m in models.values())
    pol, val = models["cop_0"]["policy"], models["thief_0"]["value"]
    assert isinstance(pol, Policy) and isinstance(val, Value)
    assert [n for n, _ in pol.named_children()] == ["features_extractor", "net"] and [n for n, _ in val.named_children()] == ["net"]
    logits = pol(packing.pack_policy_input(obs["cop_0"]))
    vin = packing.pack_value_input(state)
    assert vin.shape[1] == state_width(len(agents), nc, R) == val.net[0].in_features     # the critic sees the WHOLE shared state
    assert logits.shape == (env.num_envs, 4) and val(vin).shape == (env.num_envs, 1)
    assert [m.out_features for m in val.net if isinstance(m, torch.nn.Linear)] == [512, 256, 128, 64, 1]
    rec = initialize_lstm_models_for_mappo(agents, R)
    assert isinstance(rec["cop_1"]["policy"], LSTMPolicy) and isinstance(rec["cop_1"]["value"], LSTMValue)


@pytest.mark.parametrize("kind", ["policy", "value"])
def test_stacked_non_recurrent_networks_equal_the_per_agent_modules(kind):
    """The trainer's non-recurrent variant (TrainerConfig.recurrent = False: policy_net.py:9-45, value_net.py:8-34): forward values
    and every parameter gradient of the role-stacked evaluation against the reference-shaped ``Policy`` / ``Value`` of each agent."""
    from as_cops_and_thieves_amd.selfplay.models import Policy, Value
    torch.manual_seed(0)
    R, G, T, B, W = 16, 2, 3, 4, 140
    fp = FlatParams(role_param_shapes(R, "mlp", W), G, "cpu", torch.float32)
    init_from_modules(fp, R, [21, 22], "mlp", W)
    net = StackedNet(kind, R, fp, "mlp", W)
    assert net.layers == 0 and net.in_width == (2 * R if kind == "policy" else W)
    x = torch.randn(G, T, B, net.in_width)
    st = net.initial_state(B)
    out, st2 = net.forward(x, st, None)
    assert st2[0].numel() == 0                                        # no recurrent state
    fp.grad.zero_()
    (out ** 2).sum().backward()
    for g in range(G):
        m = Policy(R) if kind == "policy" else Value(W)
        m.load_state_dict(agent_state_dict(fp, g)[kind])             # the reference modules' parameter names: features_extractor.*, net.*
        o = m(x[g].reshape(T * B, -1)).reshape(T, B, -1)
        assert torch.allclose(o, out[g], atol=1e-5)
        (o ** 2).sum().backward()
        for n, q in m.named_parameters():
            off, k, shp = fp.offsets[f"{kind}.{n}"]
            assert torch.allclose(fp.grad[g, off:off + k].view(shp), q.grad, rtol=1e-4, atol=1e-6 * float(q.grad.abs().max() + 1)), n
    sel = torch.tensor([2, 0])
    out_sel, _ = net.forward(x, net.initial_state(2), None, select=sel)      # a PPO minibatch: sequences `sel` of the buffer
    assert torch.allclose(out_sel, out[:, :, sel].detach(), atol=1e-6)


def test_trainer_runs_the_non_recurrent_pair(tmp_path):
    """MAPPOTrainer with recurrent=False: the critics read the whole flattened shared state (packing.pack_value_input), updates move
    the parameters, and the checkpoint carries the reference modules' names and loads into models.Policy / Value."""
    from as_cops_and_thieves_amd import packing
    from as_cops_and_thieves_amd.selfplay.models import Policy, Value, state_width
    env = _env()
    rc = RoleConfig(learning_epochs=1, mini_batches=2, random_timesteps=0, learning_starts=0, kl_threshold=0.0)
    tr = MAPPOTrainer(env, {"cop": rc, "thief": rc}, TrainerConfig(horizon=4, bptt=4, timesteps=8, policy_freeze_duration=0,
                                                                  opponent_freeze_duration=0, recurrent=False), seed=3)
    rl = next(iter(tr.roles.values()))
    W = state_width(len(env.possible_agents), env.nc, 16)
    assert rl.arch == "mlp" and rl.buf["vin"].shape[-1] == W == packing.pack_value_input(env.state()).shape[1]
    before = tr.param_digest()
    stats = tr.train(8)
    assert tr.param_digest() != before and all(np.isfinite(v) for v in stats.values())
    sd = tr.state_dict()
    assert sd["__cat__"]["recurrent"] is False
    assert sorted(sd["cop_0"]["value"]) == sorted(Value(W).state_dict()) and sorted(sd["cop_0"]["policy"]) == sorted(Policy(16).state_dict())
    pol, val = Policy(16), Value(W)
    pol.load_state_dict(sd["thief_0"]["policy"]); val.load_state_dict(sd["thief_0"]["value"])
    obs, _ = env.reset()
    pin, vin = tr._inputs(rl, obs, env.state())
    g = rl.agents.index("thief_0")
    logits, _ = rl.policy.forward(pin.unsqueeze(1), rl.p_state, None)
    values, _ = rl.value.forward(vin.unsqueeze(1), rl.v_state, None)
    assert torch.allclose(logits[g, 0], pol(pin[g]), atol=1e-5) and torch.allclose(values[g, 0], val(vin[g]), atol=1e-5)
    torch.save(sd, tmp_path / "ck.pt")
    tr2 = MAPPOTrainer(_env(), {"cop": rc, "thief": rc}, TrainerConfig(horizon=4, bptt=4, recurrent=False), seed=9)
    tr2.load_state_dict(torch.load(tmp_path / "ck.pt", weights_only=True))
    assert tr2.param_digest() == tr.param_digest()
    with pytest.raises(KeyError):                                      # a recurrent trainer does not take a non-recurrent checkpoint
        MAPPOTrainer(_env(), {"cop": rc, "thief": rc}, TrainerConfig(horizon=4, bptt=4), seed=9).load_state_dict(sd)
