"""MAPPO on the batched env: device rollout buffer, GAE scan, PPO minibatch
update, bf16 autocast, one flat gradient all-reduce per optimiser step (RCCL over xGMI when
``torch.distributed`` is initialised with the ``nccl`` backend).

Hyper-parameters default to the reference's ``CFG_AGENT`` (``src/configs/mappo_config.py:41-50``
over skrl's MAPPO defaults: discount 0.99, lambda 0.95): rollouts 4096, 4 epochs x 4 minibatches,
lr 1e-4, ratio_clip 0.15, value_loss_scale 0.5, entropy_loss_scale 0.02, grad_norm_clip 0.5,
KL early stop 0.015.  As in skrl's MAPPO every agent owns a policy and a value network and is
optimised independently; the critic consumes the team-shared channels (``packing.py`` layout).
The reference's ``rollouts`` counts ticks of ONE env; here a rollout is ``horizon`` ticks of
``num_envs`` envs, so ``horizon * num_envs`` plays that role.
"""
from __future__ import annotations

import dataclasses
from typing import Dict, List, Optional

import torch
import torch.nn as nn

from .. import packing
from .models import LSTMPolicy, LSTMValue


@dataclasses.dataclass
class MAPPOConfig:
    horizon: int = 16                 # ticks per rollout and BPTT length (reference sequence_length = 16)
    learning_epochs: int = 4
    mini_batches: int = 4
    discount_factor: float = 0.99
    gae_lambda: float = 0.95
    learning_rate: float = 1e-4
    ratio_clip: float = 0.15
    value_loss_scale: float = 0.5
    entropy_loss_scale: float = 0.02
    grad_norm_clip: float = 0.5
    kl_threshold: float = 0.015
    random_timesteps: int = 0         # reference: 10 000 uniformly random ticks first
    learning_starts: int = 0          # reference: 15 000
    frozen_roles: tuple = ()          # e.g. ("thief",): roles whose policy is not updated (freeze schedule)
    autocast_bf16: bool = True
    reference_q11: bool = False       # True: every critic sees the alphabetically first agent's channels (quirk Q11)


def compute_gae(rewards: torch.Tensor, values: torch.Tensor, dones: torch.Tensor, last_values: torch.Tensor,
                gamma: float, lam: float):
    """Reverse scan over the time axis.  rewards/values/dones: [T, N]; last_values: [N].
    ``dones[t]`` marks that the episode ended with tick t (no bootstrap across it)."""
    T = rewards.shape[0]
    adv = torch.zeros_like(rewards)
    last = torch.zeros_like(last_values)
    nxt = last_values
    for t in range(T - 1, -1, -1):
        nd = 1.0 - dones[t].to(rewards.dtype)
        delta = rewards[t] + gamma * nxt * nd - values[t]
        last = delta + gamma * lam * nd * last
        adv[t] = last
        nxt = values[t]
    return adv, adv + values


class _FlatGradSync:
    """One fused all-reduce (sum, then / world) of all gradients of a parameter set."""

    def __init__(self, params: List[nn.Parameter]):
        self.params = [p for p in params if p.requires_grad]

    def __call__(self) -> None:
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
            return
        grads = [p.grad if p.grad is not None else torch.zeros_like(p) for p in self.params]
        flat = torch.cat([g.reshape(-1) for g in grads])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        flat /= dist.get_world_size()
        off = 0
        for p, g in zip(self.params, grads):
            n = g.numel()
            p.grad = flat[off:off + n].view_as(g).clone()
            off += n


class MAPPOTrainer:
    def __init__(self, env, cfg: Optional[MAPPOConfig] = None, device=None, seed: int = 0):
        self.env, self.cfg = env, cfg or MAPPOConfig()
        self.device = torch.device(device) if device is not None else getattr(env, "device", torch.device("cpu"))
        self.agents: List[str] = list(env.possible_agents)
        self.N = env.num_envs
        self.R = env.observation_spaces[self.agents[0]]["distance"].shape[0]
        g = torch.Generator(device="cpu").manual_seed(seed)
        torch.manual_seed(seed)
        self.policies = {a: LSTMPolicy(self.R).to(self.device) for a in self.agents}
        self.values = {a: LSTMValue(self.R).to(self.device) for a in self.agents}
        self.optimizers = {a: torch.optim.Adam(list(self.policies[a].parameters()) + list(self.values[a].parameters()),
                                               lr=self.cfg.learning_rate) for a in self.agents}
        self._sync = {a: _FlatGradSync(list(self.policies[a].parameters()) + list(self.values[a].parameters()))
                      for a in self.agents}
        self._gen = g
        self.timestep = 0
        self._p_state = {a: self.policies[a].initial_state(self.N, self.device) for a in self.agents}
        self._v_state = {a: self.values[a].initial_state(self.N, self.device) for a in self.agents}
        self._obs, _ = env.reset()
        self._starts = torch.ones(self.N, dtype=torch.bool, device=self.device)   # first tick starts an episode
        self.stats: Dict[str, float] = {}

    # ------------------------------------------------------------------ inputs
    def _policy_in(self, obs, a):
        return packing.pack_policy_input(obs[a])                             # [N, 2R] f32

    def _value_in(self, state, a):
        src = sorted(state)[0] if self.cfg.reference_q11 else a
        return packing.pack_agent_state(state[src])[:, : 4 * self.R]          # the 4 ray channels

    def _autocast(self):
        on = self.cfg.autocast_bf16 and self.device.type == "cuda"
        return torch.autocast(device_type=self.device.type, dtype=torch.bfloat16, enabled=on)

    # ------------------------------------------------------------------ rollout
    @torch.no_grad()
    def collect(self) -> Dict[str, Dict[str, torch.Tensor]]:
        T, N, cfg = self.cfg.horizon, self.N, self.cfg
        buf = {a: {k: [] for k in ("pin", "vin", "act", "logp", "val", "rew", "done", "start")} for a in self.agents}
        p0 = {a: tuple(s.clone() for s in self._p_state[a]) for a in self.agents}
        v0 = {a: tuple(s.clone() for s in self._v_state[a]) for a in self.agents}
        for _ in range(T):
            state = self.env.state()
            actions = {}
            for a in self.agents:
                pin, vin = self._policy_in(self._obs, a), self._value_in(state, a)
                st = self._starts.view(N, 1)
                with self._autocast():
                    logits, self._p_state[a] = self.policies[a](pin.unsqueeze(1), self._p_state[a], st)
                    val, self._v_state[a] = self.values[a](vin.unsqueeze(1), self._v_state[a], st)
                dist = torch.distributions.Categorical(logits=logits[:, 0].float())
                if self.timestep < cfg.random_timesteps:
                    act = torch.randint(0, 4, (N,), generator=self._gen).to(self.device)
                else:
                    act = dist.sample()
                b = buf[a]
                b["pin"].append(pin); b["vin"].append(vin); b["act"].append(act)
                b["logp"].append(dist.log_prob(act)); b["val"].append(val[:, 0].float()); b["start"].append(self._starts.clone())
                actions[a] = act.to(torch.int32)
            self._obs, rewards, terms, truncs, infos = self.env.step(actions)
            done = terms[self.agents[0]].clone()
            for a in self.agents:
                buf[a]["rew"].append(rewards[a].float().clone()); buf[a]["done"].append(done)
            self._starts = done.clone()            # the env auto-resets: the next tick starts a new episode
            self.timestep += 1
        state = self.env.state()
        out = {}
        for a in self.agents:
            with self._autocast():
                last_val, _ = self.values[a](self._value_in(state, a).unsqueeze(1),
                                             tuple(s.clone() for s in self._v_state[a]), self._starts.view(N, 1))
            b = {k: torch.stack(v) for k, v in buf[a].items()}
            adv, ret = compute_gae(b["rew"], b["val"], b["done"], last_val[:, 0].float() * (~self._starts).float(),
                                   cfg.discount_factor, cfg.gae_lambda)
            b.update(adv=adv, ret=ret, p0=p0[a], v0=v0[a])
            out[a] = b
        return out

    # ------------------------------------------------------------------ update
    def update(self, rollout) -> Dict[str, float]:
        cfg, T, N = self.cfg, self.cfg.horizon, self.N
        stats = {}
        for a in self.agents:
            b = rollout[a]
            role = a.split("_")[0]
            train_policy = role not in cfg.frozen_roles
            adv = (b["adv"] - b["adv"].mean()) / (b["adv"].std() + 1e-8)
            # sequences = env slots (each a length-T BPTT window starting from the stored recurrent state)
            tr = lambda x: x.transpose(0, 1).contiguous()                    # [T,N,..] -> [N,T,..]
            pin, vin, act, logp_old, val_old, ret, advn, start = map(
                tr, (b["pin"], b["vin"], b["act"], b["logp"], b["val"], b["ret"], adv, b["start"]))
            stop = False
            for epoch in range(cfg.learning_epochs):
                perm = torch.randperm(N, generator=self._gen).to(self.device)
                kls = []
                for idx in perm.chunk(cfg.mini_batches):
                    p_state = tuple(s[:, idx].contiguous() for s in b["p0"])
                    v_state = tuple(s[:, idx].contiguous() for s in b["v0"])
                    with self._autocast():
                        logits, _ = self.policies[a](pin[idx], p_state, start[idx])
                        values, _ = self.values[a](vin[idx], v_state, start[idx])
                    dist = torch.distributions.Categorical(logits=logits.float())
                    logp = dist.log_prob(act[idx])
                    ratio = torch.exp(logp - logp_old[idx])
                    with torch.no_grad():
                        kl = ((ratio - 1) - (logp - logp_old[idx])).mean()
                        kls.append(kl)
                    surr = advn[idx] * ratio
                    surr_c = advn[idx] * torch.clamp(ratio, 1 - cfg.ratio_clip, 1 + cfg.ratio_clip)
                    policy_loss = -torch.min(surr, surr_c).mean()
                    entropy_loss = -cfg.entropy_loss_scale * dist.entropy().mean()
                    value_loss = cfg.value_loss_scale * nn.functional.mse_loss(values.float(), ret[idx])
                    loss = value_loss + ((policy_loss + entropy_loss) if train_policy else 0.0)
                    self.optimizers[a].zero_grad(set_to_none=True)
                    loss.backward()
                    self._sync[a]()                                         # flat-buffer all-reduce over RCCL
                    nn.utils.clip_grad_norm_(list(self.policies[a].parameters()) + list(self.values[a].parameters()),
                                             cfg.grad_norm_clip)
                    self.optimizers[a].step()
                    stats[f"{a}/policy_loss"] = float(policy_loss.detach())
                    stats[f"{a}/value_loss"] = float(value_loss.detach())
                if cfg.kl_threshold and float(torch.stack(kls).mean()) > cfg.kl_threshold:
                    stop = True                                              # skrl: early stop on mean KL of the epoch
                if stop:
                    break
            stats[f"{a}/kl"] = float(torch.stack(kls).mean())
        self.stats = stats
        return stats

    def train(self, iterations: int) -> Dict[str, float]:
        for _ in range(iterations):
            rollout = self.collect()
            if self.timestep >= self.cfg.learning_starts:
                self.update(rollout)
        return self.stats

    # ------------------------------------------------------------------ checkpoints (per-role files for the archive)
    def role_state_dict(self, role: str) -> dict:
        return {a: {"policy": self.policies[a].state_dict(), "value": self.values[a].state_dict()}
                for a in self.agents if a.startswith(role)}

    def load_role_state_dict(self, sd: dict) -> None:
        for a, parts in sd.items():
            self.policies[a].load_state_dict(parts["policy"])
            self.values[a].load_state_dict(parts["value"])
