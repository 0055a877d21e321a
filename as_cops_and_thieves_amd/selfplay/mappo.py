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
    graph_rollout: bool = True        # on a GPU, capture the T-tick rollout (env ticks + all networks) in one HIP graph
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
                                               lr=self.cfg.learning_rate)
                           for a in self.agents}
        self._sync = {a: _FlatGradSync(list(self.policies[a].parameters()) + list(self.values[a].parameters()))
                      for a in self.agents}
        self._gen = g
        self.timestep = 0
        self._p_state = {a: self.policies[a].initial_state(self.N, self.device) for a in self.agents}
        self._v_state = {a: self.values[a].initial_state(self.N, self.device) for a in self.agents}
        self._obs, _ = env.reset()
        self._starts = torch.ones(self.N, dtype=torch.bool, device=self.device)   # first tick starts an episode
        self.stats: Dict[str, float] = {}
        self._buf = None                  # preallocated rollout buffers [T, N, ...] per agent
        self._graph = None                # captured rollout (GPU)
        self._eager_rollouts = 0
        self._random_phase = False

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
    def _rollout_ticks(self, buf) -> None:
        """T ticks: networks -> actions -> env.step, written into the preallocated rollout buffers ``buf``.
        Only in-place updates of persistent tensors and no host synchronisation, so the whole loop can be
        captured in a HIP graph (``torch.cuda.CUDAGraph`` is hipGraph on ROCm) and replayed per rollout."""
        N, cfg = self.N, self.cfg
        for t in range(cfg.horizon):
            state = self.env.state()
            actions = {}
            st = self._starts.view(N, 1)
            for a in self.agents:
                pin, vin = self._policy_in(self._obs, a), self._value_in(state, a)
                with self._autocast():
                    logits, p_new = self.policies[a](pin.unsqueeze(1), self._p_state[a], st)
                    val, v_new = self.values[a](vin.unsqueeze(1), self._v_state[a], st)
                for old, new in zip(self._p_state[a] + self._v_state[a], p_new + v_new):
                    old.copy_(new)
                dist = torch.distributions.Categorical(logits=logits[:, 0].float(), validate_args=False)
                if self._random_phase:
                    act = torch.randint(0, 4, (N,), generator=self._gen).to(self.device)
                else:
                    act = dist.sample()
                b = buf[a]
                b["pin"][t].copy_(pin); b["vin"][t].copy_(vin); b["act"][t].copy_(act)
                b["logp"][t].copy_(dist.log_prob(act)); b["val"][t].copy_(val[:, 0].float()); b["start"][t].copy_(self._starts)
                actions[a] = act.to(torch.int32)
            self._obs, rewards, terms, truncs, infos = self.env.step(actions)
            done = terms[self.agents[0]]
            for a in self.agents:
                buf[a]["rew"][t].copy_(rewards[a].float()); buf[a]["done"][t].copy_(done)
            self._starts.copy_(done)               # the env auto-resets: the next tick starts a new episode

    def _alloc_rollout(self):
        T, N, R, dev = self.cfg.horizon, self.N, self.R, self.device
        f32 = dict(dtype=torch.float32, device=dev)
        return {a: {"pin": torch.empty((T, N, 2 * R), **f32), "vin": torch.empty((T, N, 4 * R), **f32),
                    "act": torch.empty((T, N), dtype=torch.long, device=dev), "logp": torch.empty((T, N), **f32),
                    "val": torch.empty((T, N), **f32), "rew": torch.empty((T, N), **f32),
                    "done": torch.empty((T, N), dtype=torch.bool, device=dev),
                    "start": torch.empty((T, N), dtype=torch.bool, device=dev)} for a in self.agents}

    @torch.no_grad()
    def collect(self) -> Dict[str, Dict[str, torch.Tensor]]:
        cfg, N = self.cfg, self.N
        if self._buf is None:
            self._buf = self._alloc_rollout()
        buf = self._buf
        p0 = {a: tuple(s.clone() for s in self._p_state[a]) for a in self.agents}
        v0 = {a: tuple(s.clone() for s in self._v_state[a]) for a in self.agents}
        self._random_phase = self.timestep < cfg.random_timesteps
        use_graph = cfg.graph_rollout and self.device.type == "cuda" and not self._random_phase
        if not use_graph:
            self._rollout_ticks(buf)
            self._eager_rollouts += 1
        else:
            if self._graph is None:
                if self._eager_rollouts == 0:       # the first rollout runs eagerly: warms allocator and libraries up
                    self._rollout_ticks(buf)
                    self._eager_rollouts += 1
                    return self._finish_rollout(buf, p0, v0)
                torch.cuda.synchronize(self.device)
                self._graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self._graph):
                    self._rollout_ticks(buf)
            self._graph.replay()
        return self._finish_rollout(buf, p0, v0)

    @torch.no_grad()
    def _finish_rollout(self, buf, p0, v0):
        cfg, N = self.cfg, self.N
        self.timestep += cfg.horizon
        state = self.env.state()
        out = {}
        for a in self.agents:
            with self._autocast():
                last_val, _ = self.values[a](self._value_in(state, a).unsqueeze(1),
                                             tuple(s.clone() for s in self._v_state[a]), self._starts.view(N, 1))
            b = {k: v.clone() for k, v in buf[a].items()}
            adv, ret = compute_gae(b["rew"], b["val"], b["done"], last_val[:, 0].float() * (~self._starts).float(),
                                   cfg.discount_factor, cfg.gae_lambda)
            b.update(adv=adv, ret=ret, p0=p0[a], v0=v0[a])
            out[a] = b
        return out

    # ------------------------------------------------------------------ update
    def _minibatch_step(self, a: str, mb: Dict[str, torch.Tensor], train_policy: bool):
        """Forward, PPO losses, backward, gradient all-reduce, clip, Adam step on one minibatch of sequences.
        Returns (policy_loss, value_loss, kl) as 0-d tensors; no host synchronisation.  (Capturing this step in a
        HIP graph was tried and dropped: ending the capture of the backward pass crashed inside the runtime in some
        process states.  The rollout, which has no autograd, is captured.)"""
        cfg = self.cfg
        with self._autocast():
            logits, _ = self.policies[a](mb["pin"], (mb["ph"], mb["pc"]), mb["start"])
            values, _ = self.values[a](mb["vin"], (mb["vh"], mb["vc"]), mb["start"])
        dist = torch.distributions.Categorical(logits=logits.float(), validate_args=False)
        logp = dist.log_prob(mb["act"])
        ratio = torch.exp(logp - mb["logp"])
        with torch.no_grad():
            kl = ((ratio - 1) - (logp - mb["logp"])).mean()
        surr = mb["adv"] * ratio
        surr_c = mb["adv"] * torch.clamp(ratio, 1 - cfg.ratio_clip, 1 + cfg.ratio_clip)
        policy_loss = -torch.min(surr, surr_c).mean()
        entropy_loss = -cfg.entropy_loss_scale * dist.entropy().mean()
        value_loss = cfg.value_loss_scale * nn.functional.mse_loss(values.float(), mb["ret"])
        loss = value_loss + ((policy_loss + entropy_loss) if train_policy else 0.0)
        self.optimizers[a].zero_grad(set_to_none=True)
        loss.backward()
        self._sync[a]()                                         # flat-buffer all-reduce over RCCL
        nn.utils.clip_grad_norm_(list(self.policies[a].parameters()) + list(self.values[a].parameters()),
                                 cfg.grad_norm_clip)
        self.optimizers[a].step()
        return policy_loss.detach(), value_loss.detach(), kl

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
            seq = dict(pin=tr(b["pin"]), vin=tr(b["vin"]), act=tr(b["act"]), logp=tr(b["logp"]), ret=tr(b["ret"]),
                       adv=tr(adv), start=tr(b["start"]))
            stop = False
            for epoch in range(cfg.learning_epochs):
                perm = torch.randperm(N, generator=self._gen).to(self.device)
                kls = []
                for idx in perm.chunk(cfg.mini_batches):
                    mb = {k: v[idx] for k, v in seq.items()}
                    mb.update(ph=b["p0"][0][:, idx].contiguous(), pc=b["p0"][1][:, idx].contiguous(),
                              vh=b["v0"][0][:, idx].contiguous(), vc=b["v0"][1][:, idx].contiguous())
                    policy_loss, value_loss, kl = self._minibatch_step(a, mb, train_policy)
                    kls.append(kl.clone())
                    pl, vl = policy_loss.clone(), value_loss.clone()
                mean_kl = float(torch.stack(kls).mean())                      # one host sync per epoch
                if cfg.kl_threshold and mean_kl > cfg.kl_threshold:
                    stop = True                                              # skrl: early stop on mean KL of the epoch
                if stop:
                    break
            stats[f"{a}/policy_loss"], stats[f"{a}/value_loss"], stats[f"{a}/kl"] = float(pl), float(vl), mean_kl
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
