"""MAPPO on the batched env.

What the reference runs (``src/self_play_driver.py`` -> ``training/orchestration.py`` ->
``utils/agent_learning_utils.py:172-230`` -> skrl ``MAPPO`` + ``SequentialTrainer``) and what this module
restates, on the device env:

* every agent owns a policy and a value network and is optimised independently (skrl MAPPO); here the agents of
  one ROLE share one stacked network evaluation (``stacked.py``) -- same per-agent parameters, one launch;
* per-role hyper-parameters ``CFG_AGENT`` / ``CFG_AGENT_COP`` / ``CFG_AGENT_THIEF``
  (``src/configs/mappo_config.py:5-50``; the driver passes ``CFG_AGENT`` for both roles, which is the default);
* the timestep-driven schedule of one ``trainer.train()`` call: ``random_timesteps`` uniformly random actions,
  no update before ``learning_starts`` (``mappo_config.py:9-10``), all policies frozen until
  ``policy_freeze_duration`` and the value networks released at ``opponent_freeze_duration`` (``CFG_TRAINER``,
  ``mappo_config.py:52-63``; the reference implements the two durations by a local patch to skrl's trainers,
  ``README.md:58-154``).  A timestep is one tick of the env batch, as in the reference's single-env loop;
* PPO as skrl writes it [SKRL-RECALL: skrl is not installed here]: GAE(0.99, 0.95) with per-agent advantage
  normalisation, clipped surrogate, entropy bonus, scaled MSE value loss, one joint gradient-norm clip over the
  agent's policy + value parameters, Adam, and the per-minibatch KL early stop ("break": the rest of the epoch's
  minibatches of THAT agent are skipped, the next epoch runs again).

MI355X shape of the update: one flat fp32 master buffer per role, bf16 compute copy, gradients accumulate into one
flat buffer, ONE all-reduce per optimiser step (gradients and the KL statistics in the same buffer, so every rank
takes the same early-stop decision), per-agent clip and a masked Adam as a few flat kernels, KL / freeze decisions
as device-side gates -- no host synchronisation inside an update, so the minibatch step is captured in HIP graphs.
"""
from __future__ import annotations

import dataclasses
import warnings
from typing import Sequence, Dict, List, Optional, Tuple

import torch

from .. import packing
from .. import _learn_native
from .models import state_width
from .stacked import (FlatParams, StackedNet, adam_state_dict, agent_state_dict, init_from_modules, load_adam_state_dict,
                      load_agent_state_dict,
                      role_param_shapes)


@dataclasses.dataclass
class RoleConfig:
    """skrl MAPPO agent configuration of one role (``src/configs/mappo_config.py:5-50``)."""
    rollouts: int = 4096                   # mappo_config.py:8: ticks of the (single) env per update.  The trainer's rollout length is
                                           # ``TrainerConfig.horizon`` (default 128 ticks of thousands of envs); pass horizon=cfg.rollouts
                                           # for the reference's own setting (tests/test_gpu_mappo.py runs it at 32 envs)
    learning_epochs: int = 4
    mini_batches: int = 4
    discount_factor: float = 0.99          # skrl MAPPO_DEFAULT_CONFIG
    gae_lambda: float = 0.95               # skrl MAPPO_DEFAULT_CONFIG ("lambda")
    learning_rate: float = 1e-4
    ratio_clip: float = 0.15
    value_loss_scale: float = 0.5
    entropy_loss_scale: float = 0.02
    grad_norm_clip: float = 0.5
    kl_threshold: float = 0.015
    random_timesteps: int = 10_000
    learning_starts: int = 15_000


CFG_AGENT = RoleConfig()                                                                       # mappo_config.py:41-50
CFG_AGENT_COP = RoleConfig()                                                                   # :19-28
CFG_AGENT_THIEF = RoleConfig(learning_epochs=3, mini_batches=8, entropy_loss_scale=0.01,      # :30-39
                             learning_rate=3e-4, ratio_clip=0.2)


@dataclasses.dataclass
class TrainerConfig:
    """``CFG_TRAINER`` (``mappo_config.py:52-63``) plus the build-side knobs."""
    timesteps: int = 100_000               # TrainingConfig.training_timesteps_per_role_training
    opponent_freeze_duration: int = 15_000
    policy_freeze_duration: int = 15_000
    horizon: int = 128                     # ticks per rollout; a multiple of ``bptt``.  With 16 (one BPTT window, the setting
                                           # of the bench line's first learner figure) the cops do not learn to catch even
                                           # random thieves in 262 M env-steps; with 128 they do (tests/test_gpu_mappo.py)
    bptt: int = 16                         # BPTT window (reference LSTM sequence_length).  A rollout of W = horizon / bptt
                                           # windows trains on W * num_envs sequences per update: the same number of
                                           # optimiser steps over W times the data (the update's cost is dominated by
                                           # its kernel COUNT, which does not grow with W)
    compute_bf16: bool = True              # bf16 compute copy of the weights on a GPU (fp32 master + fp32 Adam)
    check_device_errors: bool = True       # read the env core's device error word once per update (raises on any flag)
    graph_rollout: bool = True             # capture the T-tick rollout (env ticks + all networks) in one HIP graph
    graph_update: bool = True              # capture the minibatch step (forward, losses, backward / clip, Adam)
    reference_q11: bool = False            # True: every critic sees the alphabetically first agent's channels (quirk Q11)
    random_action_roles: Tuple[str, ...] = ()   # roles that act uniformly at random throughout (a fixed random opponent)
    deferred_values: bool = True           # kernel path: the critics do not run tick by tick (nothing in a rollout reads
                                           # their output) but once per BPTT window after the last tick, on all its ticks
    recurrent: bool = True                 # True: the LSTM pair the reference's drivers use (orchestration.py:52,121:
                                           # initialize_lstm_models_for_mappo).  False: its non-recurrent pair (policy_net.py:9-45,
                                           # value_net.py:8-34, model_utils.py:45-77 initialize_models_for_mappo): Policy = conv trunk +
                                           # MLP on the agent's own rays, Value = an MLP over the WHOLE flattened shared state of all
                                           # agents; stacked and trained by the same learner (torch packing path, the dense kernels)
    resident_random_phase: bool = True     # while EVERY learner is inside its random_timesteps and none is due an update, the env is
                                           # advanced by the resident rollout launch (VecCopsEnv.rollout_random: cat_rollout_fused,
                                           # uniformly random Philox actions) instead of tick by tick -- skrl neither evaluates a
                                           # network nor keeps a transition of that phase (no update before learning_starts, and the
                                           # rollout memory is overwritten by then), so only the env's state after it matters
    normalize_inputs: bool = False         # False = the reference: raw distances (0..400) and type codes (0..4) go into the
                                           # convolutions (skrl's state_preprocessor is None).  True (build-side option):
                                           # distances / ray length, types / 4 -- see tools/learn_curve.py


def compute_gae(rewards: torch.Tensor, values: torch.Tensor, dones: torch.Tensor, last_values: torch.Tensor,
                gamma: float, lam: float):
    """Reverse scan over the time axis (dim -2).  rewards/values: [..., T, N]; dones: [T, N] (the episode ended with
    tick t: no bootstrap across it); last_values: [..., N]."""
    T = rewards.shape[-2]
    adv = torch.zeros_like(rewards)
    last = torch.zeros_like(last_values)
    nxt = last_values
    for t in range(T - 1, -1, -1):
        nd = 1.0 - dones[t].to(rewards.dtype)
        delta = rewards[..., t, :] + gamma * nxt * nd - values[..., t, :]
        last = delta + gamma * lam * nd * last
        adv[..., t, :] = last
        nxt = values[..., t, :]
    return adv, adv + values


def _rowsum(x: torch.Tensor) -> torch.Tensor:
    """x [G, ...] -> [G]: the sum over everything but the first axis, in stages of 256 so that every stage is a short
    inner-dimension reduction with many outputs.  Inside a captured HIP graph this replaces ``x.sum(dim=...)``: ``at::sum``
    of hundreds of thousands of elements into a few outputs is a MULTI-BLOCK reduction with a semaphore buffer, and
    replayed from a graph it returned garbage on this stack (ROCm 7.2 / torch 2.10) -- first for the bias gradients of the
    backward pass, then, at 8192 envs, for the validity check inside ``torch.multinomial`` (a spurious device-side
    assert).  Short reductions are done by one block per output and carry no such state."""
    flat = x.reshape(x.shape[0], -1)
    while flat.shape[1] > 2048:
        r = (-flat.shape[1]) % 256
        if r:
            flat = torch.nn.functional.pad(flat, (0, r))
        flat = flat.view(flat.shape[0], -1, 256).sum(-1)
    return flat.sum(-1)


def _sample(logp_all: torch.Tensor) -> torch.Tensor:
    """Categorical sample from log-probabilities [..., K] by inverse CDF on one uniform draw per row (elementwise ops and a
    K-term sum only; ``torch.multinomial``'s input validation is a large reduction, see ``_rowsum``)."""
    p = logp_all.exp()
    u = torch.rand(p.shape[:-1] + (1,), device=p.device, dtype=p.dtype)
    cdf = torch.cumsum(p, dim=-1)[..., :-1]
    return (u >= cdf).sum(dim=-1)


class _GradsOf(torch.autograd.Function):
    """A scalar whose gradient w.r.t. (logits, values) is the given pair: hooks the analytically computed gradient of
    the PPO loss (``cat_ppo_loss_grad``) into the autograd graph of the networks."""

    @staticmethod
    def forward(ctx, logits, values, d_logits, d_values):
        ctx.save_for_backward(d_logits, d_values)
        return logits.new_zeros(())

    @staticmethod
    def backward(ctx, g):
        d_logits, d_values = ctx.saved_tensors
        return d_logits, d_values, None, None


def _dist_ready() -> bool:
    """True when the optimiser step must all-reduce: a process group of more than one rank -- or of ONE rank with
    CAT_FORCE_ALLREDUCE=1, which sends the gradient buffer through the collective library anyway (the one-GPU test of the
    RCCL launch between the two step graphs)."""
    import os
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size() > 1 or os.environ.get("CAT_FORCE_ALLREDUCE") == "1"


def advantage_moments(adv: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """Per-agent mean and (unbiased) standard deviation of ``adv`` [G, T, N] over the WHOLE batch: with data-parallel ranks, over
    every rank's shard -- what one process holding all envs would normalise with (skrl: per agent, whole memory).  Two passes, as
    ``Tensor.std``: the mean first, then the squared deviations from it.  One rank: exactly ``adv.mean`` / ``adv.std``."""
    import torch.distributed as dist
    if not (_dist_ready() and dist.get_world_size() > 1):
        return adv.mean(dim=(1, 2), keepdim=True), adv.std(dim=(1, 2), keepdim=True)

    def total(t):
        if dist.get_backend() == "gloo" and t.is_cuda:
            host = t.cpu(); dist.all_reduce(host); return host.to(t.device)
        dist.all_reduce(t)
        return t
    n_local = float(adv.shape[1] * adv.shape[2])
    m1 = total(torch.stack([adv.sum(dim=(1, 2), dtype=torch.float64),
                            torch.full((adv.shape[0],), n_local, dtype=torch.float64, device=adv.device)]))
    n, mean64 = m1[1], m1[0] / m1[1]
    dev2 = total(((adv.double() - mean64.view(-1, 1, 1)) ** 2).sum(dim=(1, 2)))
    return mean64.float().view(-1, 1, 1), (dev2 / (n - 1)).sqrt().float().view(-1, 1, 1)


class RoleLearner:
    """G agents that share one ``RoleConfig`` -- the agents of a role, or of both roles when the roles are configured
    alike (the reference's driver: ``CFG_AGENT`` for everyone) -- as stacked policy + value networks over one flat
    parameter buffer, with their Adam state, rollout buffers and PPO minibatch step.  Every launch then serves all of
    them: the update is bound by the number of kernels, not by their size."""

    BETA1, BETA2, EPS = 0.9, 0.999, 1e-8    # torch.optim.Adam defaults (what skrl constructs)

    def __init__(self, role: str, agents: List[str], indices: List[int], R: int, N: int, T: int, cfg: RoleConfig,
                 device: torch.device, compute_dtype: torch.dtype, seeds: List[int], bptt: Optional[int] = None,
                 arch: str = "lstm", vwidth: Optional[int] = None):
        self.role, self.agents, self.indices, self.cfg = role, agents, indices, cfg
        self.agent_roles = [a.split("_")[0] for a in agents]
        self.G, self.R, self.N, self.T, self.device = len(agents), R, N, T, device
        self.bptt = bptt or T
        assert T % self.bptt == 0, "horizon must be a multiple of the BPTT window"
        self.W = T // self.bptt                                # windows per rollout; training sequences = W * N
        self.index_t = torch.tensor(indices, dtype=torch.long, device=device)     # columns of the [N, A] action tensor
        self.arch = arch
        self.fp = FlatParams(role_param_shapes(R, arch, vwidth), self.G, device, compute_dtype)
        init_from_modules(self.fp, R, seeds, arch, vwidth)
        self.policy, self.value = StackedNet("policy", R, self.fp, arch, vwidth), StackedNet("value", R, self.fp, arch, vwidth)
        G, P = self.G, self.fp.P
        f32 = dict(dtype=torch.float32, device=device)
        self.m, self.v, self.steps = torch.zeros(G, P, **f32), torch.zeros(G, P, **f32), torch.zeros(G, P, **f32)
        self.col_policy, self.col_value = self.fp.column_mask("policy."), self.fp.column_mask("value.")
        self.col_train = torch.ones(G, P, **f32)               # 1 where the parameter is trainable right now
        self.frozen = {r: [False, False] for r in set(self.agent_roles)}   # role -> [policy frozen, value frozen]
        self.epoch_active = torch.ones(G, **f32)               # KL early stop: 0 = skip the rest of this epoch
        self.ar = torch.zeros(G, P + 1, **f32)                 # all-reduce buffer: fp32 gradients | KL
        self.stat = torch.zeros(3, G, **f32)                   # last policy loss, value loss, KL per agent
        self._norm_scratch = torch.zeros(G, 256, **f32)
        self.native = self.device.type == "cuda" and compute_dtype == torch.bfloat16   # libcat_learn.so loss kernel
        W, S = self.W, self.W * N
        B = S // cfg.mini_batches
        if B < 1:
            raise ValueError(f"{S} training sequences ({W} BPTT windows x {N} envs on this rank) cannot fill {cfg.mini_batches} minibatches: "
                             "more envs per rank, a longer horizon or fewer minibatches")
        self.B = B
        self.idx = torch.zeros(B, dtype=torch.long, device=device)
        in_dt = dict(dtype=compute_dtype if self.native else torch.float32, device=device)   # the networks cast to it anyway
        self.buf = {"pin": torch.zeros(G, T, N, 2 * R, **in_dt), "vin": torch.zeros(G, T, N, self.value.in_width, **in_dt),
                    "act": torch.zeros(G, T, N, dtype=torch.long, device=device), "logp": torch.zeros(G, T, N, **f32),
                    "val": torch.zeros(G, T, N, **f32), "rew": torch.zeros(G, T, N, **f32),
                    "adv": torch.zeros(G, T, N, **f32), "ret": torch.zeros(G, T, N, **f32)}
        self.p_state = self.policy.initial_state(N)            # carried across rollouts
        self.v_state = self.value.initial_state(N)
        # recurrent states at the start of every window of the stored rollout: [W, layers, G, N, H]
        self.p0w = tuple(s.unsqueeze(0).repeat(W, 1, 1, 1, 1) for s in self.p_state)
        self.v0w = tuple(s.unsqueeze(0).repeat(W, 1, 1, 1, 1) for s in self.v_state)
        # what the minibatch step gathers from: [G, bptt, W * N, ...] (the rollout buffers themselves when W == 1)
        if W == 1:
            self.tb = self.buf
            self.p0, self.v0 = tuple(s[0] for s in self.p0w), tuple(s[0] for s in self.v0w)
            self.start = None                                  # set by update(): the trainer's [T, N] start flags
        else:
            self.tb = {k: torch.zeros((G, self.bptt, S) + tuple(v.shape[3:]), dtype=v.dtype, device=device)
                       for k, v in self.buf.items() if k not in ("val", "rew")}
            self.p0 = tuple(torch.zeros(s.shape[1], G, S, s.shape[4], dtype=s.dtype, device=device) for s in self.p0w)
            self.v0 = tuple(torch.zeros(s.shape[1], G, S, s.shape[4], dtype=s.dtype, device=device) for s in self.v0w)
            self.start = torch.zeros(self.bptt, S, dtype=torch.bool, device=device)
        self._graphs = None

    # ------------------------------------------------------------------ freezing (skrl Model.freeze_parameters)
    def set_frozen(self, role: Optional[str] = None, policy: Optional[bool] = None, value: Optional[bool] = None) -> None:
        for r, fr in self.frozen.items():
            if role is None or r == role:
                if policy is not None:
                    fr[0] = policy
                if value is not None:
                    fr[1] = value
        rows = [self.col_policy * (0.0 if self.frozen[r][0] else 1.0) + self.col_value * (0.0 if self.frozen[r][1] else 1.0)
                for r in self.agent_roles]
        self.col_train.copy_(torch.stack(rows))

    def rows(self, role: str) -> List[int]:
        return [g for g, r in enumerate(self.agent_roles) if r == role]

    # ------------------------------------------------------------------ the PPO minibatch step
    def _step_forward_backward(self) -> None:
        """zero grads, gather the minibatch ``self.idx`` (sequences = env slots), forward, losses, backward; leaves the
        fp32 gradients and the per-agent KL in ``self.ar``."""
        cfg, b, idx = self.cfg, self.tb, self.idx
        self.fp.grad.zero_()
        sel = lambda x: x.index_select(2, idx)
        keep = (~self.start.index_select(1, idx)).to(torch.float32)                    # [T, B]
        st = lambda s: s.index_select(2, idx)
        # the observations of the minibatch are read out of the rollout buffers in place (the fused trunk kernels take the index)
        logits, _ = self.policy.forward(b["pin"], (st(self.p0[0]), st(self.p0[1])), keep, select=idx)
        values, _ = self.value.forward(b["vin"], (st(self.v0[0]), st(self.v0[1])), keep, select=idx)
        M = float(logits.shape[1] * logits.shape[2])                                     # samples per agent in the minibatch
        if self.native:   # loss, statistics and d loss / d (logits, values) in one launch (csrc/cat_ppo.hip)
            sums, d_logits, d_values = _learn_native.ppo_loss_grad(
                logits, values, sel(b["act"]), sel(b["logp"]), sel(b["adv"]), sel(b["ret"]), cfg.ratio_clip, cfg.value_loss_scale,
                cfg.entropy_loss_scale)
            policy_loss, value_loss, kl = -sums[:, 0] / M, cfg.value_loss_scale * sums[:, 1] / M, sums[:, 3] / M
            _GradsOf.apply(logits, values, d_logits, d_values).backward()
        else:
            logp_all = torch.log_softmax(logits.float(), dim=-1)                        # [G, T, B, 4]
            logp = logp_all.gather(-1, sel(b["act"]).unsqueeze(-1)).squeeze(-1)
            old = sel(b["logp"])
            ratio = torch.exp(logp - old)
            with torch.no_grad():
                kl = _rowsum((ratio - 1) - (logp - old)) / M                            # [G]
            adv = sel(b["adv"])
            surr = torch.min(adv * ratio, adv * torch.clamp(ratio, 1 - cfg.ratio_clip, 1 + cfg.ratio_clip))
            policy_loss = -_rowsum(surr) / M
            entropy = -_rowsum((logp_all.exp() * logp_all).sum(-1)) / M
            value_loss = cfg.value_loss_scale * _rowsum((values.float().squeeze(-1) - sel(b["ret"])) ** 2) / M
            (policy_loss - cfg.entropy_loss_scale * entropy + value_loss).sum().backward()   # agents share no parameter
        with torch.no_grad():
            self.ar[:, :-1].copy_(self.fp.grad)
            self.ar[:, -1].copy_(kl)
            self.stat[0].copy_(policy_loss); self.stat[1].copy_(value_loss)

    @torch.no_grad()
    def _step_apply(self) -> None:
        """KL gate, joint policy+value gradient-norm clip per agent, masked Adam, refresh of the compute copy."""
        cfg = self.cfg
        if self.device.type == "cuda":   # the same step in two launches (csrc/cat_ppo.hip)
            _learn_native.ppo_adam_step(self.ar, self.col_train, self.epoch_active, self.m, self.v, self.steps, self.fp.master,
                                        None if self.fp.lp is self.fp.master else self.fp.lp, self.stat[2], self._norm_scratch,
                                        cfg.learning_rate, self.BETA1, self.BETA2, self.EPS, cfg.grad_norm_clip, cfg.kl_threshold)
            return
        kl = self.ar[:, -1]
        self.stat[2].copy_(kl)
        if cfg.kl_threshold:
            self.epoch_active.mul_((kl <= cfg.kl_threshold).to(torch.float32))
        g = self.ar[:, :-1] * self.col_train                                           # frozen parameters: no gradient
        norm = _rowsum(g * g).sqrt().unsqueeze(1)
        g = g * torch.clamp(cfg.grad_norm_clip / (norm + 1e-6), max=1.0)                 # torch.nn.utils.clip_grad_norm_
        gate = self.epoch_active.unsqueeze(1) * self.col_train                         # [G, P]: 1 = this entry steps
        self.steps.add_(gate)
        self.m.add_(gate * (1 - self.BETA1) * (g - self.m))
        self.v.add_(gate * (1 - self.BETA2) * (g * g - self.v))
        s = self.steps.clamp_min(1.0)
        bc1, bc2 = 1 - self.BETA1 ** s, 1 - self.BETA2 ** s
        denom = (self.v / bc2).sqrt_().add_(self.EPS)
        self.fp.master.sub_(gate * cfg.learning_rate * (self.m / bc1) / denom)
        self.fp.refresh()

    def minibatch_step(self, use_graph: bool) -> None:
        import torch.distributed as dist
        multi = _dist_ready()
        if use_graph and self._graphs is None:
            self._graphs = self._capture()
        if use_graph and self._graphs:
            self._graphs[0].replay()
        else:
            self._step_forward_backward()
        if multi:
            dist.all_reduce(self.ar, op=dist.ReduceOp.SUM)                             # RCCL over xGMI on a GPU node
            self.ar.div_(dist.get_world_size())
        if use_graph and self._graphs:
            self._graphs[1].replay()
        else:
            self._step_apply()

    def _capture(self):
        """Capture the two halves of the step in HIP graphs (``torch.cuda.CUDAGraph``).  Warm-up runs on a side stream
        first (library workspaces, autograd buffers), as whole-network capture requires; the optimiser state the warm-up
        touched is restored, so capturing does not train.  Returns () if the runtime refuses the capture."""
        keep = [t.clone() for t in (self.fp.master, self.m, self.v, self.steps, self.epoch_active, self.ar, self.stat)]

        def restore():
            for dst, src in zip((self.fp.master, self.m, self.v, self.steps, self.epoch_active, self.ar, self.stat), keep):
                dst.copy_(src)
            self.fp.refresh()
        try:
            side = torch.cuda.Stream(self.device)
            side.wait_stream(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(side):
                for _ in range(3):
                    self._step_forward_backward()
                    self._step_apply()
            torch.cuda.current_stream(self.device).wait_stream(side)
            torch.cuda.synchronize(self.device)
            restore()
            ga, gb = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
            with torch.cuda.graph(ga):
                self._step_forward_backward()
            with torch.cuda.graph(gb):
                self._step_apply()
            torch.cuda.synchronize(self.device)
            restore()
            return ga, gb
        except Exception as exc:   # noqa: BLE001 - keep training eagerly, but say so
            warnings.warn(f"HIP-graph capture of the {self.role} PPO step failed ({exc!r}); running it eagerly")
            torch.cuda.synchronize(self.device)
            restore()
            return ()

    # ------------------------------------------------------------------ update of one rollout
    def update(self, dones: torch.Tensor, starts: torch.Tensor, last_values: torch.Tensor, gen: torch.Generator,
               use_graph: bool) -> None:
        """dones/starts [T, N]; last_values [G, N] (bootstrap, already zeroed where the next tick starts an episode)."""
        cfg, b = self.cfg, self.buf
        if self.native:   # the reverse scan over the T ticks as one launch (csrc/cat_ppo.hip) instead of ~7 small ones per tick
            adv = torch.empty_like(b["adv"])
            _learn_native.ppo_gae(b["rew"], b["val"], dones, last_values, cfg.discount_factor, cfg.gae_lambda, adv, b["ret"])
        else:
            adv, ret = compute_gae(b["rew"], b["val"], dones, last_values, cfg.discount_factor, cfg.gae_lambda)
            b["ret"].copy_(ret)
        mean, std = advantage_moments(adv)
        b["adv"].copy_((adv - mean) / (std + 1e-8))                                     # skrl: per agent, whole memory
        if self.W == 1:
            self.start = starts
        else:   # the rollout as W * N training sequences of ``bptt`` ticks: [G, W * bptt, N, ..] -> [G, bptt, W * N, ..]
            G, W, L, N = self.G, self.W, self.bptt, self.N
            for k, dst in self.tb.items():
                src = b[k]
                dst.copy_(src.view((G, W, L, N) + tuple(src.shape[3:])).transpose(1, 2).reshape(dst.shape))
            self.start.copy_(starts.view(W, L, N).transpose(0, 1).reshape(L, W * N))
            for dst, src in zip(self.p0 + self.v0, self.p0w + self.v0w):            # [W, layers, G, N, H] -> [layers, G, W * N, H]
                dst.copy_(src.permute(1, 2, 0, 3, 4).reshape(dst.shape))
        for _ in range(cfg.learning_epochs):
            self.epoch_active.fill_(1.0)
            perm = torch.randperm(self.W * self.N, generator=gen).to(self.device)
            for k in range(cfg.mini_batches):
                self.idx.copy_(perm[k * self.B:(k + 1) * self.B])
                self.minibatch_step(use_graph)


class MAPPOTrainer:
    """MAPPO over a ``VecCopsEnv`` (or any env with its surface): roles ``cop`` and ``thief``."""

    def __init__(self, env, role_cfg: Optional[Dict[str, RoleConfig]] = None, trainer_cfg: Optional[TrainerConfig] = None,
                 device=None, seed: int = 0):
        self.env, self.tcfg = env, trainer_cfg or TrainerConfig()
        self.device = torch.device(device) if device is not None else getattr(env, "device", torch.device("cpu"))
        self.agents: List[str] = list(env.possible_agents)
        self.N = env.num_envs
        self.R = env.observation_spaces[self.agents[0]]["distance"].shape[0]
        role_cfg = role_cfg or {}
        n_cops = sum(a.startswith("cop") for a in self.agents)
        self.state_width = state_width(len(self.agents), n_cops, self.R)    # flattened shared state: the non-recurrent critic's input
        on_gpu = self.device.type == "cuda"
        dt = torch.bfloat16 if (self.tcfg.compute_bf16 and on_gpu) else torch.float32
        # roles configured alike are stacked into ONE learner (key "cop+thief"); otherwise one learner per role
        cfgs = {r: dataclasses.replace(role_cfg.get(r, CFG_AGENT)) for r in ("cop", "thief")
                if any(a.startswith(r) for a in self.agents)}
        groups: List[List[str]] = []
        for r in cfgs:
            for grp in groups:
                if cfgs[grp[0]] == cfgs[r]:
                    grp.append(r)
                    break
            else:
                groups.append([r])
        self.roles: Dict[str, RoleLearner] = {}
        for grp in groups:
            names = [a for a in self.agents if a.split("_")[0] in grp]
            idx = [self.agents.index(a) for a in names]
            self.roles["+".join(grp)] = RoleLearner("+".join(grp), names, idx, self.R, self.N, self.tcfg.horizon, cfgs[grp[0]],
                                                    self.device, dt, seeds=[seed * 1000 + i for i in idx],
                                                    bptt=min(self.tcfg.bptt, self.tcfg.horizon),
                                                    arch="lstm" if self.tcfg.recurrent else "mlp", vwidth=self.state_width)
        for rl in self.roles.values():
            mask = [r in self.tcfg.random_action_roles for r in rl.agent_roles]
            rl.random_rows = torch.tensor(mask, device=self.device).view(rl.G, 1) if any(mask) else None
        R, d, t = self.R, 1.0 / 400.0, 0.25          # ray length (entity.py:176 default sensor), number of type codes - 1
        self._pin_scale = torch.tensor([d] * R + [t] * R, device=self.device)
        self._vin_scale = torch.tensor(([d] * R + [t] * R) * 2, device=self.device)
        nt = len(self.agents) - n_cops      # per agent (sorted ids: cops first): four ray channels, then its team's f16 positions (px)
        self._state_scale = torch.tensor(sum((([d] * R + [t] * R) * 2 + [1e-3] * (2 * (n_cops if i < n_cops else nt))
                                              for i in range(len(self.agents))), []), device=self.device)
        self._gen = torch.Generator(device="cpu").manual_seed(seed)
        torch.manual_seed(seed)
        self.timestep = 0
        self._obs, _ = env.reset()
        self._starts = torch.ones(self.N, dtype=torch.bool, device=self.device)
        self._keep32 = torch.zeros(1, self.N, dtype=torch.float32, device=self.device)   # = ~_starts, as the networks take it
        T = self.tcfg.horizon
        self._done_buf = torch.zeros(T, self.N, dtype=torch.bool, device=self.device)
        self._start_buf = torch.zeros(T, self.N, dtype=torch.bool, device=self.device)
        self._actions = torch.zeros(self.N, len(self.agents), dtype=torch.int32, device=self.device)
        self._native_io = on_gpu and hasattr(env, "raw_outputs") and self.tcfg.recurrent   # cat_rollout.h packs the LSTM pair's rows
        self._native_post = self._native_io and hasattr(env, "step_raw")   # cat_rollout_post: rewards and flags of the raw step   # cat_rollout.h: packing and sampling in one launch each
        self._graph = None
        self._eager_rollouts = 0
        self.stats: Dict[str, float] = {}
        self.use_graphs = on_gpu

    # ------------------------------------------------------------------ model inputs (packing.py layouts)
    def _pack_native(self, rl: RoleLearner, pin: torch.Tensor, vin: torch.Tensor) -> None:
        """The same rows as ``_inputs`` straight from the env core's buffers into bf16 ``pin`` / ``vin`` (one launch)."""
        d, t = (1.0 / 400.0, 0.25) if self.tcfg.normalize_inputs else (1.0, 1.0)
        _learn_native.rollout_pack(self.env.raw_outputs(), rl.indices, sum(a.startswith("cop") for a in self.agents),
                                   self.tcfg.reference_q11, d, t, pin, vin)

    def _inputs(self, rl: RoleLearner, obs, state) -> Tuple[torch.Tensor, torch.Tensor]:
        if self._native_io and rl.native:
            pin = torch.empty(rl.G, self.N, 2 * self.R, dtype=torch.bfloat16, device=self.device)
            vin = torch.empty(rl.G, self.N, 4 * self.R, dtype=torch.bfloat16, device=self.device)
            self._pack_native(rl, pin, vin)
            return pin, vin
        pin = torch.stack([packing.pack_policy_input(obs[a]) for a in rl.agents])                       # [G, N, 2R]
        if not self.tcfg.recurrent:      # value_net.py: every agent's critic reads the whole flattened shared state
            vin = packing.pack_value_input(state).unsqueeze(0).expand(rl.G, -1, -1)
            if self.tcfg.normalize_inputs:
                pin, vin = pin * self._pin_scale, vin * self._state_scale
            return pin, vin
        first = sorted(state)[0]
        vin = torch.stack([packing.pack_agent_state(state[first if self.tcfg.reference_q11 else a])[:, :4 * self.R]
                           for a in rl.agents])                                                         # [G, N, 4R]
        if self.tcfg.normalize_inputs:   # channel layouts: [distance | type] and [distance_shared | type_shared | own distance | own type]
            pin, vin = pin * self._pin_scale, vin * self._vin_scale
        return pin, vin

    # ------------------------------------------------------------------ rollout
    def _rollout_ticks(self, random_actions) -> None:
        """T ticks: networks -> actions -> env.step into the preallocated buffers.  In-place updates of persistent
        tensors only and no host synchronisation: the whole loop is captured in one HIP graph and replayed.
        ``random_actions``: True / False for every learner, or the set of learner keys still in their ``random_timesteps``."""
        N, T = self.N, self.tcfg.horizon
        random_of = {key: (random_actions is True or (not isinstance(random_actions, bool) and key in random_actions))
                     for key in self.roles}
        any_random = any(random_of.values())
        all_fused = self._native_post and not any_random and all(rl.native and rl.random_rows is None for rl in self.roles.values())
        defer = all_fused and self._native_io and self.tcfg.deferred_values
        for t in range(T):
            state = self.env.state()
            keep = self._keep32 if all_fused else (~self._starts).view(1, N)
            self._start_buf[t].copy_(self._starts)
            for key, rl in self.roles.items():
                random_actions = random_of[key]
                if t % rl.bptt == 0:                # the recurrent state at the start of a BPTT window is kept
                    for dst, src in zip(rl.p0w + (() if defer else rl.v0w), rl.p_state + (() if defer else rl.v_state)):
                        dst[t // rl.bptt].copy_(src)
                b = rl.buf
                fused = self._native_io and rl.native and not random_actions and rl.random_rows is None
                if fused:   # rows packed straight into the rollout buffers; the networks read them there
                    pin, vin = b["pin"][:, t], b["vin"][:, t]
                    self._pack_native(rl, pin, vin)
                else:
                    pin, vin = self._inputs(rl, self._obs, state)
                logits, p_new = rl.policy.forward(pin.unsqueeze(1), rl.p_state, keep, update_state=True)
                if defer:                        # the critic runs after the last tick (below): no tick reads its value
                    val, v_new = None, rl.v_state
                else:
                    val, v_new = rl.value.forward(vin.unsqueeze(1), rl.v_state, keep, update_state=True)
                for old, new in zip(rl.p_state + rl.v_state, p_new + v_new):
                    if old is not new:           # the kernel path has already written the new state in place
                        old.copy_(new)
                if fused:   # draw, log-probability, value and the env's action columns in one launch
                    u = torch.rand(rl.G, N, device=self.device)
                    _learn_native.rollout_sample(logits[:, 0].contiguous(), u, None if defer else val[:, 0, :, 0].contiguous(), b["act"][:, t],
                                                 b["logp"][:, t], None if defer else b["val"][:, t], self._actions, rl.indices)
                    continue
                logp_all = torch.log_softmax(logits[:, 0].float(), dim=-1)                               # [G, N, 4]
                if random_actions:
                    act = torch.randint(0, 4, (rl.G, N), generator=self._gen).to(self.device)
                else:
                    act = _sample(logp_all)
                    if rl.random_rows is not None:   # rows of the roles in TrainerConfig.random_action_roles
                        act = torch.where(rl.random_rows, torch.randint(0, 4, (rl.G, N), device=self.device), act)
                b["pin"][:, t].copy_(pin); b["vin"][:, t].copy_(vin); b["act"][:, t].copy_(act)
                b["logp"][:, t].copy_(logp_all.gather(-1, act.unsqueeze(-1)).squeeze(-1))
                b["val"][:, t].copy_(val[:, 0, :, 0].float())
                self._actions.index_copy_(1, rl.index_t, act.t().to(torch.int32))    # device index: capturable
            if all_fused:   # one launch for the tick, one for rewards + episode-end flags (cat_rollout_post)
                raw = self.env.step_raw(self._actions)
                for i, rl in enumerate(self.roles.values()):
                    flags = (self._done_buf[t], self._starts, self._keep32) if i == 0 else (None, None, None)
                    _learn_native.rollout_post(raw, rl.indices, rl.buf["rew"][:, t], *flags)
                self._obs = self.env.observations()
                continue
            self._obs, rewards, terms, truncs, infos = self.env.step(self._actions)
            done = terms[self.agents[0]]
            for rl in self.roles.values():
                rl.buf["rew"][:, t].copy_(torch.stack([rewards[a].float() for a in rl.agents]))
            self._done_buf[t].copy_(done)
            self._starts.copy_(done)               # the env auto-resets: the next tick starts a new episode
            self._keep32.copy_((~done).view(1, N))
        if defer:
            # The critics, one BPTT window at a time over the stored input rows: 16 ticks x N envs per launch instead of N, the
            # recurrence of a window in one launch per layer, and the window-start states fall out on the way.  (The cell state
            # stays fp32 inside a window, as in the training forward, where tick-by-tick it is rounded to bf16 every tick.)
            keep_all = (~self._start_buf).to(torch.float32)
            for rl in self.roles.values():
                L = rl.bptt
                for k in range(rl.W):
                    for dst, src in zip(rl.v0w, rl.v_state):
                        dst[k].copy_(src)
                    val, _ = rl.value.forward(rl.buf["vin"][:, k * L:(k + 1) * L], rl.v_state, keep_all[k * L:(k + 1) * L], update_state=True)
                    rl.buf["val"][:, k * L:(k + 1) * L].copy_(val[..., 0])

    @torch.no_grad()
    def collect(self, random_actions=False) -> None:
        """One rollout of ``horizon`` ticks into the role buffers (``random_actions``: see ``_rollout_ticks``)."""
        random_actions = random_actions if isinstance(random_actions, bool) else (frozenset(random_actions) or False)
        use_graph = self.tcfg.graph_rollout and self.use_graphs and not random_actions
        if not use_graph:
            self._rollout_ticks(random_actions)
            self._eager_rollouts += 0 if random_actions else 1
        else:
            if self._graph is None:
                if self._eager_rollouts == 0:       # the first policy-driven rollout runs eagerly: every op of the tick
                    self._rollout_ticks(False)      # (sampling included) has then run once before it is captured
                    self._eager_rollouts += 1
                    self.timestep += self.tcfg.horizon
                    return
                torch.cuda.synchronize(self.device)
                self._graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self._graph):
                    self._rollout_ticks(False)
            self._graph.replay()
        self.timestep += self.tcfg.horizon

    # ------------------------------------------------------------------ update
    def update(self, only: Optional[Sequence[str]] = None) -> Dict[str, float]:
        """PPO update of every learner (or of the learner keys in ``only``) from the rollout just collected."""
        N = self.N
        todo = {k: rl for k, rl in self.roles.items() if only is None or k in only}
        if self.tcfg.check_device_errors:
            # the env ticks of the rollout were asynchronous launches that cannot raise: read the device error word once per update (one stream
            # synchronisation, where the rollout has to be complete anyway) -- a bad action, a dropped contact or a scheduler fault stops training
            # here, as the reference's exceptions would (entity.py:126-134)
            check = getattr(self.env, "check_errors", None)   # (the CPU test doubles of the env have no device)
            if check is not None:
                check()
        with torch.no_grad():
            state = self.env.state()
            keep = (~self._starts).view(1, N)
            last = {}
            for role, rl in todo.items():
                _, vin = self._inputs(rl, self._obs, state)
                val, _ = rl.value.forward(vin.unsqueeze(1), tuple(s.clone() for s in rl.v_state), keep)
                last[role] = val[:, 0, :, 0].float() * (~self._starts).float()
        use_graph = self.tcfg.graph_update and self.use_graphs
        for role, rl in todo.items():
            rl.update(self._done_buf, self._start_buf, last[role], self._gen, use_graph)
        return {}

    def read_stats(self) -> Dict[str, float]:
        """Host copy of the last minibatch's losses (one synchronisation; call it when you want to look)."""
        out = {}
        for rl in self.roles.values():
            s = rl.stat.cpu()
            for g, a in enumerate(rl.agents):
                out[f"{a}/policy_loss"], out[f"{a}/value_loss"], out[f"{a}/kl"] = float(s[0, g]), float(s[1, g]), float(s[2, g])
        self.stats = out
        return out

    def set_frozen(self, role: Optional[str] = None, policy: Optional[bool] = None, value: Optional[bool] = None) -> None:
        for rl in self.roles.values():
            rl.set_frozen(role, policy, value)

    def learner_of(self, agent: str) -> Tuple[RoleLearner, int]:
        for rl in self.roles.values():
            if agent in rl.agents:
                return rl, rl.agents.index(agent)
        raise KeyError(agent)

    def train(self, timesteps: Optional[int] = None, freeze_policies_first: bool = True) -> Dict[str, float]:
        """One ``SequentialTrainer.train()`` of the reference's simultaneous mode (``agent_learning_utils.py:172-197``):
        the timestep restarts at 0, every policy starts frozen and every value network trainable, and the durations of
        ``CFG_TRAINER`` release them."""
        tc = self.tcfg
        timesteps = tc.timesteps if timesteps is None else timesteps
        self.timestep = 0
        if freeze_policies_first:
            self.set_frozen(policy=tc.policy_freeze_duration > 0, value=False)
        while self.timestep < timesteps:
            t0, t1 = self.timestep, self.timestep + tc.horizon
            span = self._random_phase_span(t0, timesteps)
            if span:
                t1 = t0 + span
            if tc.opponent_freeze_duration > 0 and t0 <= tc.opponent_freeze_duration < t1:
                self.set_frozen(value=False)                     # "Unfreezing opponent agent" (README.md:104-108)
            if tc.policy_freeze_duration > 0 and t0 <= tc.policy_freeze_duration < t1:
                self.set_frozen(policy=False)                    # "Unfreezing policy network" (:109-113)
            # skrl keeps these two thresholds per agent (each agent's own cfg): a learner still inside its random_timesteps
            # acts uniformly at random while another already samples its policy, and each starts updating on its own
            if span:            # every learner acts at random and nothing of these ticks is kept: one resident launch
                self._fast_forward_random(span)
                continue
            in_random = frozenset(k for k, rl in self.roles.items() if t0 < rl.cfg.random_timesteps)
            self.collect(random_actions=in_random)
            # skrl updates an agent whenever its timestep has reached ITS learning_starts, whatever its random_timesteps (a config
            # with learning_starts below random_timesteps trains on uniformly random transitions, as in the reference's stack)
            ready = [k for k, rl in self.roles.items() if self.timestep >= rl.cfg.learning_starts]
            if ready:
                self.update(only=None if len(ready) == len(self.roles) else ready)
        return self.read_stats()

    def _random_phase_span(self, t0: int, timesteps: int) -> int:
        """How many ticks from ``t0`` can run as ONE resident random-action launch: whole rollouts (thresholds act at rollout
        granularity, as in the tick-by-tick path) during which every learner is inside its ``random_timesteps`` and after which none
        is due an update.  0: take the ordinary path."""
        tc = self.tcfg
        if not (tc.resident_random_phase and hasattr(self.env, "rollout_random")) or tc.random_action_roles:
            return 0
        H = tc.horizon
        end = min(rl.cfg.random_timesteps for rl in self.roles.values())
        first_update = min(rl.cfg.learning_starts for rl in self.roles.values())
        n = 0
        # rollout n starts at t0 + n H inside EVERY learner's random phase (no network is consulted for an action) and ends before
        # any learner's learning_starts (no update follows it, so nothing of it is ever read)
        while t0 + (n + 1) * H <= timesteps and t0 + n * H < end and t0 + (n + 1) * H < first_update:
            n += 1
        return n * H

    def _fast_forward_random(self, ticks: int) -> None:
        self._synth_tick = getattr(self, "_synth_tick", 1 << 20)
        out = self.env.rollout_random(ticks, tick0=self._synth_tick)
        self._synth_tick += ticks
        self._obs = self.env.observations()
        done = out["terminated"].bool()
        self._starts.copy_(done)                      # the env auto-resets: the next tick starts a new episode there
        self._keep32.copy_((~done).view(1, self.N))
        for rl in self.roles.values():                # no network ran during the phase: the recurrent states start from zero
            for st in rl.p_state + rl.v_state:
                st.zero_()
        self.timestep += ticks

    # ------------------------------------------------------------------ checkpoints
    def agent_models(self, agent: str) -> Dict[str, Dict[str, torch.Tensor]]:
        """{"policy": sd, "value": sd} with the reference modules' parameter names (models.LSTMPolicy / LSTMValue)."""
        rl, g = self.learner_of(agent)
        return agent_state_dict(rl.fp, g)

    META_KEY = "__cat__"     # trainer position; not an agent name, so skrl's ``MAPPO.load`` (which walks possible_agents) skips it

    def state_dict(self) -> dict:
        """The "full agent" in the layout skrl's ``MAPPO.save`` writes ([SKRL-RECALL] ``{agent: {"policy": state_dict,
        "value": state_dict, "optimizer": Adam.state_dict()}}``; the reference saves it as ``joint_iter_N_full_agent.pt``,
        agent_learning_utils.py:263) with the reference modules' parameter names, plus the trainer position under
        ``__cat__``.  Everything is stored per agent, so a checkpoint does not depend on how the agents were stacked."""
        out = {}
        for a in self.agents:
            rl, g = self.learner_of(a)
            out[a] = dict(self.agent_models(a), optimizer=adam_state_dict(rl.fp, g, rl.m, rl.v, rl.steps, rl.cfg.learning_rate,
                                                                         (rl.BETA1, rl.BETA2), rl.EPS))
        out[self.META_KEY] = {"format": "cat-mappo-3", "timestep": self.timestep, "num_rays": self.R, "recurrent": bool(self.tcfg.recurrent)}
        return out

    def load_state_dict(self, sd: dict, roles: Optional[List[str]] = None, optimizer: bool = True) -> None:
        """``roles``: restrict to these roles' models (reference ``copy_role_models``, which copies policy and value
        weights only: pass ``optimizer=False`` for that).  Accepts this class's checkpoints, a checkpoint written by the
        reference (skrl layout, no ``__cat__``; preprocessor entries are ignored, the reference configures none) and
        round-2 files (``{"format": "cat-mappo-2", "models": ..., "optimizers": ...}``)."""
        if sd.get("format") == "cat-mappo-2":
            sd = dict({a: dict(sd["models"][a], **({"optimizer": sd["optimizers"][a]} if a in sd.get("optimizers", {}) else {}))
                       for a in sd["models"]}, **{self.META_KEY: {"timestep": sd.get("timestep", 0)}})
        for a in self.agents:
            if roles is not None and a.split("_")[0] not in roles:
                continue
            if a not in sd:
                raise KeyError(f"checkpoint holds no agent {a!r} (it has {sorted(k for k in sd if k != self.META_KEY)})")
            rl, g = self.learner_of(a)
            load_agent_state_dict(rl.fp, g, sd[a])
            if optimizer and "optimizer" in sd[a]:
                load_adam_state_dict(rl.fp, g, rl.m, rl.v, rl.steps, sd[a]["optimizer"])
        if optimizer and roles is None:
            self.timestep = int(sd.get(self.META_KEY, {}).get("timestep", 0))

    def param_digest(self) -> str:
        """SHA-256 over every learner's fp32 master parameters and optimiser step counts: data-parallel replicas must agree on it."""
        import hashlib
        h = hashlib.sha256()
        for k in sorted(self.roles):
            rl = self.roles[k]
            h.update(rl.fp.master.detach().float().cpu().numpy().tobytes())
            h.update(rl.steps.detach().cpu().numpy().tobytes())
        return h.hexdigest()

    def reset_optimizers(self) -> None:
        """A fresh Adam, as every self-play iteration of the reference constructs a new ``MAPPO`` (orchestration.py:135-144)."""
        for rl in self.roles.values():
            rl.m.zero_(); rl.v.zero_(); rl.steps.zero_()

    def reset_episodes(self) -> None:
        """Restart every env slot and the recurrent states (after an evaluation used the same env, or on resume)."""
        self._obs, _ = self.env.reset()
        self._starts.fill_(True)
        self._keep32.zero_()
        for rl in self.roles.values():
            for s in rl.p_state + rl.v_state:
                s.zero_()
