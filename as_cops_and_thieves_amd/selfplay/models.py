"""Policy / value networks (PyTorch-ROCm), shaped like the reference's
``src/models/lstm_policy_net.py:28-53`` and ``lstm_value_net.py:46-75``:

    policy: Conv1d(2->64,k5,s2) ReLU Conv1d(64->32,k5,s3) ReLU Flatten Linear(->256) Tanh
            LSTM(256->128, 1 layer)  128->128->64->4
    value : same trunk with 4 input channels, LSTM(256->128, 2 layers)  128->256->128->64->1

Layer sizes are derived from the ray count (the reference hard-codes 90 rays: ``Linear(32*13, 256)``
and ``cnn_channel_length = 90``).  Inputs use the sorted-key channel layout of
``packing.py``.  The recurrent state is carried across the rollout and reset where an episode ends
(the reference, through skrl, re-starts from a zero state on every call during rollout — SURVEY §5
"long-context"; this is the intended behaviour of the same architecture).

Attribute names are the reference's (``features_extractor``, ``lstm``, ``policy_head`` / ``value_head``;
non-recurrent pair: ``features_extractor`` + ``net``, ``policy_net.py:17-33``, ``value_net.py:18-28``), so
``state_dict()`` keys here and in the reference's modules are the same strings and checkpoints move both ways.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
import torch.nn as nn


def conv_out_len(num_rays: int) -> int:
    l1 = (num_rays - 5) // 2 + 1
    return (l1 - 5) // 3 + 1


def conv1d_as_gemm(x: torch.Tensor, conv: nn.Conv1d) -> torch.Tensor:
    """``conv(x)`` for an unpadded, undilated Conv1d, evaluated as ONE GEMM over the whole batch: windows are
    gathered with ``unfold`` and multiplied by the [out, in*k] weight matrix (hipBLASLt).  MIOpen's choices for these
    tiny sequences -- im2col per sample, naive bf16 kernels for the weight gradient -- took > 40 % of a training
    step.  Same parameters and same result as ``nn.Conv1d``."""
    k, stride = conv.kernel_size[0], conv.stride[0]
    cols = x.unfold(2, k, stride)                                   # [B, C, L_out, k]
    B, C, L, _ = cols.shape
    cols = cols.permute(0, 2, 1, 3).reshape(B * L, C * k)
    out = nn.functional.linear(cols, conv.weight.reshape(conv.out_channels, C * k), conv.bias)
    return out.reshape(B, L, conv.out_channels).transpose(1, 2)      # [B, out, L_out]


class _Recurrent(nn.Module):
    """Conv trunk + LSTM shared by the recurrent policy and value networks; the head is the subclass's."""

    def __init__(self, channels: int, num_rays: int, hidden: int, layers: int):
        super().__init__()
        self.channels, self.num_rays = channels, num_rays
        self.features_extractor = nn.Sequential(
            nn.Conv1d(channels, 64, kernel_size=5, stride=2), nn.ReLU(),
            nn.Conv1d(64, 32, kernel_size=5, stride=3), nn.ReLU(),
            nn.Flatten(), nn.Linear(32 * conv_out_len(num_rays), 256), nn.Tanh())
        self.lstm = nn.LSTM(256, hidden, num_layers=layers, batch_first=True)
        self.hidden, self.layers = hidden, layers

    def initial_state(self, batch: int, device, dtype=torch.float32) -> Tuple[torch.Tensor, torch.Tensor]:
        z = torch.zeros(self.layers, batch, self.hidden, device=device, dtype=dtype)
        return z, z.clone()

    def recur(self, x: torch.Tensor, state, starts: Optional[torch.Tensor] = None):
        """x: [B, T, channels*R]; state: (h, c) each [layers, B, hidden]; starts: [B, T] bool, True where
        a new episode begins at that step (state is zeroed before consuming it).

        The recurrence is evaluated here, step by step, from ``self.lstm``'s own parameters (``nn.LSTM`` keeps
        the reference's parameter names and checkpoint layout) with the fused LSTM cell, and episode starts are
        a multiplication by a mask -- no host synchronisation, no per-step library RNN call, and the same
        launch sequence every time (so a rollout can be captured in a HIP graph)."""
        B, T, _ = x.shape
        fe = self.features_extractor
        z = x.reshape(B * T, self.channels, self.num_rays)
        z = torch.relu(conv1d_as_gemm(z, fe[0]))
        z = torch.relu(conv1d_as_gemm(z, fe[2]))
        f = torch.tanh(fe[5](z.flatten(1))).reshape(B, T, 256)      # == self.features_extractor(...)
        h, c = state
        keep = None if starts is None else (~starts).unsqueeze(-1)                     # [B, T, 1]
        hs, cs = list(h.unbind(0)), list(c.unbind(0))
        inp = f
        for layer in range(self.layers):
            w_ih, w_hh = getattr(self.lstm, f"weight_ih_l{layer}"), getattr(self.lstm, f"weight_hh_l{layer}")
            b_ih, b_hh = getattr(self.lstm, f"bias_ih_l{layer}"), getattr(self.lstm, f"bias_hh_l{layer}")
            hl, cl = hs[layer], cs[layer]
            outs = []
            for t in range(T):
                if keep is not None:
                    # every layer's state restarts with the episode (nn.LSTM semantics with a zeroed state)
                    hl, cl = hl * keep[:, t].to(hl.dtype), cl * keep[:, t].to(cl.dtype)
                # two GEMMs + one fused gate kernel (the cell nn.LSTMCell uses); gate order i, f, g, o
                hl, cl = torch._VF.lstm_cell(inp[:, t], (hl.to(inp.dtype), cl.to(inp.dtype)), w_ih, w_hh, b_ih, b_hh)
                outs.append(hl)
            hs[layer], cs[layer] = hl, cl
            inp = torch.stack(outs, dim=1)
        return inp, (torch.stack(hs, 0), torch.stack(cs, 0))


class LSTMPolicy(_Recurrent):
    """``src/models/lstm_policy_net.py:28-53``."""

    def __init__(self, num_rays: int, num_actions: int = 4, hidden: int = 128):
        super().__init__(2, num_rays, hidden, layers=1)
        self.policy_head = nn.Sequential(nn.Linear(hidden, 128), nn.ReLU(), nn.Linear(128, 64), nn.ReLU(),
                                         nn.Linear(64, num_actions))

    def forward(self, x, state, starts=None):
        out, state = self.recur(x, state, starts)
        return self.policy_head(out), state            # logits [B, T, 4]


class LSTMValue(_Recurrent):
    """``src/models/lstm_value_net.py:46-75``."""

    def __init__(self, num_rays: int, hidden: int = 128):
        super().__init__(4, num_rays, hidden, layers=2)
        self.value_head = nn.Sequential(nn.Linear(hidden, 256), nn.ReLU(), nn.Linear(256, 128), nn.ReLU(),
                                        nn.Linear(128, 64), nn.ReLU(), nn.Linear(64, 1))

    def forward(self, x, state, starts=None):
        out, state = self.recur(x, state, starts)
        return self.value_head(out).squeeze(-1), state   # values [B, T]


class Policy(nn.Module):
    """The reference's non-recurrent policy (``src/models/policy_net.py:9-45``): the same conv trunk, then ``net`` 256->128->64->4.
    Input ``[B, 2R]`` in the sorted-key layout of ``packing.pack_policy_input``; output: logits ``[B, 4]``."""

    def __init__(self, num_rays: int, num_actions: int = 4):
        super().__init__()
        self.num_rays = num_rays
        self.features_extractor = nn.Sequential(
            nn.Conv1d(2, 64, kernel_size=5, stride=2), nn.ReLU(),
            nn.Conv1d(64, 32, kernel_size=5, stride=3), nn.ReLU(),
            nn.Flatten(), nn.Linear(32 * conv_out_len(num_rays), 256), nn.Tanh())
        self.net = nn.Sequential(nn.Linear(256, 128), nn.ReLU(), nn.Linear(128, 64), nn.ReLU(), nn.Linear(64, num_actions))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self.net(self.features_extractor(x.view(x.size(0), 2, self.num_rays)))


class Value(nn.Module):
    """The reference's non-recurrent critic (``src/models/value_net.py:8-34``): an MLP over the WHOLE flattened shared state
    (``packing.pack_value_input``: every agent's four ray channels and team positions), 512-256-128-64-1."""

    def __init__(self, num_observations: int):
        super().__init__()
        self.net = nn.Sequential(nn.Linear(num_observations, 512), nn.ReLU(), nn.Linear(512, 256), nn.ReLU(),
                                 nn.Linear(256, 128), nn.ReLU(), nn.Linear(128, 64), nn.ReLU(), nn.Linear(64, 1))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self.net(x)


def state_width(num_agents: int, n_cops: int, num_rays: int) -> int:
    """Width of ``packing.pack_value_input``: per agent four ray channels and the f16 positions of its team (x, y each)."""
    n_thieves = num_agents - n_cops
    return sum(4 * num_rays + 2 * (n_cops if i < n_cops else n_thieves) for i in range(num_agents))


def initialize_models_for_mappo(possible_agents, num_rays: int, n_cops: int, device="cpu") -> dict:
    """``src/utils/model_utils.py:45-77``: ``{agent: {"policy": Policy, "value": Value}}`` (no torch.compile: the reference
    wraps them when available, which changes no parameter)."""
    width = state_width(len(possible_agents), n_cops, num_rays)
    return {a: {"policy": Policy(num_rays).to(device), "value": Value(width).to(device)} for a in possible_agents}


def initialize_lstm_models_for_mappo(possible_agents, num_rays: int, device="cpu") -> dict:
    """``src/utils/model_utils.py:80-121``: the recurrent pair the reference's drivers use (orchestration.py:52,121)."""
    return {a: {"policy": LSTMPolicy(num_rays).to(device), "value": LSTMValue(num_rays).to(device)} for a in possible_agents}
