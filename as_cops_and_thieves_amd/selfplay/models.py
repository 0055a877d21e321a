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
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
import torch.nn as nn


def conv_out_len(num_rays: int) -> int:
    l1 = (num_rays - 5) // 2 + 1
    return (l1 - 5) // 3 + 1


class _Trunk(nn.Module):
    def __init__(self, channels: int, num_rays: int, hidden: int, layers: int):
        super().__init__()
        self.channels, self.num_rays = channels, num_rays
        self.features = nn.Sequential(
            nn.Conv1d(channels, 64, kernel_size=5, stride=2), nn.ReLU(),
            nn.Conv1d(64, 32, kernel_size=5, stride=3), nn.ReLU(),
            nn.Flatten(), nn.Linear(32 * conv_out_len(num_rays), 256), nn.Tanh())
        self.lstm = nn.LSTM(256, hidden, num_layers=layers, batch_first=True)
        self.hidden, self.layers = hidden, layers

    def initial_state(self, batch: int, device, dtype=torch.float32) -> Tuple[torch.Tensor, torch.Tensor]:
        z = torch.zeros(self.layers, batch, self.hidden, device=device, dtype=dtype)
        return z, z.clone()

    def forward(self, x: torch.Tensor, state, starts: Optional[torch.Tensor] = None):
        """x: [B, T, channels*R]; state: (h, c) each [layers, B, hidden]; starts: [B, T] bool, True where
        a new episode begins at that step (state is zeroed before consuming it)."""
        B, T, _ = x.shape
        f = self.features(x.reshape(B * T, self.channels, self.num_rays)).reshape(B, T, 256)
        if starts is None or not bool(starts.any()):
            out, state = self.lstm(f, state)
            return out, state
        outs = []
        h, c = state
        for t in range(T):
            keep = (~starts[:, t]).to(h.dtype).view(1, B, 1)
            h, c = h * keep, c * keep
            o, (h, c) = self.lstm(f[:, t:t + 1], (h, c))
            outs.append(o)
        return torch.cat(outs, dim=1), (h, c)


class LSTMPolicy(nn.Module):
    def __init__(self, num_rays: int, num_actions: int = 4, hidden: int = 128):
        super().__init__()
        self.trunk = _Trunk(2, num_rays, hidden, layers=1)
        self.head = nn.Sequential(nn.Linear(hidden, 128), nn.ReLU(), nn.Linear(128, 64), nn.ReLU(),
                                  nn.Linear(64, num_actions))

    def initial_state(self, batch, device, dtype=torch.float32):
        return self.trunk.initial_state(batch, device, dtype)

    def forward(self, x, state, starts=None):
        out, state = self.trunk(x, state, starts)
        return self.head(out), state            # logits [B, T, 4]


class LSTMValue(nn.Module):
    def __init__(self, num_rays: int, hidden: int = 128):
        super().__init__()
        self.trunk = _Trunk(4, num_rays, hidden, layers=2)
        self.head = nn.Sequential(nn.Linear(hidden, 256), nn.ReLU(), nn.Linear(256, 128), nn.ReLU(),
                                  nn.Linear(128, 64), nn.ReLU(), nn.Linear(64, 1))

    def initial_state(self, batch, device, dtype=torch.float32):
        return self.trunk.initial_state(batch, device, dtype)

    def forward(self, x, state, starts=None):
        out, state = self.trunk(x, state, starts)
        return self.head(out).squeeze(-1), state   # values [B, T]
