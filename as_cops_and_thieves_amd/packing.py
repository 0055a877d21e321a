"""Observation -> model-input packing.

skrl flattens ``Dict`` spaces in SORTED-KEY order and the reference's networks depend on it:

* policy input = ``[distance(R) | object_type(R)]`` viewed as ``(B, 2, R)``
  (``src/models/lstm_policy_net.py:101-103``);
* critic input = the flattened shared state of ALL agents (agents sorted by id, each agent's keys
  sorted: ``distance_shared, object_type_shared, own_distances, own_obj_types, team_positions``);
  ``LSTMValue`` slices the first ``4*R`` entries, i.e. the first agent's four ray channels
  (``src/models/lstm_value_net.py:122-137``).

These helpers build exactly those layouts from the batched env's tensors (float32, like skrl's
``flatten_tensorized_space``).
"""
from __future__ import annotations

from typing import Dict

import torch

STATE_KEYS_SORTED = ("distance_shared", "object_type_shared", "own_distances", "own_obj_types", "team_positions")


def pack_policy_input(obs_agent: Dict[str, torch.Tensor]) -> torch.Tensor:
    """``{"distance": [N,R] f16, "object_type": [N,R] u8}`` -> ``[N, 2R]`` f32 (distance first)."""
    return torch.cat([obs_agent["distance"].float(), obs_agent["object_type"].float()], dim=-1)


def policy_view(flat: torch.Tensor, num_rays: int) -> torch.Tensor:
    """``[N, 2R]`` -> ``(N, 2, R)`` as ``lstm_policy_net.py:103`` views it."""
    return flat.view(flat.shape[0], 2, num_rays)


def pack_agent_state(state_agent: Dict[str, torch.Tensor]) -> torch.Tensor:
    """One agent's shared-state dict -> ``[N, 4R + 2T]`` in sorted-key order."""
    n = state_agent["own_distances"].shape[0]
    return torch.cat([state_agent[k].float().reshape(n, -1) for k in STATE_KEYS_SORTED], dim=-1)


def pack_value_input(state: Dict[str, Dict[str, torch.Tensor]]) -> torch.Tensor:
    """Full critic input: agents in sorted-id order, each packed by :func:`pack_agent_state`."""
    return torch.cat([pack_agent_state(state[a]) for a in sorted(state)], dim=-1)


def value_view(flat: torch.Tensor, num_rays: int) -> torch.Tensor:
    """The slice ``LSTMValue`` consumes: first ``4R`` entries viewed as ``(N, 4, R)``."""
    return flat[:, : 4 * num_rays].view(flat.shape[0], 4, num_rays)
