"""Space definitions of the team-shared observations (the MAPPO critic "state").

Mirror of reference ``src/environments/observation_spaces.py``: ``_init_shared_observation_space``
(:13-64) and ``get_nested_agent_observation_spaces`` (:134-163).  The VALUES of the shared
observations (``get_shared_observations``, :67-131) are produced by the device kernel.
"""
from __future__ import annotations

from typing import List

import numpy as np

from . import spaces


def _init_shared_observation_space(map, cops: List, thieves: List) -> spaces.Dict:
    shared = {}
    max_dim = max(map.window_dimensions)
    for team_agents in (cops, thieves):
        if not team_agents:
            continue
        example = team_agents[0].observation_space
        obj_space, dist_space = example["object_type"], example["distance"]
        team_positions_space = spaces.Box(low=0.0, high=max_dim, shape=(len(team_agents), 2), dtype=np.float16)
        team_shared_space = spaces.Dict({
            "own_obj_types": obj_space,
            "own_distances": dist_space,
            "object_type_shared": obj_space,
            "distance_shared": dist_space,
            "team_positions": team_positions_space,
        })
        for agent in team_agents:
            shared[agent.get_id()] = team_shared_space
    return spaces.Dict(shared)


def get_nested_agent_observation_spaces(shared_observation_spaces: spaces.Dict) -> spaces.Dict:
    flat = {}
    for agent_id in shared_observation_spaces:
        agent_space = dict(shared_observation_spaces[agent_id].spaces.items())
        for other_id in shared_observation_spaces:
            if other_id != agent_id:
                for key, space in shared_observation_spaces[other_id].spaces.items():
                    agent_space[f"{other_id}_{key}"] = space
        flat[agent_id] = spaces.Dict(agent_space)
    return spaces.Dict(flat)
