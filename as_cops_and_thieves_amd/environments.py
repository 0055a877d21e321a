"""Environment classes: the drop-in boundary of the hot path.

``BaseEnv`` / ``SimpleEnv`` mirror reference ``src/environments/base_env.py`` and
``simple_env.py`` (PettingZoo ``ParallelEnv`` surface, one env, per-agent dicts of NumPy arrays),
so a ``driver.py``-style loop, skrl's PettingZoo wrapper and ``evaluate_agents``
(``src/utils/eval_pfsp_agents.py:25-49``) run against them unchanged — except that the tick is
computed by the HIP kernels behind ``libcat_sim.so`` instead of Pymunk.  ``VecCopsEnv`` is the
batched form of the same surface: same keys, values are torch tensors with a leading ``num_envs``
axis, finished episodes auto-reset on device.

``BaseEnv`` subclasses ``pettingzoo.ParallelEnv`` when pettingzoo is importable (it is not installable in the build
container, where the class is duck-typed to the attributes its wrappers use; ``tests/test_gpu_boundary_double.py``
drives exactly those).  There is no CPU path: constructing an env without the built extension or without a GPU raises.
"""
from __future__ import annotations

import itertools
from pathlib import Path
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from . import spaces
from .agents import Cop, Entity, Thief
from .config import SimConfig
from .constants import DEFAULT_SENSOR, PhysicalParams, load_physical_params
from .maps import CompiledMap, Map
from .observation_spaces import (_init_shared_observation_space,
                                 get_nested_agent_observation_spaces as _nested_spaces)
from .sim import CatSim

WINNER_NAMES = {-1: None, 0: "cop", 1: "thief"}

try:  # pragma: no cover - depends on the installation
    from pettingzoo import ParallelEnv as _ParallelEnvBase  # type: ignore
    HAVE_PETTINGZOO = True
except ModuleNotFoundError:
    _ParallelEnvBase = object
    HAVE_PETTINGZOO = False


def _physical_from_cwd() -> PhysicalParams:
    """The reference reads ``pyproject.toml [tool.physical-params]`` from the CWD
    (``src/utils/toml_utils.py:40-46``); honour such a file when present, else the defaults."""
    p = Path("pyproject.toml")
    if p.exists():
        try:
            return load_physical_params(p)
        except (KeyError, OSError, ValueError):
            pass
    return load_physical_params(None)


class BaseEnv(_ParallelEnvBase):
    """Single-env PettingZoo ``ParallelEnv`` surface (reference ``BaseEnv(ParallelEnv)``, base_env.py:30-554)."""

    metadata = {"render_modes": ["human", "rgb_array"], "name": "cops_and_thieves_amd"}

    def __init__(self, map: Map, map_image: Optional[Path] = None, render_mode: Optional[str] = None,
                 max_step_count: int = 400, time_step: float = 1 / 15.0, *,
                 num_rays: int = DEFAULT_SENSOR.num_rays, device=None, seed: int = 1,
                 physical: Optional[PhysicalParams] = None, bbtree_gate: bool = True):
        assert render_mode is None or render_mode in self.metadata["render_modes"]  # base_env.py:117
        self.map, self.map_image = map, map_image
        self.width, self.height = self.map.window_dimensions
        self.step_count = 0
        self.max_step_count, self.time_step = max_step_count, time_step
        self.render_mode = render_mode
        physical = physical or _physical_from_cwd()
        self.cop_category = physical.pymunk_cop_category
        self.thief_category = physical.pymunk_thief_category
        self._termination_radius = physical.termination_radius

        self._cfg = SimConfig.from_params(
            physical=physical, n_envs=1, n_cops=map.cops_count, n_thieves=map.thieves_count,
            max_step_count=max_step_count, dt=time_step, seed=seed, bbtree_gate=int(bbtree_gate))
        self._cfg.n_rays = num_rays
        self._compiled: CompiledMap = map.compile(self._cfg.wall_radius)
        self._sim = CatSim(self._cfg, [self._compiled], device=device)

        group_counter = itertools.count(1)                              # base_env.py:90-92
        A = self._cfg.n_agents
        self.cops: List[Cop] = [
            Cop(self, i, f"cop_{i}", next(group_counter), self.cop_category, num_rays, self._cfg.ray_length, physical)
            for i in range(map.cops_count)]
        self.thieves: List[Thief] = [
            Thief(self, map.cops_count + j, f"thief_{j}", next(group_counter), self.thief_category, num_rays,
                  self._cfg.ray_length, physical)
            for j in range(map.thieves_count)]
        everyone: List[Entity] = self.cops + self.thieves
        assert len(everyone) == A
        self.possible_agents = [a.get_id() for a in everyone]           # base_env.py:96
        self.agents: List[str] = []
        self.agent_name_mapping = {a.get_id(): a for a in everyone}
        self.observation_spaces = {a.get_id(): a.observation_space for a in everyone}
        self.action_spaces = {a.get_id(): a.action_space for a in everyone}
        shared = _init_shared_observation_space(map=self.map, cops=self.cops, thieves=self.thieves)
        self.shared_observation_spaces = shared                         # replaced by VALUES after reset()
        self._shared_observation_spaces = shared
        self.state_space = shared
        self._state_cache: Dict[str, np.ndarray] = {}

    # ---- PettingZoo conveniences -------------------------------------------------------
    @property
    def unwrapped(self):
        return self

    @property
    def num_agents(self) -> int:
        return len(self.agents)

    @property
    def max_num_agents(self) -> int:
        return len(self.possible_agents)

    def observation_space(self, agent: str):
        return self.agent_name_mapping[agent].observation_space

    def action_space(self, agent: str):
        return self.agent_name_mapping[agent].action_space

    def get_base_observation_space_structure(self):
        return self._shared_observation_spaces

    def get_nested_agent_observation_spaces(self):
        return _nested_spaces(self._shared_observation_spaces)

    # ---- helpers ---------------------------------------------------------------------------
    def _host_state(self, name: str) -> np.ndarray:
        return self._sim.get_state()[name].cpu().numpy()

    def _observations_from_outputs(self) -> Dict[str, dict]:
        out = self._sim.out
        torch.cuda.current_stream(self._sim.device).synchronize()
        dist = out["obs_distance"][0].cpu().numpy()                     # [A,R] float16
        typ = out["obs_type"][0].cpu().numpy()
        obs = {aid: {"distance": dist[i].copy(), "object_type": typ[i].copy()}
               for i, aid in enumerate(self.possible_agents)}
        # get_shared_observations (observation_spaces.py:98-129): the two aggregate arrays and the
        # team positions are shared (aliased) by all members of a team
        sd = out["shared_distance"][0].cpu().numpy()
        st = out["shared_type"][0].cpu().numpy()
        tp = out["team_positions"][0].cpu().numpy()
        nc = self.map.cops_count
        shared = {}
        for team, members, positions in ((0, self.cops, tp[:nc]), (1, self.thieves, tp[nc:])):
            if not members:
                continue
            team_obj, team_dist, team_pos = st[team].copy(), sd[team].copy(), positions.copy()
            for agent in members:
                aid = agent.get_id()
                shared[aid] = {"own_obj_types": obs[aid]["object_type"], "own_distances": obs[aid]["distance"],
                               "object_type_shared": team_obj, "distance_shared": team_dist,
                               "team_positions": team_pos}
        self.shared_observation_spaces = shared
        return obs

    # ---- reset / step ------------------------------------------------------------------------
    def reset(self, seed: Optional[int] = None, options: Optional[dict] = None):
        """base_env.py:286-352.  ``options={"positions": [[x, y], ...]}`` (cops first) injects the
        spawn positions instead of sampling them (build-side extension used by parity tests)."""
        if seed is not None:
            self._sim.set_seed(seed)
        self.agents = self.possible_agents[:]
        for agent_id in self.agents:                                    # base_env.py:323-332 warnings
            regions = self.map.agent_spawn_regions.get(agent_id)
            if agent_id not in self.map.agent_spawn_regions:
                print(f"Warning: No spawn regions defined in map for agent {agent_id}. Using default reset "
                      f"(current position or initial).")
            elif not regions:
                print(f"Warning: Agent {agent_id} has an empty list of spawn regions. Using default reset "
                      f"(current position or initial).")
        positions = None
        if options and options.get("positions") is not None:
            positions = torch.as_tensor(np.asarray(options["positions"], dtype=np.float64)).reshape(1, -1, 2)
        self._sim.reset(positions=positions)
        observations = self._observations_from_outputs()
        infos = {agent_id: {} for agent_id in self.agents}
        self.step_count = 0
        return observations, infos

    def step(self, action: dict):
        """base_env.py:354-413."""
        self.step_count += 1
        if not action:                                                  # :374-376
            self.agents = []
            sc = self._sim.get_state()["step_count"]
            self._sim.set_state(step_count=sc + 1)
            return {}, {}, {}, {}, {}
        if not self.agents:
            # the reference unpacks an empty result list here (base_env.py:384-386)
            sc = self._sim.get_state()["step_count"]
            self._sim.set_state(step_count=sc + 1)
            raise ValueError("not enough values to unpack (expected 5, got 0)")
        acts = np.empty((1, len(self.possible_agents)), dtype=np.int32)
        for i, agent in enumerate(self.agents):
            a = action[agent]                                           # KeyError for a missing agent
            a = int(a.item()) if hasattr(a, "item") else int(a)
            if a not in (0, 1, 2, 3):
                raise TypeError(f"invalid action {a!r} for agent {agent}: expected one of 0..3")
            acts[0, i] = a
        self._sim.step(torch.from_numpy(acts))
        observations = self._observations_from_outputs()
        out = self._sim.out
        rew = out["reward"][0].cpu().numpy()
        terminated = bool(out["terminated"][0].item())
        truncated = bool(out["truncated"][0].item())
        winner = WINNER_NAMES[int(out["winner"][0].item())]
        rewards = {a: float(rew[i]) for i, a in enumerate(self.agents)}
        terminations = {a: terminated for a in self.agents}             # entity.py:146
        truncations = {a: truncated for a in self.agents}               # base_env.py:397
        infos = {a: {"winner": winner} for a in self.agents}            # :410-411
        if terminated:
            self.agents = []                                            # :401-402
        return observations, rewards, terminations, truncations, infos

    def state(self) -> dict:
        return self.shared_observation_spaces                           # base_env.py:415-425

    def render(self):
        if self.render_mode == "rgb_array":
            from .render import render_rgb_array
            st = self._sim.get_state()
            return render_rgb_array(self._compiled, st["pos"][0].cpu().numpy(), self.map.cops_count,
                                    self._cfg.agent_radius)
        return None

    def close(self) -> None:
        self._sim.close()

    def _get_info(self) -> dict:
        return {"step_count": self.step_count, "thief_count": len(self.thieves), "cop_count": len(self.cops)}


class SimpleEnv(BaseEnv):
    """reference ``SimpleEnv`` (simple_env.py:6-58): dt = 1/60, render_mode "rgb_array"."""

    def __init__(self, map: Map, render_mode: str = "rgb_array", map_image: Optional[Path] = None,
                 max_step_count: int = 400, time_step: float = 1 / 60.0, **kw):
        super().__init__(map=map, map_image=map_image, render_mode=render_mode,
                         max_step_count=max_step_count, time_step=time_step, **kw)

    def _get_info(self) -> dict:
        info = super()._get_info()
        pos = self._host_state("pos")[0]
        nc = self.map.cops_count
        info.update({"environment_type": "SimpleEnv",
                     "thief_positions": [tuple(int(v) for v in p) for p in pos[nc:]],
                     "cop_positions": [tuple(int(v) for v in p) for p in pos[:nc]]})
        return info


def raw_env(map: Map, render_mode: str = "rgb_array", **kw):
    """reference ``raw_env`` (base_env.py:555-569): the single env wrapped for PettingZoo's AEC API with
    ``pettingzoo.utils.conversions.parallel_to_aec``.  Needs pettingzoo (the conversion is its code, not restated here): without
    it an ImportError says so.  ``kw`` goes to ``BaseEnv`` (device, num_rays, ...)."""
    try:
        from pettingzoo.utils.conversions import parallel_to_aec
    except ImportError as exc:
        raise ImportError("raw_env() returns pettingzoo's AEC wrapper of BaseEnv and needs the pettingzoo package "
                          "(BaseEnv / SimpleEnv / VecCopsEnv work without it)") from exc
    return parallel_to_aec(BaseEnv(map=map, render_mode=render_mode, **kw))


class VecCopsEnv:
    """Batched env: ``num_envs`` independent envs advanced in lock-step on one GPU.

    Same agent ids, spaces and dict keys as ``BaseEnv``; every value is a torch tensor on the
    device with a leading ``num_envs`` axis (zero-copy views of the buffers the kernels write).
    After a terminal tick the env slot is reset on device (``auto_reset``), and the observation
    returned for that slot is the first observation of the new episode; ``infos`` carries the
    per-slot ``winner`` (int8: -1 none, 0 cop, 1 thief), ``terminated`` and ``truncated`` flags of
    the tick that just ended.
    """

    metadata = BaseEnv.metadata

    def __init__(self, maps, num_envs: int, *, slot_map_ids: Optional[Sequence[int]] = None,
                 num_rays: int = 64, max_step_count: int = 400, time_step: float = 1 / 60.0,
                 auto_reset: bool = True, device=None, seed: int = 1, env_id_offset: int = 0,
                 physical: Optional[PhysicalParams] = None, bbtree_gate: bool = True):
        self.maps: List[Map] = list(maps) if isinstance(maps, (list, tuple)) else [maps]
        m0 = self.maps[0]
        for m in self.maps[1:]:
            if (m.cops_count, m.thieves_count) != (m0.cops_count, m0.thieves_count):
                raise ValueError("all maps of a batch must share one roster")
        physical = physical or _physical_from_cwd()
        self.num_envs, self.auto_reset = num_envs, auto_reset
        self._cfg = SimConfig.from_params(
            physical=physical, n_envs=num_envs, n_cops=m0.cops_count, n_thieves=m0.thieves_count,
            max_step_count=max_step_count, dt=time_step, seed=seed, env_id_offset=env_id_offset,
            bbtree_gate=int(bbtree_gate))
        self._cfg.n_rays = num_rays
        self._compiled = [m.compile(self._cfg.wall_radius) for m in self.maps]
        self._sim = CatSim(self._cfg, self._compiled, slot_map_ids, device=device)
        self.device = self._sim.device
        self.possible_agents = [f"cop_{i}" for i in range(m0.cops_count)] + \
                               [f"thief_{j}" for j in range(m0.thieves_count)]
        self.agents = self.possible_agents[:]
        group = itertools.count(1)
        self.cops = [Cop(self, i, f"cop_{i}", next(group), physical.pymunk_cop_category, num_rays,
                         self._cfg.ray_length, physical) for i in range(m0.cops_count)]
        self.thieves = [Thief(self, m0.cops_count + j, f"thief_{j}", next(group), physical.pymunk_thief_category,
                              num_rays, self._cfg.ray_length, physical) for j in range(m0.thieves_count)]
        everyone = self.cops + self.thieves
        self.agent_name_mapping = {a.get_id(): a for a in everyone}
        self.observation_spaces = {a.get_id(): a.observation_space for a in everyone}
        self.action_spaces = {a.get_id(): a.action_space for a in everyone}
        self._shared_observation_spaces = _init_shared_observation_space(m0, self.cops, self.thieves)
        self.state_space = self._shared_observation_spaces
        self.max_step_count, self.time_step = max_step_count, time_step
        self._actions = torch.zeros((num_envs, len(everyone)), dtype=torch.int32, device=self.device)

    # spaces
    def observation_space(self, agent: str):
        return self.observation_spaces[agent]

    def action_space(self, agent: str):
        return self.action_spaces[agent]

    def get_base_observation_space_structure(self):
        return self._shared_observation_spaces

    def get_nested_agent_observation_spaces(self):
        return _nested_spaces(self._shared_observation_spaces)

    def _host_state(self, name: str) -> np.ndarray:
        return self._sim.get_state()[name].cpu().numpy()

    def _obs(self) -> Dict[str, Dict[str, torch.Tensor]]:
        o = self._sim.out
        return {aid: {"distance": o["obs_distance"][:, i], "object_type": o["obs_type"][:, i]}
                for i, aid in enumerate(self.possible_agents)}

    def reset(self, seed: Optional[int] = None, options: Optional[dict] = None):
        if seed is not None:
            self._sim.set_seed(seed)
        positions = None if not options else options.get("positions")
        mask = None if not options else options.get("mask")
        self._sim.reset(mask=mask, positions=positions)
        return self._obs(), {a: {} for a in self.possible_agents}

    def step(self, actions):
        """``actions``: int tensor ``[num_envs, A]`` or ``{agent_id: tensor[num_envs]}``."""
        if isinstance(actions, dict):
            for i, aid in enumerate(self.possible_agents):
                self._actions[:, i] = actions[aid].reshape(self.num_envs).to(self._actions.dtype)
            acts = self._actions
        else:
            acts = actions
        # one launch: finished episodes are reset inside the tick kernel, which leaves the NEW episode's first
        # observations in the buffers of those slots (rewards / flags / winner are the terminal tick's)
        out = self._sim.step_fused(acts, auto_reset=self.auto_reset)
        rewards = {aid: out["reward"][:, i] for i, aid in enumerate(self.possible_agents)}
        terminated, truncated = out["terminated"].bool(), out["truncated"].bool()   # fresh tensors
        infos = {"winner": out["winner"].clone(), "terminated": terminated, "truncated": truncated}
        terminations = {aid: terminated for aid in self.possible_agents}
        truncations = {aid: truncated for aid in self.possible_agents}
        return self._obs(), rewards, terminations, truncations, infos

    def state(self) -> Dict[str, Dict[str, torch.Tensor]]:
        o = self._sim.out
        nc = len(self.cops)
        res = {}
        for i, aid in enumerate(self.possible_agents):
            team = 0 if i < nc else 1
            sl = slice(0, nc) if team == 0 else slice(nc, None)
            res[aid] = {"own_obj_types": o["obs_type"][:, i], "own_distances": o["obs_distance"][:, i],
                        "object_type_shared": o["shared_type"][:, team], "distance_shared": o["shared_distance"][:, team],
                        "team_positions": o["team_positions"][:, sl]}
        return res

    def observations(self) -> Dict[str, Dict[str, torch.Tensor]]:
        """The per-agent observation dictionaries over the current output buffers (what ``step`` / ``reset`` return)."""
        return self._obs()

    def step_raw(self, actions: torch.Tensor) -> Dict[str, torch.Tensor]:
        """``step`` without the per-agent dictionaries: one launch (tick + auto-reset), returns the output buffers
        themselves (``raw_outputs()``): reward fp32 [N, A], terminated / truncated u8 [N], winner, observations."""
        return self._sim.step_fused(actions, auto_reset=self.auto_reset)

    def raw_outputs(self) -> Dict[str, torch.Tensor]:
        """The env core's output buffers as they lie on the device (``include/cat_sim.h`` ``cat_outputs``: f16 distances,
        u8 types, [N, A, R] / [N, 2, R]): what ``_obs()`` / ``state()`` slice per agent, for callers that pack the model
        inputs themselves (``include/cat_rollout.h``)."""
        return self._sim.out

    def random_actions(self, tick: int) -> torch.Tensor:
        return self._sim.random_actions(tick, out=self._actions)

    def rollout_random(self, ticks: int, tick0: int = 0) -> Dict[str, torch.Tensor]:
        """``ticks`` env ticks under uniformly random actions -- what ``driver.py:65-69`` does tick by tick with
        ``action_space(agent).sample()`` and skrl's trainer for ``random_timesteps`` (``mappo_config.py:9``) -- as ONE resident
        launch (``cat_rollout_fused``: the map stays in LDS, the state records stay in LDS) for the first ``ticks - 1`` ticks, whose
        outputs nobody reads (NULL output pointers: nothing is stored), and one ordinary step for the last, which leaves the
        ``[N, ...]`` output buffers as ``step`` does.  The actions are the synthetic Philox draws of ticks ``tick0 .. tick0 + ticks - 1``
        (``cat_random_actions``).  Returns the raw output buffers (``raw_outputs()``)."""
        ticks = int(ticks)
        if ticks < 1:
            raise ValueError("ticks must be >= 1")
        done, cap = 0, 65536
        while ticks - 1 - done > 0:
            n = min(cap, ticks - 1 - done)
            self._sim.rollout_fused(n, None, tick=tick0 + done, auto_reset=self.auto_reset, out={})
            done += n
        out = self._sim.step_fused(None, tick=tick0 + ticks - 1, auto_reset=self.auto_reset)
        self._sim.check_errors()   # thousands of ticks went by inside one launch: a flag raised by any of them ends the run here, not never
        return out

    def check_errors(self) -> None:
        """The step launches are asynchronous and cannot raise; this synchronises and raises ``ValueError`` if any
        action since the last check was outside ``Discrete(4)`` (the reference raises at once), ``CatSimError`` if a
        wall contact had to be dropped."""
        self._sim.check_errors()

    def get_env_state(self) -> Dict[str, torch.Tensor]:
        """Full simulator state (bodies, caches, counters) for checkpointing."""
        return self._sim.get_state()

    def set_env_state(self, **arrays) -> None:
        self._sim.set_state(**arrays)

    def close(self) -> None:
        self._sim.close()
