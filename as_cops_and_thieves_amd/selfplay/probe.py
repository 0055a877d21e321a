"""A learnability probe (not part of the reference): the device env with its rewards replaced by a contextual-bandit
signal computed from the SAME ray observations the policies see -- +1 when the action is the impulse that points at the
agent's nearest ray, else 0.  Used by ``tools/learn_probe.py`` and ``tests/test_gpu_mappo.py`` to show that the learner
pipeline (packing, stacked conv/LSTM networks, graph-captured PPO update) learns a function of the observations within
seconds; the game's own objective needs tens of millions of env-steps (``tools/learn_curve.py``)."""
from __future__ import annotations

import torch

# ray k has world angle 2 pi k / R from +x toward +y (entity.py:182-193); impulses: 0 = -x, 1 = +y, 2 = +x, 3 = -y
# (entity.py:126-134) -> quadrant of the ray (centred on +x, +y, -x, -y) -> action
_QUADRANT_TO_ACTION = (2, 1, 0, 3)


class NearestRayRewardEnv:
    """Wraps a ``VecCopsEnv``: same surface, rewards replaced as described in the module docstring."""

    def __init__(self, env):
        self._env = env
        self._lut = torch.tensor(_QUADRANT_TO_ACTION, device=env.device)
        self._target = None

    def __getattr__(self, name):
        if name == "step_raw":       # the rewards are replaced in step(): a caller must not step the wrapped env behind it
            raise AttributeError(name)
        return getattr(self._env, name)

    def _targets(self, obs):
        R = next(iter(obs.values()))["distance"].shape[-1]
        t = {}
        for a, o in obs.items():
            k = o["distance"].float().argmin(dim=-1)                       # nearest ray of the observation acted on
            t[a] = self._lut[((k + R // 8) // (R // 4)) % 4]
        return t

    def _remember(self, obs):
        new = self._targets(obs)
        if self._target is None:
            self._target = new
        else:   # in place: the trainer replays its rollout as a HIP graph, so state carried between ticks keeps its address
            for a, t in new.items():
                self._target[a].copy_(t)

    def reset(self, *args, **kw):
        obs, infos = self._env.reset(*args, **kw)
        self._remember(obs)
        return obs, infos

    def step(self, actions):
        acts = actions if not isinstance(actions, dict) else torch.stack([actions[a] for a in self._env.possible_agents], dim=1)
        obs, _, terms, truncs, infos = self._env.step(actions)
        rewards = {a: (acts[:, i].long() == self._target[a]).float() for i, a in enumerate(self._env.possible_agents)}
        self._remember(obs)
        return obs, rewards, terms, truncs, infos
