"""GPU: a test double of the reference's callers at the drop-in boundary.  skrl and pettingzoo are not
installable in the build container, so this double does by hand what skrl's PettingZoo wrapper (``wrap_env(env,
wrapper="pettingzoo")``, src/self_play_driver.py:35), the model initialisation (src/training/orchestration.py:52-69)
and ``evaluate_agents`` (src/utils/eval_pfsp_agents.py:25-49) do with the env -- and nothing else -- against
``SimpleEnv`` on the HIP library."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _flatten_sorted(space_value) -> np.ndarray:
    """skrl's ``flatten_tensorized_space`` on a Dict: values concatenated in SORTED-KEY order, as float32."""
    if isinstance(space_value, dict):
        return np.concatenate([_flatten_sorted(space_value[k]) for k in sorted(space_value)])
    return np.asarray(space_value, dtype=np.float32).reshape(-1)


class SkrlStyleWrapper:
    """What skrl's multi-agent PettingZoo wrapper touches: possible_agents / agents, the space dicts and accessors,
    state_space, reset, step with untensorised Discrete actions, state(), render, close; presents num_envs = 1."""

    def __init__(self, env):
        self._env = env
        self.num_envs = 1
        self.possible_agents = list(env.possible_agents)
        self.observation_spaces = {a: env.observation_space(a) for a in self.possible_agents}
        self.action_spaces = {a: env.action_space(a) for a in self.possible_agents}
        assert set(env.observation_spaces) == set(env.action_spaces) == set(self.possible_agents)
        self.state_space = env.state_space

    @property
    def agents(self):
        return self._env.agents

    def reset(self):
        import torch
        obs, infos = self._env.reset()
        return {a: torch.from_numpy(_flatten_sorted(o)).view(1, -1) for a, o in obs.items()}, infos

    def state(self):
        import torch
        return torch.from_numpy(_flatten_sorted(self._env.state())).view(1, -1)

    def step(self, actions):
        import torch
        acts = {a: int(t.reshape(-1)[0].item()) for a, t in actions.items()}          # untensorize_space(Discrete)
        obs, rew, term, trunc, infos = self._env.step(acts)
        t = lambda d, dt: {a: torch.tensor(v, dtype=dt).view(1, -1) for a, v in d.items()}
        return ({a: torch.from_numpy(_flatten_sorted(o)).view(1, -1) for a, o in obs.items()}, t(rew, torch.float32),
                t(term, torch.bool), t(trunc, torch.bool), infos)


def test_skrl_style_wrapper_and_model_init_see_the_reference_layouts():
    import torch
    from as_cops_and_thieves_amd import SimpleEnv, load_preset, packing
    env = SimpleEnv(map=load_preset("squarinth"), render_mode="rgb_array", max_step_count=2000)   # self_play_driver.py:34
    w = SkrlStyleWrapper(env)
    R = env.observation_space("cop_0")["distance"].shape[0]
    # orchestration.py:52-69: models are sized from these
    base = env.get_base_observation_space_structure()
    nested = env.get_nested_agent_observation_spaces()
    assert set(nested) == set(env.possible_agents) and callable(env.observation_space) and callable(env.action_space)
    assert sorted(base) == sorted(env.possible_agents) and env.action_space("thief_0").n == 4
    assert sorted(env.observation_space("cop_0").spaces if hasattr(env.observation_space("cop_0"), "spaces")
                  else env.observation_space("cop_0").keys()) == ["distance", "object_type"]
    obs, infos = w.reset()
    assert list(obs) == env.possible_agents and infos == {a: {} for a in env.possible_agents}
    raw_obs, _ = None, None
    for a in env.possible_agents:
        assert obs[a].shape == (1, 2 * R) and obs[a].dtype == torch.float32
    # the policy's (B, 2, R) view: row 0 = distances, row 1 = object types (lstm_policy_net.py:101-103)
    st_dict = env.state()
    first = sorted(st_dict)[0]
    pol = obs[first].view(1, 2, R)
    assert np.array_equal(pol[0, 0].numpy(), st_dict[first]["own_distances"].astype(np.float32))
    assert np.array_equal(pol[0, 1].numpy(), st_dict[first]["own_obj_types"].astype(np.float32))
    # the critic's input: flattened state of ALL agents, sorted ids, sorted keys; LSTMValue slices the first 4R entries
    flat = w.state()
    T_total = sum(v["team_positions"].size for v in st_dict.values())
    assert flat.shape == (1, len(st_dict) * 4 * R + T_total)
    as_t = {a: {k: torch.from_numpy(np.ascontiguousarray(v)).unsqueeze(0) for k, v in d.items()} for a, d in st_dict.items()}
    assert torch.equal(flat, packing.pack_value_input(as_t))
    v4 = packing.value_view(flat, R)[0].numpy()
    assert np.array_equal(v4[0], st_dict[first]["distance_shared"].astype(np.float32))
    assert np.array_equal(v4[3], st_dict[first]["own_obj_types"].astype(np.float32))
    # a few wrapped steps with tensor actions in, tensors out
    for t in range(5):
        actions = {a: torch.tensor([[env.action_space(a).sample()]]) for a in w.agents}
        obs, rew, term, trunc, infos = w.step(actions)
        assert rew["cop_0"].shape == (1, 1) and term["thief_0"].dtype == torch.bool and set(infos) == set(env.possible_agents)
    assert env.render().dtype == np.uint8
    env.close()


def test_evaluate_agents_loop_of_the_reference_runs_unchanged():
    """eval_pfsp_agents.py:25-59 restated against the env: obs -> tensor, 0-d long tensor actions straight into
    env.step, the episode ends at any(terminations.values()), the winner is read from infos[first]["winner"]."""
    import torch
    from as_cops_and_thieves_amd import SimpleEnv, load_preset
    env = SimpleEnv(map=load_preset("squarinth"), max_step_count=40)
    wins = {"cop": 0, "thief": 0}
    gen = torch.Generator().manual_seed(0)
    for episode in range(3):
        obs, _ = env.reset()
        done, ticks = False, 0
        while not done:
            actions = {}
            for agent_name in env.possible_agents:
                obs_tensor = torch.as_tensor(np.concatenate([obs[agent_name][k].astype(np.float32) for k in sorted(obs[agent_name])])).unsqueeze(0)
                assert obs_tensor.shape[0] == 1
                action = torch.randint(0, 4, (1, 1), generator=gen)                  # stands in for policy.act(...)
                assert hasattr(env.action_space(agent_name), "n")                    # Discrete branch of the reference
                actions[agent_name] = action.squeeze().to(dtype=torch.long)          # a 0-d tensor, as the reference passes
            obs, rewards, terminations, truncations, infos = env.step(actions)
            ticks += 1
            if any(terminations.values()):
                first_agent = next(iter(infos))
                winner = infos[first_agent]["winner"]
                assert winner in ("cop", "thief")
                wins[winner] += 1
                done = True
        assert ticks <= 40
    env.reset()
    assert wins["cop"] + wins["thief"] == 3
    # an action outside Discrete(4) is an error at this boundary, as in the reference
    with pytest.raises((TypeError, ValueError, KeyError, IndexError)):
        env.step({a: 7 for a in env.possible_agents})
    env.close()


def test_out_of_range_actions_are_flagged_by_the_c_abi_not_silently_ignored():
    import torch
    from as_cops_and_thieves_amd import VecCopsEnv, load_preset
    env = VecCopsEnv(load_preset("squarinth"), 16, num_rays=8)
    env.reset()
    env.step(torch.zeros(16, 3, dtype=torch.int32, device=env.device))
    env.check_errors()                                                             # nothing to report
    bad = torch.zeros(16, 3, dtype=torch.int32, device=env.device)
    bad[5, 1] = 4
    before = env.get_env_state()["vel"][5, 1].clone()
    env.step(bad)
    with pytest.raises(ValueError):
        env.check_errors()
    env.check_errors()                                                             # the flags were cleared by the read
    env.close()
