"""GPU: the Python env surface (drop-in boundary) on top of the C ABI — reference-style driver
loop, PettingZoo-shaped dicts, batched env with auto-reset, checkpointing, model-input packing —
and full-size (BASELINE configs[1]) property checks."""
import numpy as np
import pytest

from tests.util import assert_outputs_equal, compiled, to_np

pytestmark = pytest.mark.gpu


def _oracle_for(env):
    from oracle.cat_oracle import OracleSim
    return OracleSim(env._cfg, env._compiled if isinstance(env._compiled, list) else [env._compiled])


def test_simple_env_driver_loop_matches_reference_surface():
    """src/driver.py:58-69 flow: SimpleEnv(map), reset(), random Discrete(4) actions per live agent."""
    from as_cops_and_thieves_amd import SimpleEnv, load_preset
    env = SimpleEnv(map=load_preset("squarinth"), max_step_count=30)
    cpu = _oracle_for(env)
    assert env.possible_agents == ["cop_0", "cop_1", "thief_0"] and env.time_step == 1 / 60.0
    obs, infos = env.reset()
    c = cpu.reset()
    assert list(obs) == env.possible_agents and infos == {a: {} for a in env.possible_agents}
    for i, a in enumerate(env.possible_agents):
        assert obs[a]["distance"].dtype == np.float16 and obs[a]["distance"].shape == (90,)
        assert obs[a]["object_type"].dtype == np.uint8
        assert np.array_equal(obs[a]["distance"].view(np.uint16), c["obs_distance"][0, i])
        assert env.observation_space(a)["distance"].shape == (90,) and env.action_space(a).n == 4
    ended = False
    for t in range(40):
        actions = {agent: env.action_space(agent).sample() for agent in env.agents}
        if not actions:
            assert env.step(actions) == ({}, {}, {}, {}, {})          # base_env.py:374-376, driver idles (Q12)
            continue
        a_arr = np.array([[actions[a] for a in env.possible_agents]], np.int32)
        obs, rewards, terminations, truncations, infos = env.step(actions)
        c = cpu.step(a_arr)
        for i, a in enumerate(env.possible_agents):
            assert np.array_equal(obs[a]["object_type"], c["obs_type"][0, i])
            assert np.float32(rewards[a]) == c["reward"][0, i] and isinstance(rewards[a], float)
            assert terminations[a] == bool(c["terminated"][0]) and truncations[a] == bool(c["truncated"][0])
        st = env.state()
        assert st["cop_0"]["object_type_shared"] is st["cop_1"]["object_type_shared"]     # aliased per team (Q7)
        assert st["cop_0"]["own_distances"] is obs["cop_0"]["distance"]
        assert st["thief_0"]["team_positions"].shape == (1, 2) and st["cop_0"]["team_positions"].dtype == np.float16
        if any(terminations.values()):
            ended = True
            assert env.agents == [] and infos["cop_0"]["winner"] in ("cop", "thief")
            with pytest.raises(ValueError):                            # stepping a finished episode (base_env.py:384)
                env.step({"cop_0": 0, "cop_1": 0, "thief_0": 0})
            break
        assert infos["thief_0"]["winner"] is None
    assert ended
    frame = env.render()
    assert frame.shape == (1280, 800, 3) and frame.dtype == np.uint8
    assert env.get_nested_agent_observation_spaces()["cop_0"]["thief_0_distance_shared"].shape == (90,)
    env.close()


def test_reset_seed_reproducible_and_missing_regions_warn(capsys):
    from as_cops_and_thieves_amd import SimpleEnv, load_preset
    env = SimpleEnv(map=load_preset("squarinth"))
    o1, _ = env.reset(seed=123)
    p1 = env.cops[0].body.position
    o2, _ = env.reset(seed=123)
    assert p1 != (350.0, 350.0)
    env2 = SimpleEnv(map=load_preset("squarinth"))
    env2.reset(seed=123)
    assert env2.cops[0].body.position == p1
    env.close(); env2.close()
    env3 = SimpleEnv(map=load_preset("grandbyrinth"))
    capsys.readouterr()
    env3.reset()
    assert "No spawn regions defined in map for agent cop_0" in capsys.readouterr().out   # base_env.py:328-332
    assert env3.cops[0].body.position == (300.0, 400.0)
    env3.close()


def test_vec_env_matches_oracle_with_autoreset_and_packing():
    import torch
    from as_cops_and_thieves_amd import VecCopsEnv, load_preset
    from as_cops_and_thieves_amd import packing
    N, R = 64, 64
    env = VecCopsEnv(load_preset("lbirinth"), N, num_rays=R, max_step_count=20, seed=3)
    cpu = _oracle_for(env)
    obs, _ = env.reset()
    c = cpu.reset()
    torch.cuda.synchronize()
    assert obs["thief_0"]["distance"].shape == (N, R) and obs["thief_0"]["distance"].dtype == torch.float16
    for t in range(50):
        acts = env.random_actions(t).clone()
        obs, rew, term, trunc, infos = env.step({a: acts[:, i] for i, a in enumerate(env.possible_agents)})
        c = cpu.step(acts.cpu().numpy())
        done = c["terminated"].copy()
        assert np.array_equal(infos["winner"].cpu().numpy(), c["winner"])
        assert np.array_equal(term["cop_0"].cpu().numpy(), done.astype(bool))
        assert np.array_equal(rew["thief_0"].cpu().numpy().view(np.uint32), c["reward"][:, 2].view(np.uint32))
        c = cpu.reset(mask=done)                                        # the batched env auto-resets on device
        assert np.array_equal(obs["cop_1"]["object_type"].cpu().numpy(), c["obs_type"][:, 1])
    st = env.state()
    pol = packing.pack_policy_input(obs["cop_0"])
    assert pol.shape == (N, 2 * R) and torch.equal(packing.policy_view(pol, R)[:, 0], obs["cop_0"]["distance"].float())
    val = packing.pack_value_input(st)
    per_cop, per_thief = 4 * R + 2 * 2, 4 * R + 2 * 1
    assert val.shape == (N, 2 * per_cop + per_thief)
    v4 = packing.value_view(val, R)                                     # what LSTMValue slices (lstm_value_net.py:122-137)
    assert torch.equal(v4[:, 0], st["cop_0"]["distance_shared"].float()) and torch.equal(v4[:, 3], st["cop_0"]["own_obj_types"].float())
    env.close()


def test_env_state_checkpoint_resume_is_bit_exact():
    import torch
    from as_cops_and_thieves_amd import VecCopsEnv, load_preset
    env = VecCopsEnv(load_preset("squarinth"), 32, num_rays=64, max_step_count=25, seed=8)
    env.reset()
    for t in range(10):
        env.step(env.random_actions(t))
    snap = {k: v.clone() for k, v in env.get_env_state().items()}
    trace = []
    for t in range(10, 30):
        obs, rew, *_ = env.step(env.random_actions(t))
        trace.append((obs["cop_0"]["distance"].clone(), rew["thief_0"].clone()))
    env.set_env_state(**snap)
    for t in range(10, 30):
        obs, rew, *_ = env.step(env.random_actions(t))
        assert torch.equal(obs["cop_0"]["distance"], trace[t - 10][0]) and torch.equal(rew["thief_0"], trace[t - 10][1])
    env.close()


def test_full_size_properties_4096_envs():
    """BASELINE configs[1] size: determinism, slot independence, and domain invariants."""
    import torch
    from as_cops_and_thieves_amd.config import SimConfig
    from as_cops_and_thieves_amd.sim import CatSim
    from oracle.cat_oracle import OracleSim
    m = compiled("labyrinth")
    N, R, T = 4096, 64, 60
    cfg = SimConfig(n_envs=N, n_rays=R, max_step_count=40, seed=0)

    def run():
        sim = CatSim(cfg, [m], device="cuda:0")
        sim.reset()
        caps = 0
        for t in range(T):
            out = sim.step(sim.random_actions(t))
            caps += int((out["winner"] == 0).sum())
            ty, d = out["obs_type"], out["obs_distance"]
            assert int(((ty != 0) & (ty != 1) & (ty != 2) & (ty != 4)).sum()) == 0
            assert bool((d[ty == 4] == 400.0).all()) and float(d.float().max()) < 410.0
            st = sim.get_state()
            assert float(st["vel"].norm(dim=-1).max()) <= 125.0 * (1 + 1e-12) + 20.0   # clamp, plus at most one solve
            sim.reset_done()
        res = {k: v.clone() for k, v in sim.out.items()}, {k: v.clone() for k, v in sim.get_state().items()}
        sim.close()
        return res, caps

    (o1, s1), caps = run()
    (o2, s2), _ = run()
    for k in o1:
        assert torch.equal(o1[k], o2[k]), k                                # same seed -> same bits
    for k in s1:
        assert torch.equal(s1[k], s2[k]), k
    assert int(s1["reset_count"].min()) >= 2                              # every slot went through auto-reset
    # slot independence: slot k of the big batch == a 1-env run with that global env id (checked on the oracle side
    # too: the oracle with env_id_offset = k reproduces slot k of the GPU batch)
    for k in (0, 1777, 4095):
        cpu = OracleSim(SimConfig(n_envs=1, n_rays=R, max_step_count=40, seed=0, env_id_offset=k), [m])
        cpu.reset()
        for t in range(T):
            c = cpu.step(cpu.random_actions(t))
            cpu.reset(mask=c["terminated"].copy())
        assert np.array_equal(cpu.get_state()["pos"][0], s1["pos"][k].cpu().numpy())
        assert np.array_equal(cpu.out["obs_type"][0], o1["obs_type"][k].cpu().numpy())
    # domain invariant: an agent centre never gets within 0.5 px of a wall surface or of another agent's
    # circle (that would be > 4.5 px of penetration; the soft contact equilibrium is ~1.8 px)
    cpu = OracleSim(SimConfig(n_envs=1, n_rays=R), [m])
    pos, tc = s1["pos"].cpu().numpy(), s1["tc"].cpu().numpy()
    for e in range(0, N, 61):
        cpu.set_state(tc=tc[e:e + 1])
        for i in range(3):
            assert not cpu.point_query_any(0, i, pos[e, i], 0.5)


@pytest.mark.parametrize("label,names,cops,thieves,N", [
    ("configs[3]", ["grandbyrinth"], 3, 2, 8192),
    ("configs[4]", ["agh-map", "grandbyrinth", "labyrinth", "lbirinth", "squarinth"], 2, 1, 16384),
])
def test_full_size_other_baseline_configs(label, names, cops, thieves, N):
    """BASELINE configs[3] (3v2, 8192 envs) and configs[4] (five maps interleaved, 16384 envs) at full size:
    same seed -> same bits, and sampled slots equal a one-env oracle run keyed with that slot's global env id."""
    import torch
    from as_cops_and_thieves_amd.config import SimConfig
    from as_cops_and_thieves_amd.sim import CatSim
    from oracle.cat_oracle import OracleSim
    maps = [compiled(n, cops, thieves) for n in names]
    slot = (np.arange(N) % len(maps)).astype(np.int32)
    R, T = 64, 24
    cfg = SimConfig(n_envs=N, n_cops=cops, n_thieves=thieves, n_rays=R, max_step_count=15, seed=7)

    def run():
        sim = CatSim(cfg, maps, slot, device="cuda:0")
        sim.reset()
        for t in range(T):
            sim.step_fused(None, tick=t, auto_reset=True)       # in-kernel Philox actions + in-kernel auto-reset
        res = {k: v.clone() for k, v in sim.out.items()}, {k: v.clone() for k, v in sim.get_state().items()}
        sim.close()
        return res

    (o1, s1), (o2, s2) = run(), run()
    assert all(torch.equal(o1[k], o2[k]) for k in o1) and all(torch.equal(s1[k], s2[k]) for k in s1), label
    assert int(s1["reset_count"].min()) >= 2                      # every slot saw at least one auto-reset (truncation at 15)
    for k in (0, N // 3 + 1, N - 1):
        cpu = OracleSim(SimConfig(n_envs=1, n_cops=cops, n_thieves=thieves, n_rays=R, max_step_count=15, seed=7,
                                  env_id_offset=k), [maps[slot[k]]])
        cpu.reset()
        for t in range(T):
            c = cpu.step(cpu.random_actions(t))
            cpu.reset(mask=c["terminated"].copy())
        assert np.array_equal(cpu.get_state()["pos"][0], s1["pos"][k].cpu().numpy()), (label, k)
        assert np.array_equal(cpu.out["obs_distance"][0].view(np.uint16), o1["obs_distance"][k].cpu().numpy().view(np.uint16)), (label, k)


def test_raw_env_is_the_aec_wrapper_of_the_single_env():
    """reference base_env.py:555-569: raw_env = parallel_to_aec(BaseEnv(map)).  pettingzoo's conversion is used as is; this
    image has no pettingzoo, where raw_env must say so instead of returning something else."""
    from as_cops_and_thieves_amd import load_preset, raw_env
    try:
        import pettingzoo  # noqa: F401
    except ImportError:
        with pytest.raises(ImportError, match="pettingzoo"):
            raw_env(load_preset("squarinth"), device="cuda:0")
        pytest.skip("pettingzoo is not installed: the AEC path itself cannot run here")
    env = raw_env(load_preset("squarinth"), device="cuda:0", num_rays=16)
    env.reset(seed=0)
    seen = []
    for agent in env.agent_iter(max_iter=9):              # three rounds of the three agents, in roster order
        obs, rew, term, trunc, info = env.last()
        seen.append(agent)
        env.step(None if term or trunc else 1)
    assert seen[:3] == ["cop_0", "cop_1", "thief_0"] and set(obs) == {"distance", "object_type"}
    env.close()
