"""GPU parity: the HIP env core (through the C ABI) against the CPU oracle, bit for bit.

Integer/byte outputs (ray classes, hit shape indices, f16 distance bits, flags, winner) and the
f64 body state are compared for exact equality on identical seeded inputs: both sides evaluate
the same formulas with contraction off, so even positions/velocities must agree to the last bit
(north_star asks 1e-5; exact is the bar here).  Oracle parity against Pymunk itself is UNPINNED
(see oracle/cat_oracle.h).
"""
import numpy as np
import pytest

from tests.util import (assert_outputs_equal, assert_state_equal, compiled, free_positions, to_np)

pytestmark = pytest.mark.gpu


def _pair(cfg, maps, slot=None):
    import torch
    from as_cops_and_thieves_amd.sim import CatSim
    from oracle.cat_oracle import OracleSim
    return CatSim(cfg, maps, slot, device="cuda:0", debug_hit_shape=True), OracleSim(cfg, maps, slot)


def _run(cfg, maps, slot, ticks, rng, spread=None, check_every=1, auto_reset=False):
    import torch
    from as_cops_and_thieves_amd.config import SimConfig
    gpu, cpu = _pair(cfg, maps, slot)
    pos = free_positions(cpu, maps[0] if slot is None else maps[int(slot[0])], rng, spread=spread) \
        if slot is None else None
    if pos is not None:
        g = gpu.reset(positions=torch.from_numpy(pos)); c = cpu.reset(positions=pos)
    else:
        g = gpu.reset(); c = cpu.reset()
    torch.cuda.synchronize()
    assert_outputs_equal(to_np(g), c, keys=("obs_distance", "obs_type", "hit_shape", "shared_distance",
                                            "shared_type", "team_positions"), ctx="reset")
    assert_state_equal(to_np(gpu.get_state()), cpu.get_state(), ctx="reset")
    n_contacts = n_captured = n_done = 0
    for t in range(ticks):
        a = cpu.random_actions(t)
        ag = gpu.random_actions(t)
        assert np.array_equal(ag.cpu().numpy(), a), "device Philox actions differ from the oracle's"
        g = gpu.step(ag); c = cpu.step(a)
        if t % check_every == 0 or t == ticks - 1:
            torch.cuda.synchronize()
            assert_outputs_equal(to_np(g), c, ctx=f"tick {t}")
            st = cpu.get_state()
            assert_state_equal(to_np(gpu.get_state()), st, ctx=f"tick {t}")
            n_contacts += int((st["wall_shape"] >= 0).sum() + (st["pair_age"] >= 0).sum())
        n_captured += int((c["winner"] == 0).sum())
        n_done += int(c["terminated"].sum())
        if auto_reset:
            done = c["terminated"].copy()
            gpu.reset_done()
            cpu.reset(mask=done)
            if done.any():
                torch.cuda.synchronize()
                assert_outputs_equal(to_np(gpu.out), cpu.out, keys=("obs_distance", "obs_type", "hit_shape",
                                     "shared_distance", "shared_type", "team_positions"), ctx=f"auto-reset {t}")
                assert_state_equal(to_np(gpu.get_state()), cpu.get_state(), ctx=f"auto-reset {t}")
    assert gpu.device_errors() == 0          # no wall contact dropped for want of a cache slot, no bad action
    gpu.close()
    return dict(contacts=n_contacts, captured=n_captured, done=n_done)


@pytest.mark.parametrize("name,rays", [("squarinth", 90), ("lbirinth", 90), ("labyrinth", 64),
                                         ("grandbyrinth", 64), ("agh-map", 90)])
def test_step_parity_injected_positions(name, rays):
    from as_cops_and_thieves_amd.config import SimConfig
    import zlib
    rng = np.random.default_rng(zlib.crc32(name.encode()))
    m = compiled(name)
    cfg = SimConfig(n_envs=32, n_rays=rays, max_step_count=60, seed=7)
    stats = _run(cfg, [m], None, ticks=80, rng=rng, spread=40.0)
    assert stats["done"] > 0          # timeouts reached (and rewards/winner paths exercised)


@pytest.mark.parametrize("name,rays", [("labyrinth", 64), ("squarinth", 90)])
def test_chunk_form_of_the_ray_fan_on_the_light_maps(name, rays, monkeypatch):
    """cat_create picks the form of the ray fan from the maps: rays that meet few walls (all but the agh-map) -> agent groups with
    the candidate-less rays sorted out first (fan_group); else one 64-ray chunk per work unit (fan_chunk).  The other parity tests
    therefore run fan_group on the four small maps and fan_chunk on the agh-map and on the mixed batches; this one forces
    fan_chunk on small maps (CAT_FAN=chunks), so that each form is held against the oracle on each kind of map."""
    from as_cops_and_thieves_amd.config import SimConfig
    monkeypatch.setenv("CAT_FAN", "chunks")
    cfg = SimConfig(n_envs=32, n_rays=rays, max_step_count=40, seed=17)
    stats = _run(cfg, [compiled(name)], np.zeros(32, np.int32), ticks=60, rng=np.random.default_rng(2), auto_reset=True)
    assert stats["done"] >= 32


@pytest.mark.parametrize("name,rays,cops,thieves,pool", [("labyrinth", 64, 2, 1, "0"), ("lbirinth", 64, 2, 1, "1"), ("squarinth", 64, 2, 1, "1"),
                                                         ("squarinth", 90, 1, 1, "1"), ("grandbyrinth", 48, 2, 2, "1"), ("labyrinth", 90, 2, 1, "0"),
                                                         ("lbirinth", 90, 2, 1, "1"), ("grandbyrinth", 64, 3, 2, "1"), ("grandbyrinth", 64, 3, 2, "0")])
def test_pooled_and_unit_form_of_the_light_maps_fan(name, rays, cops, thieves, pool, monkeypatch):
    """Where the rays of a workgroup fit an LDS ring (wpb * A * R <= 4096) cat_create picks between two schedulers of the group form: fan
    units of a slot's own rays (step_kernel / rollout_kernel), or one pool of the active rays of ALL slots from which any wave takes rounds
    (step_kernel_pooled / rollout_kernel_pooled), by the share of candidate-less rays around the spawn points.  The other parity tests run
    what it picks (the labyrinth: pooled; lbirinth, squarinth: units); this one forces the other choice (CAT_POOL), with the generic
    instantiation of the pooled kernels on rosters that have no fixed one, one-tick launches and a resident launch, against the oracle."""
    import torch
    from as_cops_and_thieves_amd.config import SimConfig
    from as_cops_and_thieves_amd.maps import load_preset
    from as_cops_and_thieves_amd.sim import CatSim
    from oracle.cat_oracle import OracleSim
    monkeypatch.setenv("CAT_POOL", pool)
    cfg = SimConfig(n_envs=48, n_cops=cops, n_thieves=thieves, n_rays=rays, max_step_count=40, seed=23)
    m = load_preset(name, cops, thieves).compile()
    stats = _run(cfg, [m], np.zeros(48, np.int32), ticks=60, rng=np.random.default_rng(4), auto_reset=True)
    assert stats["done"] >= 48
    gpu = CatSim(cfg, [m], device="cuda:0")
    if (cops, thieves) != (2, 2):   # (the four-agent roster only gets the pooled kernels if its ring fits the LDS left beside its larger env areas)
        assert gpu.one_tick_kernel == ("step_kernel_pooled" if pool == "1" else "step_kernel")
        assert gpu.rollout_kernel == ("rollout_kernel_pooled" if pool == "1" else "rollout_kernel")
    print(name, rays, cops, thieves, gpu.one_tick_kernel, gpu.rollout_kernel)
    cpu = OracleSim(cfg, [m])
    gpu.reset(); cpu.reset()
    rows = to_np(gpu.rollout_fused(50, None, tick=0, auto_reset=True))
    torch.cuda.synchronize()
    for t in range(50):
        c = cpu.step(cpu.random_actions(t))
        flags = {k: c[k].copy() for k in ("reward", "terminated", "truncated", "winner")}
        cpu.reset(mask=c["terminated"].copy())
        got = {k: v[t] for k, v in rows.items()}
        assert_outputs_equal(got, cpu.out, keys=("obs_distance", "obs_type", "shared_distance", "shared_type", "team_positions"), ctx=f"{name} resident tick {t}")
        assert_outputs_equal(got, flags, keys=tuple(flags), ctx=f"{name} resident tick {t}")
    assert_state_equal(to_np(gpu.get_state()), cpu.get_state(), ctx=f"{name} resident launch")
    assert gpu.device_errors() == 0
    gpu.close()


@pytest.mark.parametrize("name,N", [("labyrinth", 4096), ("squarinth", 2048), ("lbirinth", 1024)])
def test_pooled_and_unit_schedulers_agree_at_batch_size(name, N, monkeypatch):
    """Beyond the sizes the oracle reaches in seconds: two sims of the same seed, one per scheduler, through 300 one-tick launches and one 200-tick resident
    launch each (400-tick episodes would not end: max_step_count 120, so every slot resets twice) -- every output of every tick's last row and the whole state, bit for bit."""
    import torch
    from as_cops_and_thieves_amd.config import SimConfig
    from as_cops_and_thieves_amd.sim import CatSim
    cfg = SimConfig(n_envs=N, n_rays=64, max_step_count=120, seed=29)
    m = compiled(name)
    sims = []
    for pool in ("1", "0"):
        monkeypatch.setenv("CAT_POOL", pool)
        sims.append(CatSim(cfg, [m], device="cuda:0"))
    assert sims[0].one_tick_kernel == "step_kernel_pooled" and sims[1].one_tick_kernel == "step_kernel"
    outs = []
    for sim in sims:
        sim.reset()
        for t in range(300):
            o = sim.step_fused(None, tick=t, auto_reset=True)
        a = {k: v.clone() for k, v in o.items()}
        rows = sim.rollout_fused(200, None, tick=300, auto_reset=True)
        b = {k: v[-1].clone() for k, v in rows.items()}
        outs.append((a, b, {k: v.clone() for k, v in sim.get_state().items()}, sim.device_errors()))
    for part in range(3):
        for k in outs[0][part]:
            assert torch.equal(outs[0][part][k], outs[1][part][k]), (name, part, k)
    assert outs[0][3] == 0 and outs[1][3] == 0
    assert int(outs[0][2]["reset_count"].min()) >= 4
    for sim in sims:
        sim.close()


def test_default_choice_of_the_scheduler_per_entry():
    """Where the ring fits, cat_create gives the resident launch the pooled kernel on every map and the one-tick launch the pooled kernel unless practically every
    ray around the spawn points meets a wall (lbirinth) or the roster has more than four agents.  3v2 at 64 rays (BASELINE configs[3]): the ring FITS since the
    contact arrays of the scratch unions are sized by what the map makes possible (round 4: 5 KB short), but beside it the group arrays hold two agents, and the
    one-tick launch -- which stays on the unit form for this roster (97.4 us against 99.0 pooled) -- would fall from units of 4 + 1 agents to 2 + 2 + 1
    (102.5 us): the ring stays out unless CAT_POOL=1 asks for it (then both entries run pooled: the forced-form parity test above).  The dense map keeps
    the chunk form."""
    from as_cops_and_thieves_amd.config import SimConfig
    from as_cops_and_thieves_amd.maps import load_preset
    from as_cops_and_thieves_amd.sim import CatSim
    want = {("labyrinth", 2, 1): ("step_kernel_pooled", "rollout_kernel_pooled"), ("squarinth", 2, 1): ("step_kernel_pooled", "rollout_kernel_pooled"),
            ("lbirinth", 2, 1): ("step_kernel", "rollout_kernel_pooled"), ("grandbyrinth", 3, 2): ("step_kernel", "rollout_kernel"),
            ("labyrinth", 2, 1, 90): ("step_kernel_pooled", "rollout_kernel_pooled"),   # a ring of exactly wpb * A * R + 64 entries (no power of two fits)
            ("agh-map", 2, 1): ("step_kernel", "rollout_kernel")}
    for key, kernels in want.items():
        name, c, t = key[:3]
        rays = key[3] if len(key) > 3 else 64
        sim = CatSim(SimConfig(n_envs=64, n_cops=c, n_thieves=t, n_rays=rays, seed=1), [load_preset(name, c, t).compile()], device="cuda:0")
        assert (sim.one_tick_kernel, sim.rollout_kernel) == kernels, (name, sim.one_tick_kernel, sim.rollout_kernel)
        sim.close()


@pytest.mark.parametrize("n_envs", [1, 5, 17])
def test_pooled_fan_with_a_partly_filled_workgroup(n_envs, monkeypatch):
    """The pooled kernels with fewer env slots than a workgroup has waves (slots without an env publish nothing and count nothing), and with a
    second workgroup that holds a single env: one-tick launches, then a resident launch, against the oracle; the scheduler's watchdog flag stays clear."""
    import torch
    from as_cops_and_thieves_amd.config import SimConfig
    from as_cops_and_thieves_amd.sim import CatSim
    from oracle.cat_oracle import OracleSim
    monkeypatch.setenv("CAT_POOL", "1")
    m = compiled("labyrinth")
    cfg = SimConfig(n_envs=n_envs, n_rays=64, max_step_count=25, seed=5)
    stats = _run(cfg, [m], np.zeros(n_envs, np.int32), ticks=40, rng=np.random.default_rng(8), auto_reset=True)
    assert stats["done"] >= n_envs
    gpu, cpu = CatSim(cfg, [m], device="cuda:0"), OracleSim(cfg, [m])
    assert gpu.one_tick_kernel == "step_kernel_pooled"
    gpu.reset(); cpu.reset()
    rows = to_np(gpu.rollout_fused(60, None, tick=0, auto_reset=True))
    torch.cuda.synchronize()
    for t in range(60):
        c = cpu.step(cpu.random_actions(t))
        flags = {k: c[k].copy() for k in ("reward", "terminated", "truncated", "winner")}
        cpu.reset(mask=c["terminated"].copy())
        got = {k: v[t] for k, v in rows.items()}
        assert_outputs_equal(got, cpu.out, keys=("obs_distance", "obs_type", "shared_distance", "shared_type", "team_positions"), ctx=f"tick {t}")
        assert_outputs_equal(got, flags, keys=tuple(flags), ctx=f"tick {t}")
    assert_state_equal(to_np(gpu.get_state()), cpu.get_state(), ctx="resident launch")
    assert gpu.device_errors() == 0
    gpu.close()


@pytest.mark.parametrize("switch,value", [("CAT_GRID_HULLS", "0"), ("CAT_GRID_OCCLUSION", "0"), ("CAT_GRID_CELL", "16"), ("CAT_GRID_CELL", "5"),
                                          ("CAT_GRID_FIELDS", "0"), ("CAT_FAN", "chunks")])
@pytest.mark.parametrize("name", ["agh-map", "labyrinth"])
def test_results_do_not_depend_on_the_rules_that_build_the_candidate_table(name, switch, value, monkeypatch):
    """The spatial hash lists fewer walls with each of its rules on (cat_sim.hip build_grids: the hull rule, the occlusion rule, the
    smaller cell) and stores them in one of three row formats (four-byte fields for the group form, eight-byte fields or count + id
    bytes for the chunk form: finalize_rows); the oracle visits every wall.  Bit-exact parity with each rule switched off or coarsened
    and with each format forced."""
    from as_cops_and_thieves_amd.config import SimConfig
    monkeypatch.setenv(switch, value)
    cfg = SimConfig(n_envs=32, n_rays=64, max_step_count=60, seed=17)
    _run(cfg, [compiled(name)], None, ticks=80, rng=np.random.default_rng(6), spread=40.0)


@pytest.mark.parametrize("seed,rays", [(11, 64), (12, 90), (13, 64)])
def test_parity_on_random_polygon_maps(tmp_path, seed, rays):
    """Slanted, touching, overlapping and sliver blocks (tests/test_spatial_grid.py builds them): the candidate table's hull and
    occlusion rules meet every case they reason about, with agents spawned all over the map."""
    import json
    from as_cops_and_thieves_amd.config import SimConfig
    from as_cops_and_thieves_amd.maps import Map
    from tests.test_spatial_grid import _random_polygon_map
    _random_polygon_map(tmp_path, seed, 14)
    f = tmp_path / f"random_{seed}.json"
    data = json.loads(f.read_text())
    everywhere = {"x": 20, "y": 20, "w": 600, "h": 440}
    for a in data["agents"]:
        a["spawn_region"] = everywhere
    f.write_text(json.dumps(data))
    m = Map(f).compile()
    cfg = SimConfig(n_envs=64, n_rays=rays, max_step_count=25, seed=seed)
    _run(cfg, [m], np.zeros(64, np.int32), ticks=80, rng=np.random.default_rng(seed), auto_reset=True)


@pytest.mark.parametrize("rays", [100, 130, 200, 256, 300])
def test_group_form_of_the_ray_fan_across_ray_counts(rays):
    """fan_group groups two agents while their rays fill at most four 64-ray chunks (R <= 128) and one agent beyond (R <= 256: three
    or four chunks per agent); above 256 rays cat_create falls back to fan_chunk.  Every branch against the oracle."""
    from as_cops_and_thieves_amd.config import SimConfig
    cfg = SimConfig(n_envs=8, n_rays=rays, max_step_count=30, seed=rays)
    stats = _run(cfg, [compiled("lbirinth")], np.zeros(8, np.int32), ticks=40, rng=np.random.default_rng(rays), auto_reset=True)
    assert stats["done"] >= 8


def test_contacts_and_captures_are_exercised():
    from as_cops_and_thieves_amd.config import SimConfig
    rng = np.random.default_rng(3)
    m = compiled("lbirinth")
    cfg = SimConfig(n_envs=64, n_rays=90, max_step_count=400, seed=11)
    stats = _run(cfg, [m], None, ticks=150, rng=rng, spread=25.0)
    assert stats["contacts"] > 0 and stats["captured"] > 0


def test_philox_spawn_reset_and_autoreset_parity():
    from as_cops_and_thieves_amd.config import SimConfig
    rng = np.random.default_rng(5)
    m = compiled("squarinth")
    cfg = SimConfig(n_envs=48, n_rays=64, max_step_count=25, seed=99)
    stats = _run(cfg, [m], np.zeros(48, np.int32), ticks=70, rng=rng, auto_reset=True)
    assert stats["done"] >= 48 * 2


def test_three_vs_two_roster():
    from as_cops_and_thieves_amd.config import SimConfig
    rng = np.random.default_rng(6)
    m = compiled("grandbyrinth", 3, 2)
    cfg = SimConfig(n_envs=16, n_cops=3, n_thieves=2, n_rays=64, max_step_count=50, seed=5)
    _run(cfg, [m], None, ticks=60, rng=rng, spread=30.0)


def test_mixed_map_batch_ragged():
    """BASELINE config 5 shape: all five maps interleaved across env slots (slot mod 5), with a
    slot count that does not divide the workgroup's env count."""
    from as_cops_and_thieves_amd.config import SimConfig
    rng = np.random.default_rng(8)
    maps = [compiled(n) for n in ("agh-map", "grandbyrinth", "labyrinth", "lbirinth", "squarinth")]
    N = 37
    slot = (np.arange(N) % 5).astype(np.int32)
    cfg = SimConfig(n_envs=N, n_rays=64, max_step_count=30, seed=21)
    stats = _run(cfg, maps, slot, ticks=45, rng=rng, auto_reset=True)
    assert stats["done"] >= N


@pytest.mark.parametrize("split", ["1", "0"])
@pytest.mark.parametrize("pool", [None, "1", "0"])
def test_mixed_map_batch_in_two_parts_and_in_one(split, pool, monkeypatch):
    """A sim whose maps want both forms of the ray fan (BASELINE configs[4]: the agh-map needs the chunk form, the four box maps take the group form and its
    pooled kernels) runs in one part on the chunk form, or (CAT_SPLIT=1) cut into two parts, each with its own tables, LDS carve and kernels, launched side
    by side on two streams of the handle (cat_sim.hip cat_create / launch_parts).  Both ways, with the pooled / unit choice left to cat_create and forced
    either way: one-tick launches with separate resets, then cat_step_fused, then one resident launch, every slot against the oracle."""
    import torch
    from as_cops_and_thieves_amd.config import SimConfig
    from as_cops_and_thieves_amd.sim import CatSim
    from oracle.cat_oracle import OracleSim
    monkeypatch.setenv("CAT_SPLIT", split)
    if pool is not None:
        monkeypatch.setenv("CAT_POOL", pool)
    maps = [compiled(n) for n in ("agh-map", "grandbyrinth", "labyrinth", "lbirinth", "squarinth")]
    N = 5 * 19 + 3
    slot = (np.arange(N) % 5).astype(np.int32)
    cfg = SimConfig(n_envs=N, n_rays=64, max_step_count=30, seed=22)
    stats = _run(cfg, maps, slot, ticks=45, rng=np.random.default_rng(8), auto_reset=True)
    assert stats["done"] >= N
    gpu, cpu = CatSim(cfg, maps, slot, device="cuda:0"), OracleSim(cfg, maps, slot)
    if split == "1":
        light = "_pooled" if pool != "0" else ""
        assert gpu.rollout_kernel == f"rollout_kernel{light}+rollout_kernel", gpu.rollout_kernel
        assert gpu.one_tick_kernel == f"step_kernel{light}+step_kernel", gpu.one_tick_kernel
    else:
        assert (gpu.one_tick_kernel, gpu.rollout_kernel) == ("step_kernel", "rollout_kernel")
    gpu.reset(); cpu.reset()
    keys = ("obs_distance", "obs_type", "shared_distance", "shared_type", "team_positions")
    flag_keys = ("reward", "terminated", "truncated", "winner")
    for t in range(35):
        gpu.step_fused(None, tick=t, auto_reset=True)
        c = cpu.step(cpu.random_actions(t))
        flags = {k: c[k].copy() for k in flag_keys}
        cpu.reset(mask=c["terminated"].copy())
        if t % 5 == 0 or t >= 29:
            torch.cuda.synchronize()
            assert_outputs_equal(to_np(gpu.out), cpu.out, keys=keys, ctx=f"fused tick {t}")
            assert_outputs_equal(to_np(gpu.out), flags, keys=flag_keys, ctx=f"fused tick {t}")
            assert_state_equal(to_np(gpu.get_state()), cpu.get_state(), ctx=f"fused tick {t}")
    rows = to_np(gpu.rollout_fused(50, None, tick=35, auto_reset=True))
    torch.cuda.synchronize()
    for t in range(50):
        c = cpu.step(cpu.random_actions(35 + t))
        flags = {k: c[k].copy() for k in flag_keys}
        cpu.reset(mask=c["terminated"].copy())
        got = {k: v[t] for k, v in rows.items()}
        assert_outputs_equal(got, cpu.out, keys=keys, ctx=f"resident tick {t}")
        assert_outputs_equal(got, flags, keys=flag_keys, ctx=f"resident tick {t}")
    assert_state_equal(to_np(gpu.get_state()), cpu.get_state(), ctx="resident launch")
    assert gpu.device_errors() == 0
    gpu.close()


@pytest.mark.parametrize("item_cap,chunks,chunks_tick", [(None, None, None), (None, "3", "3"), ("200", "3", "3"), ("120", "3", "3"), ("48", None, None), (None, "2", "1"), (None, "1", "3")])
def test_several_chunks_of_a_slot_in_one_work_unit_on_the_dense_map(item_cap, chunks, chunks_tick, monkeypatch):
    """Chunk form (agh-map): the 64-ray chunks of a slot are traced as ONE work unit with one item list (fan_slot: shape queries of all chunks in shared rounds
    of 64 items) instead of chunk by chunk.  Held against the oracle as cat_create sets it up for the dense map (two chunks per unit, a list of 240 items),
    with all three chunks of a 2v1 slot in one unit (a second turn whenever they do not fit together), with shorter lists (200: two chunks + one; 120: one chunk
    per turn), with a list no chunk fits (48: every chunk is handed back to fan_chunk), and with different unit spans per entry."""
    import torch
    from as_cops_and_thieves_amd.config import SimConfig
    from as_cops_and_thieves_amd.maps import load_preset
    from as_cops_and_thieves_amd.sim import CatSim
    from oracle.cat_oracle import OracleSim
    for k, v in (("CAT_ITEM_CAP", item_cap), ("CAT_SLOT_CHUNKS", chunks), ("CAT_SLOT_CHUNKS_TICK", chunks_tick)):
        if v is not None:
            monkeypatch.setenv(k, v)
    N = 40
    cfg = SimConfig(n_envs=N, n_rays=64, max_step_count=30, seed=31)
    m = load_preset("agh-map").compile()
    stats = _run(cfg, [m], np.zeros(N, np.int32), ticks=45, rng=np.random.default_rng(12), auto_reset=True)
    assert stats["done"] >= N
    gpu, cpu = CatSim(cfg, [m], device="cuda:0"), OracleSim(cfg, [m])
    assert (gpu.chunks_per_unit(True), gpu.chunks_per_unit(False)) == (int(chunks or 2), int(chunks_tick or 2))   # fan_slot is what runs
    gpu.reset(); cpu.reset()
    rows = to_np(gpu.rollout_fused(50, None, tick=0, auto_reset=True))
    torch.cuda.synchronize()
    for t in range(50):
        c = cpu.step(cpu.random_actions(t))
        flags = {k: c[k].copy() for k in ("reward", "terminated", "truncated", "winner")}
        cpu.reset(mask=c["terminated"].copy())
        got = {k: v[t] for k, v in rows.items()}
        assert_outputs_equal(got, cpu.out, keys=("obs_distance", "obs_type", "shared_distance", "shared_type", "team_positions"), ctx=f"resident tick {t}")
        assert_outputs_equal(got, flags, keys=tuple(flags), ctx=f"resident tick {t}")
    assert_state_equal(to_np(gpu.get_state()), cpu.get_state(), ctx="resident launch")
    assert gpu.device_errors() == 0
    gpu.close()


def test_gate_off_parity():
    from as_cops_and_thieves_amd.config import SimConfig
    rng = np.random.default_rng(9)
    m = compiled("labyrinth")
    cfg = SimConfig(n_envs=8, n_rays=90, max_step_count=100, seed=2, bbtree_gate=0)
    _run(cfg, [m], None, ticks=30, rng=rng, spread=40.0)


def test_device_arithmetic_is_ieee_exact():
    """sqrt / divide / f16 conversions on the device equal the CPU's correctly rounded results."""
    import ctypes as C
    import torch
    from as_cops_and_thieves_amd import _native as nat
    from oracle import cat_oracle
    L, O = nat.lib(), cat_oracle.lib()
    rng = np.random.default_rng(0)
    n = 200000
    a = np.abs(rng.standard_normal(n) * 10.0 ** rng.integers(-8, 8, n))
    b = rng.standard_normal(n) * 10.0 ** rng.integers(-8, 8, n)
    ta, tb = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
    out = torch.empty_like(ta)
    for op, want in ((0, np.sqrt(a)), (1, a / b)):
        assert L.cat_selftest_arith(op, ta.data_ptr(), tb.data_ptr(), out.data_ptr(), n, 0, None) == 0
        torch.cuda.synchronize()
        assert np.array_equal(out.cpu().numpy().view(np.uint64), want.view(np.uint64)), f"op {op}"
    x = rng.uniform(-70000, 70000, n)
    tx = torch.from_numpy(x).cuda()
    assert L.cat_selftest_arith(2, tx.data_ptr(), tb.data_ptr(), out.data_ptr(), n, 0, None) == 0
    torch.cuda.synchronize()
    with np.errstate(over="ignore"):
        assert np.array_equal(out.cpu().numpy().astype(np.uint16), x.astype(np.float16).view(np.uint16))
    px, py = rng.uniform(-400, 400, n), rng.uniform(-400, 400, n)
    tpx, tpy = torch.from_numpy(px).cuda(), torch.from_numpy(py).cuda()
    assert L.cat_selftest_arith(3, tpx.data_ptr(), tpy.data_ptr(), out.data_ptr(), n, 0, None) == 0
    torch.cuda.synchronize()
    want = np.array([O.cato_obs_distance_f16(float(u), float(v), 0.0, 0.0) for u, v in zip(px[:20000], py[:20000])])
    assert np.array_equal(out.cpu().numpy()[:20000].astype(np.uint16), want.astype(np.uint16))


def test_fused_step_equals_three_separate_calls():
    """cat_step_fused(actions=NULL, tick, auto_reset) == cat_random_actions + cat_step + cat_reset_done."""
    import torch
    from as_cops_and_thieves_amd.config import SimConfig
    from as_cops_and_thieves_amd.sim import CatSim
    m = compiled("lbirinth")
    cfg = SimConfig(n_envs=96, n_rays=64, max_step_count=15, seed=13)
    a, b = CatSim(cfg, [m], device="cuda:0"), CatSim(cfg, [m], device="cuda:0")
    a.reset(); b.reset()
    for t in range(40):
        a.step(a.random_actions(t)); a.reset_done()
        b.step_fused(None, tick=t, auto_reset=True)
        torch.cuda.synchronize()
        for k in a.out:
            assert torch.equal(a.out[k], b.out[k]), (t, k)
    sa, sb = a.get_state(), b.get_state()
    for k in sa:
        assert torch.equal(sa[k], sb[k]), k
    assert int(sa["reset_count"].min()) >= 2
    a.close(); b.close()


def _arena_map(tmp_path, n_cops, n_thieves):
    import json
    from as_cops_and_thieves_amd.maps import Map
    blocks = [{"type": "rect", "x": 100, "y": 100, "w": 5, "h": 400}, {"type": "rect", "x": 100, "y": 500, "w": 400, "h": 5},
              {"type": "rect", "x": 500, "y": 100, "w": 5, "h": 405}, {"type": "rect", "x": 100, "y": 100, "w": 400, "h": 5},
              {"type": "poly", "vs": [{"x": 250, "y": 250}, {"x": 330, "y": 270}, {"x": 350, "y": 330}, {"x": 300, "y": 370},
                                      {"x": 240, "y": 340}, {"x": 230, "y": 290}]},
              {"type": "poly", "vs": [{"x": 400, "y": 150}, {"x": 450, "y": 160}, {"x": 430, "y": 220}]}]
    region = {"x": 115, "y": 115, "w": 370, "h": 370}
    agents = [{"type": "cop", "x": 130 + 25 * i, "y": 130, "spawn_region": region} for i in range(n_cops)] + \
             [{"type": "thief", "x": 130 + 25 * i, "y": 470, "spawn_region": region} for i in range(n_thieves)]
    f = tmp_path / "arena.json"
    f.write_text(json.dumps({"window": {"w_px": 640, "h_px": 640}, "canvas": {"w": 640, "h": 640},
                             "objects": {"blocks": blocks}, "agents": agents}))
    return Map(f).compile()


def test_maximum_roster_eight_agents(tmp_path):
    """A = 8 (CAT_MAX_AGENTS): 64 pair cones, 64 wall-cache slots, 28 agent pairs; crowded arena -> many contacts."""
    from as_cops_and_thieves_amd.config import SimConfig
    m = _arena_map(tmp_path, 5, 3)
    cfg = SimConfig(n_envs=24, n_cops=5, n_thieves=3, n_rays=32, max_step_count=40, seed=31)
    stats = _run(cfg, [m], np.zeros(24, np.int32), ticks=90, rng=np.random.default_rng(0), auto_reset=True)
    assert stats["contacts"] > 0 and stats["done"] >= 24


def test_many_rays_not_a_multiple_of_the_wave(tmp_path):
    """R = 200: four ray chunks per agent with a ragged tail, more items than one pass holds."""
    from as_cops_and_thieves_amd.config import SimConfig
    m = _arena_map(tmp_path, 2, 1)
    cfg = SimConfig(n_envs=8, n_cops=2, n_thieves=1, n_rays=200, max_step_count=30, seed=17)
    _run(cfg, [m], np.zeros(8, np.int32), ticks=50, rng=np.random.default_rng(1), auto_reset=True)


def test_gate_off_on_dense_map():
    from as_cops_and_thieves_amd.config import SimConfig
    m = compiled("agh-map")
    cfg = SimConfig(n_envs=8, n_rays=90, max_step_count=40, seed=4, bbtree_gate=0)
    _run(cfg, [m], None, ticks=30, rng=np.random.default_rng(2), spread=60.0)


def _picket_map(tmp_path, n_posts):
    """A row of thin posts: a ray shot along the row has every post as a candidate (bb gate passes them all)."""
    import json
    from as_cops_and_thieves_amd.maps import Map
    blocks = [{"type": "rect", "x": 60 + 9 * q, "y": 195 + (q % 2), "w": 3, "h": 12} for q in range(n_posts)]
    blocks += [{"type": "rect", "x": 10, "y": 10, "w": 5, "h": 380}, {"type": "rect", "x": 10, "y": 390, "w": 620, "h": 5},
               {"type": "rect", "x": 630, "y": 10, "w": 5, "h": 385}, {"type": "rect", "x": 10, "y": 10, "w": 620, "h": 5}]
    region = {"x": 20, "y": 150, "w": 30, "h": 100}
    agents = [{"type": "cop", "x": 30, "y": 200, "spawn_region": region},
              {"type": "cop", "x": 35, "y": 230, "spawn_region": {"x": 20, "y": 30, "w": 600, "h": 100}},
              {"type": "thief", "x": 40, "y": 170, "spawn_region": {"x": 20, "y": 280, "w": 600, "h": 100}}]
    f = tmp_path / "picket.json"
    f.write_text(json.dumps({"window": {"w_px": 640, "h_px": 400}, "canvas": {"w": 640, "h": 400},
                             "objects": {"blocks": blocks}, "agents": agents}))
    return Map(f).compile()


@pytest.mark.parametrize("occlusion", [0, 1])
def test_more_than_31_candidate_walls_on_one_ray(tmp_path, monkeypatch, occlusion):
    """Packed ray-grid rows hold 31 ids; 40 posts in a row force the CSR continuation of the candidate list -- with the table's
    occlusion rule switched off (CAT_GRID_OCCLUSION=0: every post behind the first is listed); with it on the same map runs with
    short lists, and both must agree with the oracle, which visits all 44 walls."""
    import ctypes as C
    from as_cops_and_thieves_amd.config import SimConfig
    monkeypatch.setenv("CAT_GRID_OCCLUSION", str(occlusion))
    m = _picket_map(tmp_path, 40)
    cfg = SimConfig(n_envs=16, n_cops=2, n_thieves=1, n_rays=64, max_step_count=60, seed=5)
    gpu, cpu = _pair(cfg, [m], np.zeros(16, np.int32))
    out = (C.c_int * 256)()
    from as_cops_and_thieves_amd import _native as nat
    longest = max(nat.lib().cat_debug_grid_lookup(gpu._h, 0, 30.0, 201.0, k, out, 256) for k in range(64))
    gpu.close()
    assert (longest > 31) if occlusion == 0 else (longest <= 31), longest
    _run(cfg, [m], np.zeros(16, np.int32), ticks=70, rng=np.random.default_rng(3), auto_reset=True)


def test_wall_with_too_many_hull_edges_is_rejected(tmp_path):
    import json, math
    from as_cops_and_thieves_amd.config import SimConfig
    from as_cops_and_thieves_amd.maps import Map
    from as_cops_and_thieves_amd.sim import CatSim
    vs = [{"x": 300 + 100 * math.cos(2 * math.pi * q / 40), "y": 300 + 100 * math.sin(2 * math.pi * q / 40)} for q in range(40)]
    agents = [{"type": "cop", "x": 50, "y": 50}, {"type": "cop", "x": 80, "y": 50}, {"type": "thief", "x": 50, "y": 550}]
    f = tmp_path / "round.json"
    f.write_text(json.dumps({"window": {"w_px": 640, "h_px": 640}, "canvas": {"w": 640, "h": 640},
                             "objects": {"blocks": [{"type": "poly", "vs": vs}]}, "agents": agents}))
    with pytest.raises(RuntimeError, match="hull edges"):
        CatSim(SimConfig(n_envs=4, n_rays=16), [Map(f).compile()], device="cuda:0")


def test_map_needing_more_than_eight_cached_wall_contacts_is_rejected(tmp_path):
    """cat_create holds the map against CAT_WALL_CACHE (tests/test_maps_and_constants.py has the bound itself): 9 thin walls
    meeting at a point are refused with a message, 8 are accepted and run."""
    from tests.test_maps_and_constants import _star_junction
    from as_cops_and_thieves_amd.config import SimConfig
    from as_cops_and_thieves_amd.sim import CatSim
    with pytest.raises(RuntimeError, match="9 walls at once"):
        CatSim(SimConfig(n_envs=4, n_rays=16), [_star_junction(tmp_path, 9)], device="cuda:0")
    m = _star_junction(tmp_path, 8)
    cfg = SimConfig(n_envs=8, n_rays=32, max_step_count=30, seed=4)
    _run(cfg, [m], np.zeros(8, np.int32), ticks=40, rng=np.random.default_rng(1), auto_reset=True)


@pytest.mark.parametrize("wpb", [1, 2, 4, 8, 16])
def test_every_workgroup_size_gives_the_same_bits(wpb, monkeypatch):
    """The waves of a workgroup share ray chunks and physics steps through LDS counters; the workgroup size is
    picked per sim (16 for short launches, 4 for long ones).  Every size must reproduce the oracle, with env counts
    that leave ragged workgroups and with two maps (workgroups are map-homogeneous)."""
    from as_cops_and_thieves_amd.config import SimConfig
    monkeypatch.setenv("CAT_WAVES_PER_BLOCK", str(wpb))
    maps = [compiled("labyrinth"), compiled("squarinth")]
    n = 37
    slot = (np.arange(n) % 2).astype(np.int32)
    cfg = SimConfig(n_envs=n, n_rays=64, max_step_count=25, seed=11)
    stats = _run(cfg, maps, slot, ticks=60, rng=np.random.default_rng(5), auto_reset=True)
    assert stats["done"] >= n


@pytest.mark.parametrize("name,rays,cops,thieves", [("labyrinth", 64, 2, 1), ("squarinth", 90, 2, 1), ("grandbyrinth", 64, 3, 2)])
def test_generic_kernel_on_the_rosters_that_have_a_fixed_instantiation(name, rays, cops, thieves, monkeypatch):
    """(agents, rays, cops) = (3, 64, 2), (3, 90, 2), (5, 64, 3) run compile-time-dimension instantiations of the
    kernels; CAT_GENERIC_KERNEL=1 forces the generic one.  Both must reproduce the oracle (the default path is
    what every other test in this file exercises)."""
    from as_cops_and_thieves_amd.config import SimConfig
    monkeypatch.setenv("CAT_GENERIC_KERNEL", "1")
    m = compiled(name, cops, thieves)
    cfg = SimConfig(n_envs=24, n_cops=cops, n_thieves=thieves, n_rays=rays, max_step_count=30, seed=23)
    _run(cfg, [m], np.zeros(24, np.int32), ticks=45, rng=np.random.default_rng(9), auto_reset=True)


def test_one_cop_vs_one_thief_on_squarinth():
    """BASELINE configs[0]'s roster (the reference's plumbing case): A = 2, one agent pair, 90 rays, generic kernel."""
    from as_cops_and_thieves_amd.config import SimConfig
    m = compiled("squarinth", 1, 1)
    cfg = SimConfig(n_envs=20, n_cops=1, n_thieves=1, n_rays=90, max_step_count=35, seed=13)
    stats = _run(cfg, [m], np.zeros(20, np.int32), ticks=80, rng=np.random.default_rng(11), auto_reset=True)
    assert stats["done"] >= 20


def test_maximum_rays_and_agents_together(tmp_path):
    """CAT_MAX_AGENTS x CAT_MAX_RAYS: 8 agents, 512 rays (64 ray chunks per env, 18 KB of output staging per slot:
    the workgroup shrinks to fit LDS)."""
    from as_cops_and_thieves_amd.config import SimConfig
    m = _arena_map(tmp_path, 4, 4)
    cfg = SimConfig(n_envs=6, n_cops=4, n_thieves=4, n_rays=512, max_step_count=20, seed=3)
    _run(cfg, [m], np.zeros(6, np.int32), ticks=25, rng=np.random.default_rng(4), auto_reset=True)


def test_empty_inputs():
    """No env to create is a configuration error (clear message, no crash); a reset whose mask selects nothing
    changes nothing; a step of a single env works (one busy wave, fifteen helpers)."""
    import torch
    from as_cops_and_thieves_amd.config import SimConfig
    from as_cops_and_thieves_amd.sim import CatSim
    m = compiled("squarinth")
    with pytest.raises(RuntimeError):
        CatSim(SimConfig(n_envs=0, n_rays=16), [m], device="cuda:0")
    gpu, cpu = _pair(SimConfig(n_envs=1, n_rays=90, max_step_count=30, seed=8), [m])
    g = gpu.reset(); c = cpu.reset()
    for t in range(5):
        a = cpu.random_actions(t)
        g = gpu.step(torch.from_numpy(a)); c = cpu.step(a)
    before = {k: v.clone() for k, v in gpu.get_state().items()}
    obs_before = gpu.out["obs_distance"].clone()
    gpu.reset(mask=torch.zeros(1, dtype=torch.uint8, device="cuda:0"))
    torch.cuda.synchronize()
    after = gpu.get_state()
    assert all(torch.equal(before[k], after[k]) for k in before) and torch.equal(obs_before, gpu.out["obs_distance"])
    assert_outputs_equal(to_np(g), c, ctx="single env")
    gpu.close()


@pytest.mark.parametrize("name", ["labyrinth", "agh-map"])
def test_long_rollout_soak(name):
    """2000 ticks with auto-reset, compared every 50 ticks and at the end: rare paths (arbiter cache eviction and
    ageing, many captures and respawns, grazing rays) accumulate over a long rollout."""
    from as_cops_and_thieves_amd.config import SimConfig
    m = compiled(name)
    cfg = SimConfig(n_envs=32, n_rays=64, max_step_count=120, seed=29)
    stats = _run(cfg, [m], np.zeros(32, np.int32), ticks=2000, rng=np.random.default_rng(7), check_every=50, auto_reset=True)
    assert stats["done"] >= 32 * 10


def test_long_rollout_soak_single_launch_step():
    """The one-launch rollout step (in-kernel Philox actions, in-kernel auto-reset) against oracle step + masked reset,
    1500 ticks on the mixed five-map batch."""
    import torch
    from as_cops_and_thieves_amd.config import SimConfig
    names = ["agh-map", "grandbyrinth", "labyrinth", "lbirinth", "squarinth"]
    maps = [compiled(n) for n in names]
    n = 40
    slot = (np.arange(n) % len(maps)).astype(np.int32)
    cfg = SimConfig(n_envs=n, n_rays=64, max_step_count=90, seed=41)
    gpu, cpu = _pair(cfg, maps, slot)
    gpu.reset(); cpu.reset()
    obs_keys = ("obs_distance", "obs_type", "hit_shape", "shared_distance", "shared_type", "team_positions")
    done = 0
    for t in range(1500):
        g = gpu.step_fused(None, tick=t, auto_reset=True)
        c = cpu.step(cpu.random_actions(t))
        term = c["terminated"].copy()
        done += int(term.sum())
        keep = {k: c[k].copy() for k in ("reward", "terminated", "truncated", "winner")}
        cpu.reset(mask=term)
        if t % 100 == 0 or t == 1499:
            torch.cuda.synchronize()
            assert_outputs_equal(to_np(g), {**cpu.out, **keep}, keys=obs_keys + tuple(keep), ctx=f"tick {t}")
            assert_state_equal(to_np(gpu.get_state()), cpu.get_state(), ctx=f"tick {t}")
    assert done >= n * 10
    assert gpu.device_errors() == 0          # no wall contact was dropped for want of a cache slot, no bad action flagged
    gpu.close()
