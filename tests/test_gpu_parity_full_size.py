"""GPU: whole-batch parity at the BASELINE.json sizes -- EVERY env slot, every output and the whole f64 state, bit for bit
against the CPU oracle (all host cores), not a sample of slots.  The GPU side takes the production rollout step
(cat_step_fused: in-kernel Philox actions + tick + in-kernel auto-reset, one launch); the oracle takes the three separate calls
(random actions, step, masked reset).  max_step_count is short so that every slot passes through the auto-reset several times.
Oracle parity against Pymunk itself is UNPINNED (oracle/cat_oracle.h)."""
import os

import numpy as np
import pytest

from tests.util import assert_outputs_equal, assert_state_equal, compiled, to_np

pytestmark = pytest.mark.gpu

FIVE = ["agh-map", "grandbyrinth", "labyrinth", "lbirinth", "squarinth"]
OBS_KEYS = ("obs_distance", "obs_type", "hit_shape", "shared_distance", "shared_type", "team_positions")


@pytest.mark.parametrize("label,names,cops,thieves,N,ticks,max_steps,check_every", [
    ("configs[1] labyrinth 2v1 x4096", ["labyrinth"], 2, 1, 4096, 64, 25, 4),
    ("configs[2] per-GPU shard: agh-map 2v1 x4096", ["agh-map"], 2, 1, 4096, 60, 25, 6),
    ("configs[3] grandbyrinth 3v2 x8192", ["grandbyrinth"], 3, 2, 8192, 60, 25, 6),
    ("configs[4] five maps mixed 2v1 x16384", FIVE, 2, 1, 16384, 60, 25, 10),
])
def test_every_slot_matches_the_oracle(label, names, cops, thieves, N, ticks, max_steps, check_every):
    import torch
    from as_cops_and_thieves_amd.config import SimConfig
    from as_cops_and_thieves_amd.sim import CatSim
    from oracle import cat_oracle
    from oracle.cat_oracle import OracleSim
    maps = [compiled(n, cops, thieves) for n in names]
    slot = (np.arange(N) % len(maps)).astype(np.int32) if len(maps) > 1 else None
    cfg = SimConfig(n_envs=N, n_cops=cops, n_thieves=thieves, n_rays=64, max_step_count=max_steps, seed=20261004)
    threads = max(1, min(16, len(os.sched_getaffinity(0))))
    cat_oracle.lib().cato_set_threads(threads)
    try:
        gpu = CatSim(cfg, maps, slot, device="cuda:0", debug_hit_shape=True)
        cpu = OracleSim(cfg, maps, slot)
        g, c = gpu.reset(), cpu.reset()
        torch.cuda.synchronize()
        assert_outputs_equal(to_np(g), c, keys=OBS_KEYS, ctx=f"{label}: reset")
        assert_state_equal(to_np(gpu.get_state()), cpu.get_state(), ctx=f"{label}: reset")
        captured = contacts = 0
        for t in range(ticks):
            gpu.step_fused(None, tick=t, auto_reset=True)
            c = cpu.step(cpu.random_actions(t))
            flags = {k: c[k].copy() for k in ("reward", "terminated", "truncated", "winner")}
            captured += int((c["winner"] == 0).sum())
            cpu.reset(mask=c["terminated"].copy())          # overwrites the observations of the slots that restarted
            if t % check_every == 0 or t == ticks - 1:
                torch.cuda.synchronize()
                got = to_np(gpu.out)
                assert_outputs_equal(got, cpu.out, keys=OBS_KEYS, ctx=f"{label}: tick {t}")
                assert_outputs_equal(got, flags, keys=tuple(flags), ctx=f"{label}: tick {t}")
                st = cpu.get_state()
                assert_state_equal(to_np(gpu.get_state()), st, ctx=f"{label}: tick {t}")
                contacts += int((st["wall_shape"] >= 0).sum() + (st["pair_age"] >= 0).sum())
        assert int(cpu.get_state()["reset_count"].min()) >= 2      # every slot restarted at least once after the first reset
        assert contacts > 0, (captured, contacts)   # cached arbiters occurred (captures need longer episodes than these: the
        #                                              small-batch parity tests place thieves next to cops for them)
        assert gpu.device_errors() == 0
        gpu.close()
    finally:
        cat_oracle.lib().cato_set_threads(1)


@pytest.mark.parametrize("label,names,cops,thieves,N,switch", [
    ("configs[3] grandbyrinth 3v2 x8192 with the ring in (CAT_POOL=1: both entries pooled)", ["grandbyrinth"], 3, 2, 8192, ("CAT_POOL", "1")),
    ("configs[4] five maps x16384 in two parts (CAT_SPLIT=1: one dispatch per fan form on two streams)", FIVE, 2, 1, 16384, ("CAT_SPLIT", "1")),
])
def test_the_optional_schedulers_at_full_size(label, names, cops, thieves, N, switch, monkeypatch):
    """What cat_create does not choose by itself, at the BASELINE sizes: the pooled kernels for the 3v2 roster (the ring of exact capacity beside group arrays
    for two agents) and the mixed batch cut into a pooled group-form part and a chunk-form part -- every slot, outputs and state, against the oracle through
    one-tick launches, then one resident launch for the second half of the ticks."""
    import torch
    from oracle import cat_oracle
    monkeypatch.setenv(*switch)
    threads = max(1, min(16, len(os.sched_getaffinity(0))))
    cat_oracle.lib().cato_set_threads(threads)
    try:
        gpu, cpu = _full_pair(names, cops, thieves, N, 64, 25, seed=20261006)
        want = {"CAT_POOL": ("step_kernel_pooled", "rollout_kernel_pooled"), "CAT_SPLIT": ("step_kernel_pooled+step_kernel", "rollout_kernel_pooled+rollout_kernel")}[switch[0]]
        assert (gpu.one_tick_kernel, gpu.rollout_kernel) == want
        g, c = gpu.reset(), cpu.reset()
        torch.cuda.synchronize()
        assert_outputs_equal(to_np(g), c, keys=OBS_KEYS, ctx=f"{label}: reset")
        _lockstep(gpu, cpu, 60, set(range(0, 60, 6)) | {59}, label, resident_from=30)
        assert int(cpu.get_state()["reset_count"].min()) >= 2
        gpu.close()
    finally:
        cat_oracle.lib().cato_set_threads(1)


def test_configs2_at_its_stated_size_and_a_shard_at_its_real_offset():
    """BASELINE configs[2] as stated: agh-map 2v1, 32768 envs.  (i) all of them in ONE batch on one GPU, 60 ticks of 25-tick episodes, every slot against the
    oracle; (ii) the shard rank 7 of an 8-GPU job owns -- 4096 slots at env_id_offset 28672, which key their Philox streams with the GLOBAL env ids -- run as
    its own sim and compared with slots 28672 ... of that batch (outputs and state), i.e. with what one big batch simulates for the same envs."""
    import torch
    from dataclasses import replace
    from as_cops_and_thieves_amd.config import SimConfig
    from as_cops_and_thieves_amd.sim import CatSim
    from oracle import cat_oracle
    from oracle.cat_oracle import OracleSim
    N, n_shard, off, ticks = 32768, 4096, 28672, 60
    m = compiled("agh-map", 2, 1)
    cfg = SimConfig(n_envs=N, n_cops=2, n_thieves=1, n_rays=64, max_step_count=25, seed=20261005)
    threads = max(1, min(16, len(os.sched_getaffinity(0))))
    cat_oracle.lib().cato_set_threads(threads)
    try:
        gpu = CatSim(cfg, [m], device="cuda:0", debug_hit_shape=True)
        shard = CatSim(replace(cfg, n_envs=n_shard, env_id_offset=off), [m], device="cuda:0", debug_hit_shape=True)
        cpu = OracleSim(cfg, [m])
        g, c = gpu.reset(), cpu.reset()
        shard.reset()
        torch.cuda.synchronize()
        assert_outputs_equal(to_np(g), c, keys=OBS_KEYS, ctx="configs[2] x32768: reset")
        flag_keys = ("reward", "terminated", "truncated", "winner")
        for t in range(ticks):
            gpu.step_fused(None, tick=t, auto_reset=True)
            shard.step_fused(None, tick=t, auto_reset=True)
            c = cpu.step(cpu.random_actions(t))
            flags = {k: c[k].copy() for k in flag_keys}
            cpu.reset(mask=c["terminated"].copy())
            if t % 10 == 0 or t == ticks - 1:
                torch.cuda.synchronize()
                got = to_np(gpu.out)
                assert_outputs_equal(got, cpu.out, keys=OBS_KEYS, ctx=f"configs[2] x32768: tick {t}")
                assert_outputs_equal(got, flags, keys=flag_keys, ctx=f"configs[2] x32768: tick {t}")
                st = to_np(gpu.get_state())
                assert_state_equal(st, cpu.get_state(), ctx=f"configs[2] x32768: tick {t}")
                sh_out, sh_st = to_np(shard.out), to_np(shard.get_state())
                assert_outputs_equal(sh_out, {k: v[off:off + n_shard] for k, v in got.items()}, ctx=f"shard at offset {off}: tick {t}")
                assert_state_equal(sh_st, {k: v[off:off + n_shard] for k, v in st.items()}, ctx=f"shard at offset {off}: tick {t}")
        assert int(cpu.get_state()["reset_count"].min()) >= 2
        assert gpu.device_errors() == 0 and shard.device_errors() == 0
        gpu.close(); shard.close()
    finally:
        cat_oracle.lib().cato_set_threads(1)


def _full_pair(names, cops, thieves, N, rays, max_steps, seed=0):
    from as_cops_and_thieves_amd.config import SimConfig
    from as_cops_and_thieves_amd.sim import CatSim
    from oracle.cat_oracle import OracleSim
    maps = [compiled(n, cops, thieves) for n in names]
    slot = (np.arange(N) % len(maps)).astype(np.int32) if len(maps) > 1 else None
    cfg = SimConfig(n_envs=N, n_cops=cops, n_thieves=thieves, n_rays=rays, max_step_count=max_steps, seed=seed)
    return CatSim(cfg, maps, slot, device="cuda:0", debug_hit_shape=True), OracleSim(cfg, maps, slot)


def _lockstep(gpu, cpu, ticks, checks, label, resident_from=None):
    """cat_step_fused tick by tick on the GPU (from tick `resident_from` on: ONE resident launch for the remaining ticks, compared
    at the same ticks), random actions + step + masked reset on the oracle; every slot compared at the ticks in `checks`."""
    import torch
    flag_keys = ("reward", "terminated", "truncated", "winner")
    rows = None
    for t in range(ticks):
        if resident_from is not None and t == resident_from:
            rows = to_np(gpu.rollout_fused(ticks - t, None, tick=t, auto_reset=True))
        elif rows is None:
            gpu.step_fused(None, tick=t, auto_reset=True)
        c = cpu.step(cpu.random_actions(t))
        flags = {k: c[k].copy() for k in flag_keys}
        cpu.reset(mask=c["terminated"].copy())
        if t in checks:
            torch.cuda.synchronize()
            got = to_np(gpu.out) if rows is None else {k: v[t - resident_from] for k, v in rows.items()}
            assert_outputs_equal(got, cpu.out, keys=OBS_KEYS, ctx=f"{label}: tick {t}")
            assert_outputs_equal(got, flags, keys=flag_keys, ctx=f"{label}: tick {t}")
            if rows is None:
                assert_state_equal(to_np(gpu.get_state()), cpu.get_state(), ctx=f"{label}: tick {t}")
    torch.cuda.synchronize()
    assert_state_equal(to_np(gpu.get_state()), cpu.get_state(), ctx=f"{label}: after tick {ticks - 1}")
    assert gpu.device_errors() == 0


@pytest.mark.parametrize("name", ["labyrinth", "agh-map"])
def test_the_regime_the_bench_times(name):
    """bench.py's own sequence on its own batch: reset, 600 ticks of cat_step_fused with 400-tick episodes (every slot times out
    together at tick 400 and restarts), then the 25 ticks the driver's command times (--warmup 5 --steps 20) -- every slot, outputs
    and the whole state against the oracle at the episode end (ticks 399 - 401), mid-episode (599 - 600) and through the timed ticks.
    Agents have spread over the map by then (table rows that miss L2, contacts, long candidate lists), which the 25-tick
    episodes of the tests above never reach."""
    import torch
    from oracle import cat_oracle
    threads = max(1, min(16, len(os.sched_getaffinity(0))))
    cat_oracle.lib().cato_set_threads(threads)
    try:
        gpu, cpu = _full_pair([name], 2, 1, 4096, 64, 400, seed=0)
        g, c = gpu.reset(), cpu.reset()
        torch.cuda.synchronize()
        assert_outputs_equal(to_np(g), c, keys=OBS_KEYS, ctx=f"{name}: reset")
        _lockstep(gpu, cpu, 625, {0, 200, 398, 399, 400, 401, 599, 600, 605, 612, 619, 624}, f"bench regime, {name} x4096")
        assert int(cpu.get_state()["reset_count"].min()) >= 2
        gpu.close()
    finally:
        cat_oracle.lib().cato_set_threads(1)


@pytest.mark.parametrize("name,rays,ticks,max_steps", [
    ("labyrinth", 90, 60, 25),          # the reference's own sensor (entity.py:86) at the BASELINE batch size
    ("agh-map", 90, 60, 25),
    ("labyrinth-inside", 64, 60, 25),   # every agent spawned inside the maze
])
def test_reference_sensor_and_inside_spawns_at_full_batch(name, rays, ticks, max_steps):
    import torch
    from oracle import cat_oracle
    threads = max(1, min(16, len(os.sched_getaffinity(0))))
    cat_oracle.lib().cato_set_threads(threads)
    try:
        gpu, cpu = _full_pair([name], 2, 1, 4096, rays, max_steps, seed=20261004)
        g, c = gpu.reset(), cpu.reset()
        torch.cuda.synchronize()
        assert_outputs_equal(to_np(g), c, keys=OBS_KEYS, ctx=f"{name}: reset")
        # the second half of the ticks in ONE resident launch (cat_rollout_fused), compared row by row
        _lockstep(gpu, cpu, ticks, set(range(0, ticks, 5)) | {ticks - 1}, f"{name} x4096, {rays} rays", resident_from=ticks // 2)
        assert int(cpu.get_state()["reset_count"].min()) >= 2
        gpu.close()
    finally:
        cat_oracle.lib().cato_set_threads(1)


@pytest.mark.parametrize("name,rays", [("labyrinth", 64), ("agh-map", 64), ("agh-map", 90)])
def test_soak_128_envs_3000_ticks(name, rays):
    """tools/soak_parity.py's rows as a test: 128 envs x 3000 ticks with 90-tick episodes (respawns all over the map, ~4 000
    episodes), outputs and state compared every 10 ticks; the last 1000 ticks run as resident launches of 50 ticks."""
    import torch
    from oracle import cat_oracle
    threads = max(1, min(16, len(os.sched_getaffinity(0))))
    cat_oracle.lib().cato_set_threads(threads)
    try:
        gpu, cpu = _full_pair([name], 2, 1, 128, rays, 90, seed=41)
        gpu.reset(); cpu.reset()
        flag_keys = ("reward", "terminated", "truncated", "winner")
        done = 0
        t = 0
        while t < 3000:
            span = 1 if t < 2000 else 50
            rows = to_np(gpu.rollout_fused(span, None, tick=t, auto_reset=True)) if span > 1 else None
            if rows is None:
                gpu.step_fused(None, tick=t, auto_reset=True)
            for q in range(span):
                c = cpu.step(cpu.random_actions(t + q))
                flags = {k: c[k].copy() for k in flag_keys}
                done += int(c["terminated"].sum())
                cpu.reset(mask=c["terminated"].copy())
                if (t + q) % 10 == 0:
                    torch.cuda.synchronize()
                    got = to_np(gpu.out) if rows is None else {k: v[q] for k, v in rows.items()}
                    assert_outputs_equal(got, cpu.out, keys=OBS_KEYS, ctx=f"soak {name}: tick {t + q}")
                    assert_outputs_equal(got, flags, keys=flag_keys, ctx=f"soak {name}: tick {t + q}")
                    if rows is None:
                        assert_state_equal(to_np(gpu.get_state()), cpu.get_state(), ctx=f"soak {name}: tick {t + q}")
            t += span
            if rows is not None:
                torch.cuda.synchronize()
                assert_state_equal(to_np(gpu.get_state()), cpu.get_state(), ctx=f"soak {name}: after tick {t - 1}")
        assert done >= 128 * 25
        assert gpu.device_errors() == 0
        gpu.close()
    finally:
        cat_oracle.lib().cato_set_threads(1)
