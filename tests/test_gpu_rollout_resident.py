"""GPU: the resident T-tick rollout launch (cat_rollout_fused) against T launches of the one-tick step (cat_step_fused) and
against the CPU oracle -- every env slot, every tick's outputs (rows of the [T, N, ...] buffers) and the whole f64 state
afterwards, bit for bit.  Oracle parity against Pymunk itself is UNPINNED (oracle/cat_oracle.h)."""
import os

import numpy as np
import pytest

from tests.util import OUT_KEYS, assert_outputs_equal, assert_state_equal, compiled, to_np

pytestmark = pytest.mark.gpu

FIVE = ["agh-map", "grandbyrinth", "labyrinth", "lbirinth", "squarinth"]
OBS_KEYS = ("obs_distance", "obs_type", "hit_shape", "shared_distance", "shared_type", "team_positions")
FLAG_KEYS = ("reward", "terminated", "truncated", "winner")


def _pair(names, cops, thieves, N, rays, max_steps, seed=77):
    from as_cops_and_thieves_amd.config import SimConfig
    from as_cops_and_thieves_amd.sim import CatSim
    maps = [compiled(n, cops, thieves) for n in names]
    slot = (np.arange(N) % len(maps)).astype(np.int32) if len(maps) > 1 else None
    cfg = SimConfig(n_envs=N, n_cops=cops, n_thieves=thieves, n_rays=rays, max_step_count=max_steps, seed=seed)
    return cfg, maps, slot, CatSim(cfg, maps, slot, device="cuda:0", debug_hit_shape=True)


@pytest.mark.parametrize("names,cops,thieves,N,rays,T,max_steps", [
    (["squarinth"], 2, 1, 37, 64, 40, 11),          # ragged batch: the last workgroup has empty slots
    (["labyrinth"], 2, 1, 256, 64, 48, 17),
    (["agh-map"], 2, 1, 64, 90, 24, 9),             # chunk form, the reference's sensor
    (["grandbyrinth"], 3, 2, 128, 64, 30, 13),
    (FIVE, 2, 1, 320, 64, 30, 12),                  # map-homogeneous workgroups of a mixed batch
    (["lbirinth"], 1, 1, 16, 33, 20, 7),            # generic instantiation
])
def test_resident_rollout_equals_one_launch_per_tick(names, cops, thieves, N, rays, T, max_steps):
    import torch
    cfg, maps, slot, a = _pair(names, cops, thieves, N, rays, max_steps)
    from as_cops_and_thieves_amd.sim import CatSim
    b = CatSim(cfg, maps, slot, device="cuda:0", debug_hit_shape=True)
    a.reset(); b.reset()
    # synthetic actions, then an explicit action tape
    for tape in (False, True):
        acts = None
        if tape:
            acts = torch.stack([a.random_actions(1000 + t) for t in range(T)])
        rows = a.rollout_fused(T, acts, tick=5, auto_reset=True)
        torch.cuda.synchronize()
        rows = {k: to_np({k: v})[k] for k, v in rows.items()}
        for t in range(T):
            b.step_fused(None if acts is None else acts[t], tick=5 + t, auto_reset=True)
            torch.cuda.synchronize()
            assert_outputs_equal({k: v[t] for k, v in rows.items()}, to_np(b.out), keys=OUT_KEYS, ctx=f"{names} tape={tape} tick {t}")
        assert_state_equal(to_np(a.get_state()), to_np(b.get_state()), ctx=f"{names} tape={tape}: state after the rollout")
    assert int(to_np(a.get_state())["reset_count"].min()) >= 2
    assert a.device_errors() == 0
    a.close(); b.close()


def test_rollout_without_auto_reset_and_single_tick():
    import torch
    cfg, maps, slot, a = _pair(["squarinth"], 2, 1, 48, 64, 6)
    from as_cops_and_thieves_amd.sim import CatSim
    b = CatSim(cfg, maps, slot, device="cuda:0", debug_hit_shape=True)
    a.reset(); b.reset()
    rows = a.rollout_fused(10, None, tick=0, auto_reset=False)     # episodes end at tick 6 and are NOT restarted
    torch.cuda.synchronize()
    rows = {k: to_np({k: v})[k] for k, v in rows.items()}
    for t in range(10):
        b.step_fused(None, tick=t, auto_reset=False)
        torch.cuda.synchronize()
        assert_outputs_equal({k: v[t] for k, v in rows.items()}, to_np(b.out), keys=OUT_KEYS, ctx=f"no auto-reset, tick {t}")
    assert rows["truncated"][5:].all()
    one = a.rollout_fused(1, None, tick=10, auto_reset=True)       # T = 1 is the one-tick step
    b.step_fused(None, tick=10, auto_reset=True)
    torch.cuda.synchronize()
    assert_outputs_equal({k: v[0] for k, v in to_np(one).items()}, to_np(b.out), keys=OUT_KEYS, ctx="T = 1")
    assert_state_equal(to_np(a.get_state()), to_np(b.get_state()), ctx="T = 1")
    with pytest.raises(Exception):
        a.rollout_fused(0)
    a.close(); b.close()


@pytest.mark.parametrize("label,names,cops,thieves,N,T,max_steps", [
    ("configs[1] labyrinth 2v1 x4096", ["labyrinth"], 2, 1, 4096, 64, 25),
    ("configs[2] per-GPU shard: agh-map 2v1 x4096", ["agh-map"], 2, 1, 4096, 48, 25),
    ("configs[3] grandbyrinth 3v2 x8192", ["grandbyrinth"], 3, 2, 8192, 48, 25),
    ("configs[4] five maps mixed 2v1 x16384", FIVE, 2, 1, 16384, 40, 25),
])
def test_resident_rollout_matches_the_oracle_on_every_slot(label, names, cops, thieves, N, T, max_steps):
    """BASELINE sizes: the [T, N, ...] rows of ONE resident launch against the oracle stepped tick by tick (random actions,
    step, masked reset), every slot, every tick; then the state.  The oracle runs on the host's CPU share."""
    import torch
    from oracle import cat_oracle
    from oracle.cat_oracle import OracleSim
    cfg, maps, slot, gpu = _pair(names, cops, thieves, N, 64, max_steps, seed=20261004)
    threads = max(1, min(16, len(os.sched_getaffinity(0))))
    cat_oracle.lib().cato_set_threads(threads)
    try:
        cpu = OracleSim(cfg, maps, slot)
        g, c = gpu.reset(), cpu.reset()
        torch.cuda.synchronize()
        assert_outputs_equal(to_np(g), c, keys=OBS_KEYS, ctx=f"{label}: reset")
        rows = gpu.rollout_fused(T, None, tick=0, auto_reset=True)
        torch.cuda.synchronize()
        rows = to_np(rows)
        for t in range(T):
            c = cpu.step(cpu.random_actions(t))
            flags = {k: c[k].copy() for k in FLAG_KEYS}
            cpu.reset(mask=c["terminated"].copy())          # overwrites the observations of the slots that restarted
            got = {k: v[t] for k, v in rows.items()}
            assert_outputs_equal(got, cpu.out, keys=OBS_KEYS, ctx=f"{label}: tick {t}")
            assert_outputs_equal(got, flags, keys=FLAG_KEYS, ctx=f"{label}: tick {t}")
        assert_state_equal(to_np(gpu.get_state()), cpu.get_state(), ctx=f"{label}: state after {T} ticks")
        assert int(cpu.get_state()["reset_count"].min()) >= 2
        assert gpu.device_errors() == 0
        gpu.close()
    finally:
        cat_oracle.lib().cato_set_threads(1)
