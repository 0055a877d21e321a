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
