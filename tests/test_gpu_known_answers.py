"""GPU: the analytic known-answer cases of tests/test_oracle_known_answers.py run through the C ABI of the HIP
library (cat_create / cat_reset / cat_set_state / cat_step / cat_get_state) instead of the oracle: the product
itself is shown cases whose answers follow from the specification, not from a comparison with the oracle."""
import numpy as np
import pytest

import tests.test_oracle_known_answers as ka
from tests.util import to_np

pytestmark = pytest.mark.gpu


class GpuSim:
    """The OracleSim surface the cases use, on top of CatSim (ctypes -> libcat_sim.so)."""

    def __init__(self, cfg, cmaps):
        import torch
        from as_cops_and_thieves_amd.sim import CatSim
        self.torch, self.cfg = torch, cfg
        self.sim = CatSim(cfg, cmaps, device="cuda:0", debug_hit_shape=True)

    def _out(self):
        self.torch.cuda.synchronize()
        return to_np(self.sim.out)

    def reset(self, mask=None, positions=None):
        t = self.torch
        self.sim.reset(mask=None if mask is None else t.as_tensor(np.ascontiguousarray(mask, np.uint8)),
                       positions=None if positions is None else t.as_tensor(np.ascontiguousarray(positions, np.float64)))
        return self._out()

    def step(self, actions):
        self.sim.step(self.torch.as_tensor(np.ascontiguousarray(actions, np.int32)))
        return self._out()

    def random_actions(self, tick):
        return self.sim.random_actions(tick).cpu().numpy()

    def get_state(self):
        return to_np(self.sim.get_state())

    def set_state(self, **arrays):
        self.sim.set_state(**arrays)


@pytest.fixture(autouse=True)
def _hip_backend(monkeypatch):
    monkeypatch.setattr(ka, "make_sim", lambda cfg, cmaps: GpuSim(cfg, cmaps))


# the cases themselves (collected here under the gpu marker, with the HIP backend patched in)
from tests.test_oracle_known_answers import (  # noqa: E402,F401
    test_agent_pressed_into_wall_settles_within_slop, test_batch_independence_and_determinism,
    test_capture_needs_wall_line_of_sight, test_capture_radius_is_strict, test_free_flight_impulse_and_speed_clamp,
    test_ray_hits_other_agent_and_classifies_by_category, test_ray_hits_rounded_wall_at_computed_distance,
    test_reset_keeps_stale_shape_caches, test_rewards_follow_reference_formulas,
    test_shared_observations_first_nonempty_member_wins, test_single_wall_map_is_gated_like_any_other_wall,
    test_spawn_sampling_respects_regions_and_falls_back_to_centre, test_termination_is_one_tick_late_and_timeout_semantics,
    test_two_agents_collide_inelastically, test_vertex_region_behind_an_adjacent_edge_is_not_a_contact,
    test_ray_hits_a_rounded_corner_at_the_computed_distance, test_ray_grazing_a_corner_and_the_bbtree_gate,
    test_origin_within_the_ray_radius_of_a_wall_reports_the_segment_end, test_mirrored_scenes_evolve_as_mirror_images,
    test_circle_against_a_hull_corner_loses_exactly_its_normal_velocity)


def test_philox_known_answers_through_the_device_action_stream():
    """Random123 Philox4x32-10 known answers are checked on the oracle's raw function (CPU); on the device the same
    generator is only reachable through its users, so: the device's synthetic action stream equals the oracle's, which
    the CPU test pins to the Random123 vectors."""
    from as_cops_and_thieves_amd.config import SimConfig
    from as_cops_and_thieves_amd.maps import load_preset
    from oracle.cat_oracle import OracleSim
    cmap = load_preset("squarinth").compile()
    cfg = SimConfig(n_envs=33, n_rays=8, seed=0xDEADBEEFCAFE, env_id_offset=(1 << 33) + 5)
    g, c = GpuSim(cfg, [cmap]), OracleSim(cfg, [cmap])
    for tick in (0, 1, 2**31 + 7):
        assert np.array_equal(g.random_actions(tick), c.random_actions(tick))
