"""CPU: the oracle against Pymunk itself -- the only test that can move parity from "unpinned" to pinned.

SKIPS in this image: Pymunk (-> Chipmunk2D) is a dependency of the reference that is not installed and cannot be installed
here (no network).  On a box that has it, a scene is built through the very Pymunk calls the reference makes
(oracle/pymunk_double.py: own code, cited line by line), spawn positions are injected on both sides, a fixed action tape
is played, and the oracle must agree: capture / timeout flags and winner exactly, ray classes exactly, f16 distances within
one float16 step, rewards to 2e-3, positions and velocities to 1e-5 (north_star) -- while the env has had at most one
contact at a time; with several simultaneous contacts Chipmunk's arbiter ORDER (an artefact of its BBTree, SURVEY Q15, DESIGN D2)
differs from the oracle's fixed order, so an env is compared up to its first multi-contact tick."""
import json

import numpy as np
import pytest

pymunk = pytest.importorskip("pymunk", reason="Pymunk / Chipmunk2D is not installed in this image: oracle parity stays UNPINNED")

from oracle import cat_oracle_host as host                      # noqa: E402
from oracle.cat_oracle import OracleSim                         # noqa: E402
from oracle.pymunk_double import PymunkScene                    # noqa: E402

MAPS = {   # name -> (roster override, start positions, scale): the build-side presets of SURVEY 0.2 as plain data
    "squarinth": (None, None, None),
    "lbirinth": (None, None, None),
    "grandbyrinth": (None, None, None),
    "agh-map": (None, None, None),
    "labyrinth": (["cop", "cop", "thief"], [(64.0, 54.0), (64.0, 198.0), (1130.0, 342.0)], (1280.0 / 30.0, 720.0 / 20.0)),
}


class _Cfg:
    def __init__(self, **kw):
        self.__dict__.update(kw)


def _free_positions(scene: PymunkScene, rng, window, n):
    out = []
    for i in range(n):
        for _ in range(10000):
            p = (float(rng.uniform(20, window[0] - 20)), float(rng.uniform(20, window[1] - 20)))
            if scene.spawn_is_free(i, p) and all(np.hypot(p[0] - q[0], p[1] - q[1]) > 14 for q in out):
                out.append(p)
                break
        else:
            raise RuntimeError("no free spawn position")
    return out


@pytest.mark.parametrize("name", sorted(MAPS))
def test_oracle_agrees_with_pymunk(name):
    roster, starts, scale = MAPS[name]
    data = host.raw_map(name)
    R, T, EPISODES = 90, 200, 6
    rng = np.random.default_rng(abs(hash(name)) % (1 << 31))
    compared = 0
    for ep in range(EPISODES):
        scene = PymunkScene(data, n_rays=R, max_step_count=150, roster=roster, start_positions=starts, scale=scale)
        blob = host.compile_blob(data, roster, starts, None, scale)
        cfg = _Cfg(n_envs=1, n_cops=scene.n_cops, n_thieves=scene.n_thieves, n_rays=R, max_step_count=150, seed=ep + 1)
        cpu = OracleSim(cfg, [blob])
        window = tuple(data["window"].values())
        pos = _free_positions(scene, rng, window, scene.A)
        if ep % 2:   # half of the episodes: thieves near cop 0, so that agent contacts and captures occur
            pos[scene.n_cops:] = [(pos[0][0] + 30.0 + 12 * k, pos[0][1] + 7.0) for k in range(scene.n_thieves)]
        scene.reset(pos)
        cpu.reset(positions=np.asarray(pos, np.float64).reshape(1, scene.A, 2))
        for t in range(T):
            a = cpu.random_actions(t)
            want = scene.step(a[0])
            got = cpu.step(a)
            st = cpu.get_state()
            ctx = f"{name} episode {ep} tick {t}"
            assert (int(got["terminated"][0]), int(got["truncated"][0]), int(got["winner"][0])) == \
                   (want["terminated"], want["truncated"], want["winner"]), ctx
            assert np.array_equal(got["obs_type"][0], want["obs_type"]), ctx
            d_got = got["obs_distance"][0].view(np.float16).astype(np.float64)
            d_want = want["obs_distance"].view(np.float16).astype(np.float64)
            assert np.all(np.abs(d_got - d_want) <= np.spacing(np.maximum(d_got, d_want).astype(np.float16)).astype(np.float64)), ctx
            assert np.allclose(got["reward"][0], want["reward"], atol=2e-3), ctx
            assert np.allclose(st["pos"][0], want["pos"], atol=1e-5, rtol=0) and np.allclose(st["vel"][0], want["vel"], atol=1e-5, rtol=0), ctx
            compared += 1
            contacts = int((st["wall_shape"][0] >= 0).sum() + (st["pair_age"][0] >= 0).sum())
            if contacts > 1 or want["terminated"]:
                break
    assert compared >= 100, f"{name}: only {compared} ticks could be compared"


@pytest.mark.parametrize("name", sorted(MAPS))
def test_winning_shapes_follow_the_restated_bbtree_descent(name):
    """DESIGN D2 against the engine: WHICH shape a ray's ``segment_query_first`` returns depends on the order of Chipmunk's visits.  The oracle's
    diagnostic order 2 restates that order for the walls (cpBBTreeInsert in file order, nearer child first): with it the winning shape of every ray
    must be Pymunk's own; with the product's index order the rays that differ are counted and reported (``tools/query_order_diff.py`` predicts
    0.0001 % on the labyrinth ... 0.5 % on lbirinth).  Static scenes only: the agents stand where they were put, nothing is stepped."""
    from oracle import cat_oracle
    roster, starts, scale = MAPS[name]
    data = host.raw_map(name)
    R = 90
    rng = np.random.default_rng(abs(hash(name)) % (1 << 31) + 7)
    L = cat_oracle.lib()
    rays = differ_index = 0
    try:
        for ep in range(40):
            scene = PymunkScene(data, n_rays=R, max_step_count=150, roster=roster, start_positions=starts, scale=scale)
            blob = host.compile_blob(data, roster, starts, None, scale)
            cfg = _Cfg(n_envs=1, n_cops=scene.n_cops, n_thieves=scene.n_thieves, n_rays=R, max_step_count=150, seed=ep + 1)
            window = tuple(data["window"].values())
            pos = _free_positions(scene, rng, window, scene.A)
            scene.reset(pos)
            for i in range(scene.A):
                scene.observe(i)
            want = np.stack([scene.last_hit_shape[i] for i in range(scene.A)])
            got = {}
            for mode in (2, 1):
                cpu = OracleSim(cfg, [blob])
                L.cato_set_index_order(mode)
                out = cpu.reset(positions=np.asarray(pos, np.float64).reshape(1, scene.A, 2))
                got[mode] = out["hit_shape"][0].reshape(scene.A, R)
            assert np.array_equal(got[2], want), f"{name} scene {ep}: the restated tree descent returns another shape than Pymunk on {int((got[2] != want).sum())} rays"
            rays += want.size
            differ_index += int((got[1] != want).sum())
    finally:
        L.cato_set_index_order(1)
    print(f"{name}: index order (the product's) returns another shape than Pymunk on {differ_index} of {rays} rays")
