"""Pins the oracle's float16 arithmetic against NumPy (the reference's own dependency).

Reference lines: src/agents/entity.py:200-210 (distance pipeline), src/agents/cop.py:69-74 and
src/agents/thief.py:63-69 (float16 reward scalars), src/environments/observation_spaces.py:92-95
(team positions cast to float16).  [NUMPY]-pinned, SURVEY.md quirks Q3/Q4.
"""
import numpy as np
import pytest

from as_cops_and_thieves_amd import tables


def test_f64_to_f16_matches_numpy_exhaustive_neighbourhoods(oracle_lib):
    rng = np.random.default_rng(0)
    # every f16 value, its midpoints to the next value, and tiny perturbations around them
    h = np.arange(0, 0x7C00, dtype=np.uint16).view(np.float16).astype(np.float64)
    mids = (h[:-1] + h[1:]) / 2
    xs = np.concatenate([h, mids, np.nextafter(mids, np.inf), np.nextafter(mids, -np.inf),
                         rng.uniform(-70000, 70000, 20000), rng.uniform(-1e-7, 1e-7, 5000),
                         [0.0, -0.0, 65504.0, 65519.99, 65520.0, 1e9, -1e9, np.inf, -np.inf, 2.0**-25, 2.0**-24]])
    xs = np.concatenate([xs, -xs])
    with np.errstate(over="ignore"):
        want = xs.astype(np.float16).view(np.uint16)
    got = np.array([oracle_lib.cato_f64_to_f16(float(x)) for x in xs], dtype=np.uint16)
    assert np.array_equal(got, want)


def test_f16_to_f64_roundtrip(oracle_lib):
    bits = np.arange(0, 0x7C01, dtype=np.uint16)
    want = bits.view(np.float16).astype(np.float64)
    got = np.array([oracle_lib.cato_f16_to_f64(int(b)) for b in bits])
    assert np.array_equal(got, want)


def _numpy_distance(points, origin):
    """entity.py:206-210 verbatim shape: f16 points, python-float origin, np.hypot."""
    valid_points = np.array(points, dtype=np.float16)
    out = []
    for p, (ox, oy) in zip(valid_points, origin):
        row = p.reshape(1, 2)
        dx = row[:, 0] - ox
        dy = row[:, 1] - oy
        out.append(np.hypot(dx, dy).astype(np.float16)[0])
    return np.array(out, dtype=np.float16)


def test_obs_distance_pipeline_matches_numpy(oracle_lib):
    rng = np.random.default_rng(1)
    n = 20000
    origin = rng.uniform(0, 1300, (n, 2))
    ang = rng.uniform(0, 2 * np.pi, n)
    rad = rng.uniform(0, 400, n)
    pts = origin + np.stack([rad * np.cos(ang), rad * np.sin(ang)], 1)
    want = _numpy_distance(pts, [tuple(map(float, o)) for o in origin]).view(np.uint16)
    got = np.array([oracle_lib.cato_obs_distance_f16(*map(float, (p[0], p[1], o[0], o[1])))
                    for p, o in zip(pts, origin)], dtype=np.uint16)
    assert np.array_equal(got, want)


def test_reward_luts_equal_scalar_numpy_expressions():
    """LUT entries == the reference's scalar expressions evaluated on np.float16 scalars."""
    cop, thief = tables.cop_reward_lut(), tables.thief_reward_lut()
    rng = np.random.default_rng(2)
    for bits in np.concatenate([rng.integers(0, 0x5F00, 500), [0, 1, 0x5E40, 0x3C00]]):
        d = np.uint16(bits).view(np.float16)
        reward = -0.02
        reward += 1.5 * np.exp(-d / 50.0)                     # cop.py:69-72
        assert isinstance(reward, np.float16)
        assert np.float32(reward) == cop[bits]
        t = np.tanh((d - 100.0) / 50.0) / 10.0                # thief.py:66
        assert isinstance(t, np.float16)
        assert np.float32(t) == thief[bits]
    assert tables.COP_NO_THIEF_REWARD == float(np.float32(-0.02 - 0.02))


def test_survey_q4_spot_values():
    # cop 0.691 vs f64 0.69121; thief -0.08496 vs -0.08493 (NumPy 2 float16 scalars)
    cop, thief = tables.cop_reward_lut(), tables.thief_reward_lut()
    vals = cop[(cop > 0.69) & (cop < 0.6925)]
    assert vals.size > 0
    assert np.isclose(thief[np.float16(50.0).view(np.uint16)], np.float32(np.float16(np.tanh(np.float16(-1.0)) / 10.0)))
