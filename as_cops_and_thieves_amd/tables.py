"""Host-built lookup tables handed to the device library at construction.

* Ray direction table: built with NumPy exactly as the reference builds its ray endpoints
  (``src/agents/entity.py:182-193``): ``angles = np.linspace(0, fov, R, endpoint=False)``,
  offsets ``ray_length * cos/sin`` in float64.  The device adds the agent origin, so ray end
  points are bit-identical to the reference's NumPy arithmetic.
* Reward tables: the reference's non-terminal rewards are NumPy *float16* scalar expressions
  of the minimum observed distance (``src/agents/cop.py:69-74``, ``src/agents/thief.py:63-66``;
  SURVEY.md quirk Q4).  A float16 distance has at most 32768 non-negative bit patterns, so the
  expressions are evaluated once per pattern with NumPy itself and the kernel indexes the table
  with the f16 bits — bit-exact to NumPy by construction, no device transcendental involved.
"""
from __future__ import annotations

import numpy as np

from .constants import DEFAULT_SENSOR, SensorParams


def ray_table(sensor: SensorParams = DEFAULT_SENSOR) -> tuple[np.ndarray, np.ndarray]:
    angles = np.linspace(0, sensor.fov, sensor.num_rays, endpoint=False)
    return (np.ascontiguousarray(sensor.ray_length * np.cos(angles), dtype=np.float64),
            np.ascontiguousarray(sensor.ray_length * np.sin(angles), dtype=np.float64))


def _all_nonneg_f16() -> np.ndarray:
    return np.arange(32768, dtype=np.uint16).view(np.float16)


def _h(x: np.ndarray) -> np.ndarray:
    """One float16 operation's result: the exact (float64) value rounded once to float16, widened again."""
    return x.astype(np.float16).astype(np.float64)


def cop_reward_lut() -> np.ndarray:
    """``reward = -0.02; reward += 1.5 * np.exp(-d / 50.0)`` with ``d`` a float16 SCALAR (cop.py:63-72), widened to f32.

    NumPy evaluates every operation of a float16 scalar expression by widening, operating and rounding back to float16 (Python
    floats are "weak": they become float16 first).  The table restates exactly that with float64 intermediates -- one rounding
    to float16 per operation -- so it does not depend on the host's SIMD paths: NumPy's vectorised float16 ``exp`` (AVX512
    float32 kernel, then a second rounding) is off by one float16 ulp for 2 of the 32768 distances on this image's CPUs, while
    the scalar call the reference makes is not (tests/test_oracle_independent_host.py compares with NumPy's scalar results)."""
    d = _all_nonneg_f16().astype(np.float64)
    with np.errstate(all="ignore"):
        t = _h(-d / 50.0)
        e = _h(np.exp(t))
        m = _h(1.5 * e)
        reward = _h(float(np.float16(-0.02)) + m)
    return np.ascontiguousarray(reward.astype(np.float32))


def thief_reward_lut() -> np.ndarray:
    """``np.tanh((d - 100.0) / 50.0) / 10.0`` with ``d`` a float16 scalar (thief.py:65-66); see ``cop_reward_lut``."""
    d = _all_nonneg_f16().astype(np.float64)
    with np.errstate(all="ignore"):
        u = _h(d - 100.0)
        v = _h(u / 50.0)
        w = _h(np.tanh(v))
        reward = _h(w / 10.0)
    return np.ascontiguousarray(reward.astype(np.float32))


COP_NO_THIEF_REWARD = float(np.float32(-0.02 - 0.02))   # cop.py:63,74
THIEF_NO_COP_REWARD = float(np.float32(0.15))           # thief.py:69
