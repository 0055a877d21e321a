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

import warnings

import numpy as np

from .constants import DEFAULT_SENSOR, SensorParams


def ray_table(sensor: SensorParams = DEFAULT_SENSOR) -> tuple[np.ndarray, np.ndarray]:
    angles = np.linspace(0, sensor.fov, sensor.num_rays, endpoint=False)
    return (np.ascontiguousarray(sensor.ray_length * np.cos(angles), dtype=np.float64),
            np.ascontiguousarray(sensor.ray_length * np.sin(angles), dtype=np.float64))


def _all_nonneg_f16() -> np.ndarray:
    return np.arange(32768, dtype=np.uint16).view(np.float16)


def cop_reward_lut() -> np.ndarray:
    """``-0.02 + 1.5 * np.exp(-d / 50.0)`` with ``d`` float16 (cop.py:69-72), widened to f32."""
    d = _all_nonneg_f16()
    with warnings.catch_warnings(), np.errstate(all="ignore"):
        warnings.simplefilter("ignore")
        reward = -0.02
        reward = reward + 1.5 * np.exp(-d / 50.0)
    assert reward.dtype == np.float16
    return np.ascontiguousarray(reward.astype(np.float32))


def thief_reward_lut() -> np.ndarray:
    """``np.tanh((d - 100.0) / 50.0) / 10.0`` with ``d`` float16 (thief.py:65-66)."""
    d = _all_nonneg_f16()
    with warnings.catch_warnings(), np.errstate(all="ignore"):
        warnings.simplefilter("ignore")
        reward = np.tanh((d - 100.0) / 50.0) / 10.0
    assert reward.dtype == np.float16
    return np.ascontiguousarray(reward.astype(np.float32))


COP_NO_THIEF_REWARD = float(np.float32(-0.02 - 0.02))   # cop.py:63,74
THIEF_NO_COP_REWARD = float(np.float32(0.15))           # thief.py:69
