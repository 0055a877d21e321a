"""Headless ``rgb_array`` frame.

The reference's ``rgb_array`` mode returns ``pygame.surfarray.array3d`` of a surface it never
draws on (``base_env.py:505-507``): an all-zero ``(width, height, 3)`` uint8 array.  This
rasteriser keeps that shape/dtype convention but draws the scene (hull walls, cops blue, thieves
red — the colours of ``cop.py:30`` / ``thief.py:29``), which is what a user wants from a frame.
"""
from __future__ import annotations

import numpy as np

from .maps import CompiledMap


def render_rgb_array(cmap: CompiledMap, positions: np.ndarray, n_cops: int, agent_radius: float) -> np.ndarray:
    W, H = int(cmap.window[0]), int(cmap.window[1])
    img = np.full((W, H, 3), 255, dtype=np.uint8)
    xs = np.arange(W, dtype=np.float64)[:, None] + 0.5
    ys = np.arange(H, dtype=np.float64)[None, :] + 0.5
    for s in range(cmap.n_shapes):
        f, c = int(cmap.shape_first[s]), int(cmap.shape_count[s])
        l, b, r, t = cmap.shape_bb[s]
        x0, x1 = max(int(l), 0), min(int(r) + 1, W)
        y0, y1 = max(int(b), 0), min(int(t) + 1, H)
        if x0 >= x1 or y0 >= y1:
            continue
        inside = np.ones((x1 - x0, y1 - y0), dtype=bool)
        for pl in cmap.planes[f:f + c]:
            inside &= (pl[0] * xs[x0:x1] + pl[1] * ys[:, y0:y1] - pl[4]) <= 1.0   # hull inflated by r = 1
        img[x0:x1, y0:y1][inside] = (60, 60, 60)
    for i, (px, py) in enumerate(np.asarray(positions, dtype=np.float64)):
        colour = (0, 0, 255) if i < n_cops else (255, 0, 0)
        x0, x1 = max(int(px - agent_radius) - 1, 0), min(int(px + agent_radius) + 2, W)
        y0, y1 = max(int(py - agent_radius) - 1, 0), min(int(py + agent_radius) + 2, H)
        if x0 >= x1 or y0 >= y1:
            continue
        disc = (xs[x0:x1] - px) ** 2 + (ys[:, y0:y1] - py) ** 2 <= agent_radius ** 2
        img[x0:x1, y0:y1][disc] = colour
    return img
