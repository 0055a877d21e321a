"""Observation/action space descriptors.

The reference uses ``gymnasium.spaces`` (``Box``, ``Discrete``, ``Dict``; e.g.
``src/agents/entity.py:88-107``).  gymnasium is used here when it is importable; otherwise a
minimal stand-in with the attributes the env's callers touch (``shape``, ``dtype``, ``low``,
``high``, ``n``, ``start``, ``spaces``, ``sample()``, ``contains()``, item access on ``Dict``).
"""
from __future__ import annotations

from collections import OrderedDict

import numpy as np

try:  # pragma: no cover - depends on the installation
    from gymnasium.spaces import Box, Dict, Discrete  # type: ignore  # noqa: F401
    HAVE_GYMNASIUM = True
except ModuleNotFoundError:
    HAVE_GYMNASIUM = False

    class _Space:
        _rng = np.random.default_rng()

        def seed(self, seed=None):
            self._rng = np.random.default_rng(seed)
            return [seed]

        def __contains__(self, x):
            return self.contains(x)

    class Box(_Space):
        def __init__(self, low, high, shape=None, dtype=np.float32):
            self.dtype = np.dtype(dtype)
            self.shape = tuple(shape) if shape is not None else np.shape(low)
            self.low = np.full(self.shape, low, dtype=self.dtype)
            self.high = np.full(self.shape, high, dtype=self.dtype)

        def sample(self):
            if self.dtype.kind == "f":
                return self._rng.uniform(self.low, self.high).astype(self.dtype)
            return self._rng.integers(self.low, self.high, endpoint=True).astype(self.dtype)

        def contains(self, x):
            x = np.asarray(x)
            return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))

        def __repr__(self):
            return f"Box({self.low.min()}, {self.high.max()}, {self.shape}, {self.dtype})"

        def __eq__(self, o):
            return isinstance(o, Box) and self.shape == o.shape and self.dtype == o.dtype and \
                np.array_equal(self.low, o.low) and np.array_equal(self.high, o.high)

    class Discrete(_Space):
        def __init__(self, n, start=0):
            self.n, self.start = int(n), int(start)
            self.shape, self.dtype = (), np.dtype(np.int64)

        def sample(self):
            return int(self.start + self._rng.integers(self.n))

        def contains(self, x):
            try:
                xi = int(x)
            except (TypeError, ValueError):
                return False
            return self.start <= xi < self.start + self.n

        def __repr__(self):
            return f"Discrete({self.n})"

        def __eq__(self, o):
            return isinstance(o, Discrete) and (self.n, self.start) == (o.n, o.start)

    class Dict(_Space):
        def __init__(self, spaces=None, **kw):
            self.spaces = OrderedDict(spaces or {})
            self.spaces.update(kw)
            self.shape, self.dtype = None, None

        def __getitem__(self, k):
            return self.spaces[k]

        def __iter__(self):
            return iter(self.spaces)

        def __len__(self):
            return len(self.spaces)

        def keys(self):
            return self.spaces.keys()

        def items(self):
            return self.spaces.items()

        def values(self):
            return self.spaces.values()

        def sample(self):
            return {k: s.sample() for k, s in self.spaces.items()}

        def contains(self, x):
            return isinstance(x, dict) and x.keys() == self.spaces.keys() and \
                all(self.spaces[k].contains(v) for k, v in x.items())

        def __repr__(self):
            return "Dict(" + ", ".join(f"{k!r}: {v!r}" for k, v in self.spaces.items()) + ")"

        def __eq__(self, o):
            return isinstance(o, Dict) and self.spaces == o.spaces
