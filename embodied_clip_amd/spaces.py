"""Minimal stand-ins for gym.spaces (gym is not installed in this image).
If gym IS importable, the real classes are used so experiment configs that
type-check against gym keep working (AllenAct's Preprocessor.observation_space
is a ``gym.spaces.Box`` -- SURVEY.md §8b)."""
from __future__ import annotations

import numpy as np

try:  # pragma: no cover - gym is absent here
    from gym.spaces import Box, Discrete, Dict  # type: ignore
except Exception:  # noqa: BLE001
    class Box:  # type: ignore
        def __init__(self, low, high, shape, dtype=np.float32):
            self.low, self.high, self.shape, self.dtype = low, high, tuple(shape), dtype

        def __repr__(self):
            return f"Box({self.low}, {self.high}, {self.shape}, {self.dtype})"

    class Discrete:  # type: ignore
        def __init__(self, n):
            self.n = int(n)

        def __repr__(self):
            return f"Discrete({self.n})"

    class Dict:  # type: ignore
        def __init__(self, spaces):
            self.spaces = dict(spaces)

        def __getitem__(self, k):
            return self.spaces[k]

        def __contains__(self, k):
            return k in self.spaces
