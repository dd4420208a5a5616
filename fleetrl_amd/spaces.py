"""`Box` space: gymnasium's when gymnasium is installed, otherwise a minimal stand-in with the same attributes
(the reference only needs `gym.spaces.Box(low, high, shape=None, dtype)`,
/root/reference/fleetrl/fleet_env/fleet_environment.py:316-325)."""
from __future__ import annotations

import numpy as np

try:  # pragma: no cover - gymnasium is not installed in the build image
    from gymnasium.spaces import Box  # type: ignore
except Exception:

    class Box:  # noqa: D401 - mirrors gymnasium.spaces.Box
        def __init__(self, low, high, shape=None, dtype=np.float32, seed=None):
            self.dtype = np.dtype(dtype)
            if shape is None:
                shape = np.shape(low)
            self.shape = tuple(int(s) for s in shape)
            self.low = np.broadcast_to(np.asarray(low, dtype=self.dtype), self.shape).copy()
            self.high = np.broadcast_to(np.asarray(high, dtype=self.dtype), self.shape).copy()
            self._rng = np.random.default_rng(seed)

        def seed(self, seed=None):
            self._rng = np.random.default_rng(seed)
            return [seed]

        def sample(self):
            lo = np.where(np.isfinite(self.low), self.low, -1.0)
            hi = np.where(np.isfinite(self.high), self.high, 1.0)
            return self._rng.uniform(lo, hi).astype(self.dtype)

        def contains(self, x) -> bool:
            x = np.asarray(x)
            return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))

        def __repr__(self):
            return f"Box({self.low.min()}, {self.high.max()}, {self.shape}, {self.dtype})"


def observation_bounds(dim: int, normalized: bool):
    """`Normalization.make_boundaries`: [0,1] for OracleNormalization (oracle_normalization.py:164-173), +-inf for
    UnitNormalization (unit_normalization.py:23-31; float64 arrays there, cast to float32 by the Box)."""
    if normalized:
        return np.zeros(dim, dtype=np.float32), np.ones(dim, dtype=np.float32)
    return np.full(dim, -np.inf), np.full(dim, np.inf)
