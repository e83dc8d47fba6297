"""Observation / action space descriptions.

If `gym` (or `gymnasium`) is importable its Box / Dict classes are used so SB3 wrappers see the real thing
(reference: base_tactile_env.py:76-114, edge_follow_env.py:169-174).  Neither is installed in the build image, so a
minimal duck-typed stand-in with the same attributes (shape, dtype, low, high, sample, contains, spaces) is provided.
"""
import numpy as np

try:  # pragma: no cover - depends on the host environment
    from gym import spaces as _sp
    Box, Dict = _sp.Box, _sp.Dict
    BACKEND = "gym"
except Exception:  # noqa: BLE001
    try:  # pragma: no cover
        from gymnasium import spaces as _sp
        Box, Dict = _sp.Box, _sp.Dict
        BACKEND = "gymnasium"
    except Exception:  # noqa: BLE001
        BACKEND = "builtin"

        class Box:
            def __init__(self, low, high, shape=None, dtype=np.float32):
                self.dtype = np.dtype(dtype)
                self.shape = tuple(shape) if shape is not None else np.shape(low)
                self.low = np.full(self.shape, low, dtype=self.dtype) if np.isscalar(low) else np.asarray(low, dtype=self.dtype)
                self.high = np.full(self.shape, high, dtype=self.dtype) if np.isscalar(high) else np.asarray(high, dtype=self.dtype)
                self._rng = np.random.default_rng()

            def seed(self, seed=None):
                self._rng = np.random.default_rng(seed)
                return [seed]

            def sample(self):
                if np.issubdtype(self.dtype, np.integer):
                    return self._rng.integers(self.low, self.high, size=self.shape, endpoint=True).astype(self.dtype)
                lo = np.where(np.isfinite(self.low), self.low, -1.0)
                hi = np.where(np.isfinite(self.high), self.high, 1.0)
                return self._rng.uniform(lo, hi, size=self.shape).astype(self.dtype)

            def contains(self, x):
                x = np.asarray(x)
                return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))

            def __repr__(self):
                return f"Box({self.low.min()}, {self.high.max()}, {self.shape}, {self.dtype})"

        class Dict:
            def __init__(self, spaces):
                self.spaces = dict(spaces)

            def __getitem__(self, k):
                return self.spaces[k]

            def keys(self):
                return self.spaces.keys()

            def items(self):
                return self.spaces.items()

            def sample(self):
                return {k: s.sample() for k, s in self.spaces.items()}

            def contains(self, x):
                return all(k in x and s.contains(x[k]) for k, s in self.spaces.items())

            def __repr__(self):
                return f"Dict({self.spaces})"
