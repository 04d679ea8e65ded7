"""Minimal Box space (gymnasium is not installed here or on the GPU box).

Only what the reference's `EnvWrapper` surface uses from `gymnasium.spaces.Box`
(/root/reference/src/mobrob/envs/wrapper.py:31-34,95-107,250-264): low/high/shape/dtype,
seed(), sample(), contains()."""
from __future__ import annotations

import numpy as np


class Box:
    def __init__(self, low, high, shape=None, dtype=np.float32, seed=None):
        self.dtype = np.dtype(dtype)
        if shape is None:
            shape = np.broadcast(np.asarray(low), np.asarray(high)).shape
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
        bounded = np.isfinite(self.low) & np.isfinite(self.high)
        u = self._rng.uniform(lo, hi)
        g = self._rng.standard_normal(self.shape)
        return np.where(bounded, u, g).astype(self.dtype)

    def contains(self, x):
        x = np.asarray(x)
        return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))

    def __repr__(self):
        return f"Box({self.low.min()}, {self.high.max()}, {self.shape}, {self.dtype})"
