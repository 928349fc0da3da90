"""Observation/action space descriptors.  Uses `gym.spaces` when gym is installed (so reference
wrappers see real gym spaces) and an attribute-compatible minimal stand-in otherwise."""
import numpy as np

try:  # pragma: no cover - depends on the user's environment
    from gym import spaces as _gym_spaces
    Box, Tuple, Discrete = _gym_spaces.Box, _gym_spaces.Tuple, _gym_spaces.Discrete
    HAVE_GYM = True
except Exception:  # gym missing (as in the build image)
    HAVE_GYM = False

    class _Space:
        def __init__(self):
            self._rng = np.random.RandomState()

        def seed(self, seed=None):
            self._rng = np.random.RandomState(seed)
            return [seed]

        @property
        def np_random(self):
            return self._rng

        def __contains__(self, x):
            return self.contains(x)

    class Box(_Space):
        def __init__(self, low, high, shape=None, dtype=np.float64):
            super().__init__()
            self.dtype = np.dtype(dtype)
            if shape is None:
                shape = np.broadcast(np.asarray(low), np.asarray(high)).shape
            self.low = np.broadcast_to(np.asarray(low, dtype=self.dtype), shape).copy()
            self.high = np.broadcast_to(np.asarray(high, dtype=self.dtype), shape).copy()
            self.shape = tuple(shape)

        def sample(self):
            lo = np.where(np.isfinite(self.low), self.low, -1e6)
            hi = np.where(np.isfinite(self.high), self.high, 1e6)
            return self._rng.uniform(lo, hi).astype(self.dtype)

        def contains(self, x):
            x = np.asarray(x)
            return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))

        def __repr__(self):
            return f'Box({self.low.min()}, {self.high.max()}, {self.shape}, {self.dtype})'

    class Discrete(_Space):
        def __init__(self, n):
            super().__init__()
            self.n = int(n)
            self.shape = ()
            self.dtype = np.dtype(np.int64)

        def sample(self):
            return int(self._rng.randint(self.n))

        def contains(self, x):
            return 0 <= int(x) < self.n

    class Tuple(_Space):
        def __init__(self, spaces):
            super().__init__()
            self.spaces = tuple(spaces)

        def seed(self, seed=None):
            out = super().seed(seed)
            for i, space in enumerate(self.spaces):
                space.seed(None if seed is None else seed + i + 1)
            return out

        def sample(self):
            return tuple(space.sample() for space in self.spaces)

        def contains(self, x):
            return len(x) == len(self.spaces) and all(s.contains(v) for s, v in zip(self.spaces, x))

        def __getitem__(self, i):
            return self.spaces[i]

        def __len__(self):
            return len(self.spaces)

        def __iter__(self):
            return iter(self.spaces)
