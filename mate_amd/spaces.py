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


def _square_grid(levels):
    """`levels` x `levels` grid on [-1, 1]^2, x fastest (np.meshgrid of two linspaces, discrete_action_spaces.py:107-113)."""
    if not (isinstance(levels, (int, np.integer)) and levels >= 3 and levels % 2 == 1):
        raise AssertionError(f'The discrete level must be an odd number that not less than 3. Got levels = {levels}.')
    axis = np.linspace(start=-1.0, stop=+1.0, num=levels, endpoint=True)
    return np.stack(np.meshgrid(axis, axis), axis=-1).reshape(-1, 2)


def camera_action_grid(levels):
    """Normalised action grid of the reference's DiscreteCamera wrapper (discrete_action_spaces.py:98-117):
    action = action_space.high * grid[index]."""
    return _square_grid(levels)


def target_action_grid(levels):
    """Normalised action grid of DiscreteTarget (discrete_action_spaces.py:204-228): the square grid pulled
    onto the unit disc along each ray, so that every direction has the same maximum step."""
    grid = _square_grid(levels)
    angle = np.arctan2(grid[..., -1], grid[..., 0])
    bound = 1.0 / np.cos(np.pi * ((angle / np.pi + 0.25) % 0.5 - 0.25))
    return grid / bound[..., np.newaxis]
