import numpy as np

from gym.utils import seeding


class Space:
    def __init__(self, shape=None, dtype=None):
        self._shape = None if shape is None else tuple(shape)
        self.dtype = None if dtype is None else np.dtype(dtype)
        self._np_random = None

    @property
    def np_random(self):
        if self._np_random is None:
            self.seed()
        return self._np_random

    @property
    def shape(self):
        return self._shape

    def seed(self, seed=None):
        self._np_random, seed = seeding.np_random(seed)
        return [seed]

    def sample(self):
        raise NotImplementedError

    def contains(self, x):
        raise NotImplementedError

    def __contains__(self, x):
        return self.contains(x)


class Box(Space):
    def __init__(self, low, high, shape=None, dtype=np.float32):
        dtype = np.dtype(dtype)
        if shape is None:
            shape = np.broadcast(np.asarray(low), np.asarray(high)).shape
        low = np.broadcast_to(np.asarray(low, dtype=np.float64), shape).astype(dtype)
        high = np.broadcast_to(np.asarray(high, dtype=np.float64), shape).astype(dtype)
        super().__init__(shape, dtype)
        self.low = low
        self.high = high
        self.bounded_below = -np.inf < self.low
        self.bounded_above = np.inf > self.high

    def is_bounded(self, manner='both'):
        below, above = np.all(self.bounded_below), np.all(self.bounded_above)
        return {'both': below and above, 'below': below, 'above': above}[manner]

    def sample(self):
        high = self.high if self.dtype.kind == 'f' else self.high.astype('int64') + 1
        sample = np.empty(self.shape)
        unbounded = ~self.bounded_below & ~self.bounded_above
        upp_bounded = ~self.bounded_below & self.bounded_above
        low_bounded = self.bounded_below & ~self.bounded_above
        bounded = self.bounded_below & self.bounded_above
        sample[unbounded] = self.np_random.normal(size=unbounded[unbounded].shape)
        sample[low_bounded] = (
            self.np_random.exponential(size=low_bounded[low_bounded].shape) + self.low[low_bounded]
        )
        sample[upp_bounded] = (
            -self.np_random.exponential(size=upp_bounded[upp_bounded].shape) + self.high[upp_bounded]
        )
        sample[bounded] = self.np_random.uniform(
            low=self.low[bounded], high=high[bounded], size=bounded[bounded].shape
        )
        if self.dtype.kind == 'i':
            sample = np.floor(sample)
        return sample.astype(self.dtype)

    def contains(self, x):
        x = np.asarray(x)
        return bool(
            np.can_cast(x.dtype, self.dtype)
            and x.shape == self.shape
            and np.all(x >= self.low)
            and np.all(x <= self.high)
        )

    def __repr__(self):
        return f'Box({self.low.min()}, {self.high.max()}, {self.shape}, {self.dtype})'

    def __eq__(self, other):
        return (
            isinstance(other, Box)
            and self.shape == other.shape
            and np.allclose(self.low, other.low)
            and np.allclose(self.high, other.high)
        )

    __hash__ = None


class Discrete(Space):
    def __init__(self, n):
        assert n >= 0
        self.n = int(n)
        super().__init__((), np.int64)

    def sample(self):
        return int(self.np_random.randint(self.n))

    def contains(self, x):
        try:
            return 0 <= int(x) < self.n and int(x) == x
        except (TypeError, ValueError):
            return False

    def __repr__(self):
        return f'Discrete({self.n})'

    def __eq__(self, other):
        return isinstance(other, Discrete) and self.n == other.n

    __hash__ = None


class MultiDiscrete(Space):
    def __init__(self, nvec, dtype=np.int64):
        self.nvec = np.asarray(nvec, dtype=dtype)
        super().__init__(self.nvec.shape, dtype)

    def sample(self):
        return (self.np_random.random_sample(self.nvec.shape) * self.nvec).astype(self.dtype)

    def contains(self, x):
        x = np.asarray(x)
        return x.shape == self.shape and bool(np.all(0 <= x) and np.all(x < self.nvec))


class Tuple(Space):
    def __init__(self, spaces):
        self.spaces = tuple(spaces)
        super().__init__(None, None)

    def seed(self, seed=None):
        seeds = super().seed(seed)
        for space in self.spaces:
            seeds.extend(space.seed(int(self.np_random.randint(2**31 - 1))))
        return seeds

    def sample(self):
        return tuple(space.sample() for space in self.spaces)

    def contains(self, x):
        return (
            isinstance(x, (tuple, list))
            and len(x) == len(self.spaces)
            and all(space.contains(part) for space, part in zip(self.spaces, x))
        )

    def __getitem__(self, index):
        return self.spaces[index]

    def __len__(self):
        return len(self.spaces)

    def __iter__(self):
        return iter(self.spaces)


class Dict(Space):
    def __init__(self, spaces=None, **kwargs):
        spaces = dict(spaces or {})
        spaces.update(kwargs)
        self.spaces = spaces
        super().__init__(None, None)

    def seed(self, seed=None):
        seeds = super().seed(seed)
        for space in self.spaces.values():
            seeds.extend(space.seed(int(self.np_random.randint(2**31 - 1))))
        return seeds

    def sample(self):
        return {k: space.sample() for k, space in self.spaces.items()}

    def contains(self, x):
        return isinstance(x, dict) and all(k in x and s.contains(x[k]) for k, s in self.spaces.items())

    def __getitem__(self, key):
        return self.spaces[key]

    def keys(self):
        return self.spaces.keys()

    def items(self):
        return self.spaces.items()
