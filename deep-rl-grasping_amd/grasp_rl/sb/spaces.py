"""Minimal Box / Discrete spaces (gym is not a dependency of the update path).  Any object with the
same attributes -- in particular real ``gym.spaces`` instances created by the reference's env
(/root/reference/manipulation_main/gripperEnv/robot.py:207-228, actuator.py:54-89) -- is accepted
wherever these are."""
import numpy as np


class Box:
    def __init__(self, low, high, shape=None, dtype=np.float32):
        if shape is None:
            low, high = np.asarray(low), np.asarray(high)
            shape = low.shape
        self.shape = tuple(shape)
        self.dtype = np.dtype(dtype)
        self.low = np.broadcast_to(np.asarray(low, dtype=self.dtype), self.shape).copy()
        self.high = np.broadcast_to(np.asarray(high, dtype=self.dtype), self.shape).copy()
        self._rng = np.random.default_rng()

    def seed(self, seed=None):
        self._rng = np.random.default_rng(seed)

    def sample(self):
        return self._rng.uniform(self.low, self.high).astype(self.dtype)

    def contains(self, x):
        x = np.asarray(x)
        return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))

    def __setstate__(self, state):          # also lets pickled gym.spaces.Box land here
        self.__dict__.update(state)
        self.shape = tuple(self.__dict__.get("shape") or self.__dict__.get("_shape") or np.shape(self.low))
        self.dtype = np.dtype(self.__dict__.get("dtype", np.float32))
        self._rng = np.random.default_rng()

    def __getstate__(self):
        d = dict(self.__dict__)
        d.pop("_rng", None)
        return d

    def __repr__(self):
        return "Box(%s, %s, %s)" % (np.min(self.low), np.max(self.high), self.shape)

    def __eq__(self, other):
        return isinstance(other, Box) and self.shape == other.shape and np.allclose(self.low, other.low) \
            and np.allclose(self.high, other.high)


class Discrete:
    def __init__(self, n):
        self.n = int(n)
        self.shape = ()
        self.dtype = np.dtype(np.int64)
        self._rng = np.random.default_rng()

    def seed(self, seed=None):
        self._rng = np.random.default_rng(seed)

    def sample(self):
        return int(self._rng.integers(self.n))

    def contains(self, x):
        return 0 <= int(x) < self.n

    def __setstate__(self, state):
        self.__dict__.update(state)
        self.shape = ()
        self._rng = np.random.default_rng()

    def __getstate__(self):
        d = dict(self.__dict__)
        d.pop("_rng", None)
        return d

    def __repr__(self):
        return "Discrete(%d)" % self.n


def is_box(space):
    return hasattr(space, "low") and hasattr(space, "high") and hasattr(space, "shape")


def is_discrete(space):
    return hasattr(space, "n") and not hasattr(space, "low")


def has_finite_bounds(space):
    """SB ``observation_input``: Box observations are min-max scaled only when the bounds are finite
    and not degenerate."""
    low, high = np.asarray(space.low, np.float64), np.asarray(space.high, np.float64)
    return bool(np.all(np.isfinite(low)) and np.all(np.isfinite(high)) and np.any(high - low != 0))
