"""Minimal stand-ins for the gym spaces the reference's env classes expose
(``gym`` 0.23 is a dependency of the reference, setup.py:13-20, and is not
installed here).  Only what callers of the env API read is provided:
``Discrete.n``, ``MultiBinary.n/.shape``, ``Box.low/.high/.shape/.dtype`` and
``sample()``.  If gym is importable its classes are used instead."""
import numpy as np

try:  # pragma: no cover - not installed in the build image
    from gym.spaces import Box, Discrete, MultiBinary, MultiDiscrete  # noqa: F401
except Exception:  # noqa: BLE001

    class Space:
        shape = ()
        dtype = None

    class Discrete(Space):
        def __init__(self, n):
            self.n = int(n)
            self.shape = ()
            self.dtype = np.int64

        def sample(self):
            return int(np.random.randint(self.n))

        def contains(self, x):
            return 0 <= int(x) < self.n

        def __repr__(self):
            return f"Discrete({self.n})"

    class MultiBinary(Space):
        def __init__(self, n):
            self.n = n
            self.shape = tuple(int(v) for v in np.atleast_1d(np.asarray(n)))
            self.dtype = np.int8

        def sample(self):
            return np.random.randint(0, 2, size=self.shape, dtype=np.int8)

        def __repr__(self):
            return f"MultiBinary({self.n})"

    class MultiDiscrete(Space):
        def __init__(self, nvec):
            self.nvec = np.asarray(nvec, dtype=np.int64)
            self.shape = self.nvec.shape
            self.dtype = np.int64

        def sample(self):
            return (np.random.random(self.shape) * self.nvec).astype(np.int64)

        def __repr__(self):
            return f"MultiDiscrete({self.nvec.tolist()})"

    class Box(Space):
        def __init__(self, low, high, shape=None, dtype=np.float32):
            self.dtype = np.dtype(dtype)
            if shape is None:
                shape = np.asarray(low).shape
            self.shape = tuple(shape)
            self.low = np.broadcast_to(np.asarray(low, dtype=self.dtype), self.shape)
            self.high = np.broadcast_to(np.asarray(high, dtype=self.dtype), self.shape)

        def sample(self):
            return np.random.uniform(np.maximum(self.low, -1e6), np.minimum(self.high, 1e6)).astype(self.dtype)

        def __repr__(self):
            return f"Box{self.shape}"
