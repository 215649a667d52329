"""gym.spaces for the adapters: the real `gym` (0.21, as pinned by the reference's requirements.yml)
when it is importable, otherwise minimal stand-ins with the attributes the reference's callers read
(`.low/.high/.shape/.dtype/.n/.spaces/.keys()/[]/.sample()`)."""
import numpy as np

try:  # pragma: no cover - gym is absent in the build image
    from gym.spaces import Box, Dict, Discrete, MultiDiscrete  # noqa: F401
except Exception:

    class _Space:
        """like gym 0.21's Space, sampling draws from a PRIVATE generator (np_random, seedable with .seed()): a
        .sample() must not advance the process-global np.random stream the env itself consumes"""

        def __init__(self, shape=None, dtype=None):
            self.shape = None if shape is None else tuple(shape)
            self.dtype = None if dtype is None else np.dtype(dtype)
            self._np_random = None

        @property
        def np_random(self):
            if self._np_random is None:
                self.seed()
            return self._np_random

        def seed(self, seed=None):
            self._np_random = np.random.RandomState(seed)
            return [seed]

    class Box(_Space):
        """gym 0.21 semantics: default dtype float32, bounds broadcast to `shape` and cast to dtype"""

        def __init__(self, low, high, shape=None, dtype=np.float32):
            dtype = np.dtype(dtype)
            if shape is None:
                shape = np.asarray(low).shape
            shape = tuple(shape)
            self.low = np.broadcast_to(np.asarray(low), shape).astype(dtype)
            self.high = np.broadcast_to(np.asarray(high), shape).astype(dtype)
            super().__init__(shape, dtype)

        def sample(self):
            return self.np_random.uniform(self.low, self.high).astype(self.dtype)

        def contains(self, x):
            x = np.asarray(x)
            return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))

    class Discrete(_Space):
        def __init__(self, n):
            self.n = int(n)
            super().__init__((), np.int64)

        def sample(self):
            return int(self.np_random.randint(self.n))

        def contains(self, x):
            return 0 <= int(x) < self.n

    class MultiDiscrete(_Space):
        def __init__(self, nvec):
            self.nvec = np.asarray(nvec, dtype=np.int64)
            super().__init__(self.nvec.shape, np.int64)

    class Dict(_Space):
        def __init__(self, spaces):
            self.spaces = dict(spaces)
            super().__init__(None, None)

        def keys(self):
            return self.spaces.keys()

        def __getitem__(self, k):
            return self.spaces[k]

        def __contains__(self, k):
            return k in self.spaces
