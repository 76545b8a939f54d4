"""gymnasium.spaces / RLlib base-class resolution for the drop-in facade.

When ``gymnasium`` (and ``ray``) are installed -- as in the reference's deployment -- the real
classes are used, so RLlib sees genuine ``gymnasium`` spaces and a genuine ``MultiAgentEnv``.
This image ships neither package, so small stand-ins with the attributes the env surface needs
(shape, dtype, low, high, n, contains, sample) keep the facade importable and testable.
"""

from __future__ import annotations

import numpy as np

try:  # pragma: no cover - not installed in the build image
    import gymnasium as _gym

    Box = _gym.spaces.Box
    Discrete = _gym.spaces.Discrete
    MultiBinary = _gym.spaces.MultiBinary
    MultiDiscrete = _gym.spaces.MultiDiscrete
    GymEnv = _gym.Env
    HAVE_GYMNASIUM = True
except ImportError:
    HAVE_GYMNASIUM = False

    class _Space:
        shape = ()
        dtype = None
        _rng = None

        def seed(self, seed=None):
            self._rng = np.random.default_rng(seed)

        def _generator(self):
            if self._rng is None:
                self._rng = np.random.default_rng()
            return self._rng

    class Box(_Space):
        def __init__(self, low, high, shape=None, dtype=np.float32):
            self.dtype = np.dtype(dtype)
            if shape is None:
                shape = np.broadcast(np.asarray(low), np.asarray(high)).shape
            self.shape = tuple(int(s) for s in shape)
            self.low = np.broadcast_to(np.asarray(low, dtype=self.dtype), self.shape).copy()
            self.high = np.broadcast_to(np.asarray(high, dtype=self.dtype), self.shape).copy()

        def contains(self, x) -> bool:
            x = np.asarray(x)
            return (x.shape == self.shape and np.can_cast(x.dtype, self.dtype)
                    and bool(np.all(x >= self.low)) and bool(np.all(x <= self.high)))

        def sample(self):
            return self._generator().uniform(self.low, self.high).astype(self.dtype)

    class Discrete(_Space):
        def __init__(self, n, start=0):
            self.n = int(n)
            self.start = int(start)
            self.shape = ()
            self.dtype = np.dtype(np.int64)

        def contains(self, x) -> bool:
            try:
                v = int(x)
            except (TypeError, ValueError):
                return False
            return self.start <= v < self.start + self.n

        def sample(self):
            return int(self._generator().integers(self.start, self.start + self.n))

    class MultiBinary(_Space):
        def __init__(self, n):
            self.n = n
            self.shape = (int(n),) if np.isscalar(n) else tuple(int(v) for v in n)
            self.dtype = np.dtype(np.int8)

        def contains(self, x) -> bool:
            x = np.asarray(x)
            return x.shape == self.shape and bool(np.all((x == 0) | (x == 1)))

        def sample(self):
            return self._generator().integers(0, 2, size=self.shape).astype(self.dtype)


    class MultiDiscrete(_Space):
        def __init__(self, nvec):
            self.nvec = np.asarray(nvec, dtype=np.int64)
            self.shape = self.nvec.shape
            self.dtype = np.dtype(np.int64)

        def contains(self, x) -> bool:
            x = np.asarray(x)
            return x.shape == self.shape and bool(np.all((x >= 0) & (x < self.nvec)))

        def sample(self):
            return self._generator().integers(0, self.nvec).astype(self.dtype)

    class GymEnv:  # gym.Env stand-in: a plain base class
        def __init__(self, *a, **k):
            pass


try:  # pragma: no cover - not installed in the build image
    from ray.rllib.env.multi_agent_env import MultiAgentEnv

    HAVE_RLLIB = True
except ImportError:
    HAVE_RLLIB = False

    class MultiAgentEnv:  # same role as RLlib's base: an empty parent with agent-id bookkeeping
        def __init__(self):
            pass

        def get_agent_ids(self):
            return set(getattr(self, "agents", []))
