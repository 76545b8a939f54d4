"""Import harness for the UNMODIFIED reference env (test infrastructure, build container only).

This file is test infrastructure.  It is never imported by the product package,
never travels into a measured path, and only works where ``/root/reference``
exists (the build container).  It is used by ``oracle/gen_golden.py`` to record
golden input/output vectors from the reference's own ``ReferenceModel`` and by a
few ``-m "not gpu"`` tests that cross-check the C restatement against the live
reference when it is present (they skip on the GPU box).

The reference file (``src/environments/reference_model_multi_agent.py:7,11``)
imports ``gymnasium`` and ``ray.rllib.env.multi_agent_env.MultiAgentEnv``; neither
package is installed in this image and there is no network.  The env only uses
``gym.spaces.{Box,Discrete,MultiBinary}`` (attributes shape/dtype/low/high/n and
``contains``) and an empty base class, so we register minimal stand-ins in
``sys.modules`` before importing.  No reference source is copied: the module is
imported from where it lies.
"""

from __future__ import annotations

import os
import sys
import types

import numpy as np

REFERENCE_ROOT = os.environ.get("MAPF_REFERENCE_ROOT", "/root/reference")


def reference_available() -> bool:
    return os.path.isfile(os.path.join(REFERENCE_ROOT, "src", "environments", "reference_model_multi_agent.py"))


# ----------------------------------------------------------------------------
# minimal gymnasium.spaces / ray stand-ins (only what the env touches)
# ----------------------------------------------------------------------------
class _Space:
    shape = ()
    dtype = None

    def contains(self, x):  # pragma: no cover - overridden
        raise NotImplementedError

    def sample(self):  # pragma: no cover - not used by the harness
        raise NotImplementedError


class _Box(_Space):
    def __init__(self, low, high, shape=None, dtype=np.float32):
        self.dtype = np.dtype(dtype)
        if shape is None:
            shape = np.asarray(low).shape
        self.shape = tuple(int(s) for s in shape)
        self.low = np.broadcast_to(np.asarray(low, dtype=self.dtype), self.shape).copy()
        self.high = np.broadcast_to(np.asarray(high, dtype=self.dtype), self.shape).copy()

    def contains(self, x):
        x = np.asarray(x)
        if x.shape != self.shape:
            return False
        if not np.can_cast(x.dtype, self.dtype):
            return False
        return bool(np.all(x >= self.low) and np.all(x <= self.high))


class _Discrete(_Space):
    def __init__(self, n, start=0):
        self.n = int(n)
        self.start = int(start)
        self.shape = ()
        self.dtype = np.dtype(np.int64)

    def contains(self, x):
        return self.start <= int(x) < self.start + self.n

    def sample(self):
        return int(np.random.randint(self.start, self.start + self.n))


class _MultiBinary(_Space):
    def __init__(self, n):
        self.n = n
        self.shape = (int(n),) if np.isscalar(n) else tuple(int(v) for v in n)
        self.dtype = np.dtype(np.int8)

    def contains(self, x):
        x = np.asarray(x)
        return x.shape == self.shape and bool(np.all((x == 0) | (x == 1)))


class _MultiDiscrete(_Space):
    def __init__(self, nvec):
        self.nvec = np.asarray(nvec, dtype=np.int64)
        self.shape = self.nvec.shape
        self.dtype = np.dtype(np.int64)

    def contains(self, x):
        x = np.asarray(x)
        return x.shape == self.shape and bool(np.all((x >= 0) & (x < self.nvec)))


class _GymEnv:  # gym.Env: only used as a base class (reference_model_single_agent.py:51,82)
    def __init__(self, *a, **k):
        pass


def _install_stand_ins() -> None:
    if "gymnasium" not in sys.modules:
        gym = types.ModuleType("gymnasium")
        spaces = types.ModuleType("gymnasium.spaces")
        spaces.Box = _Box
        spaces.Discrete = _Discrete
        spaces.MultiBinary = _MultiBinary
        spaces.Space = _Space
        spaces.MultiDiscrete = _MultiDiscrete
        gym.spaces = spaces
        gym.Env = _GymEnv
        sys.modules["gymnasium"] = gym
        sys.modules["gymnasium.spaces"] = spaces
    if "ray" not in sys.modules:
        ray = types.ModuleType("ray")
        rllib = types.ModuleType("ray.rllib")
        env = types.ModuleType("ray.rllib.env")
        mae = types.ModuleType("ray.rllib.env.multi_agent_env")

        class MultiAgentEnv:  # empty base, as used at MA-env:17,35
            def __init__(self):
                pass

        mae.MultiAgentEnv = MultiAgentEnv
        ray.rllib = rllib
        rllib.env = env
        env.multi_agent_env = mae
        sys.modules["ray"] = ray
        sys.modules["ray.rllib"] = rllib
        sys.modules["ray.rllib.env"] = env
        sys.modules["ray.rllib.env.multi_agent_env"] = mae


_REF = None


def load_reference():
    """Return (ReferenceModel class, get_grid module) of the unmodified reference."""
    global _REF
    if _REF is not None:
        return _REF
    if not reference_available():
        raise RuntimeError(f"reference tree not present at {REFERENCE_ROOT}")
    os.environ.setdefault("MPLBACKEND", "Agg")
    sys.dont_write_bytecode = True
    _install_stand_ins()
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    from src.environments import get_grid as ref_get_grid  # type: ignore
    from src.environments.reference_model_multi_agent import ReferenceModel as RefModel  # type: ignore

    _REF = (RefModel, ref_get_grid)
    return _REF


class injected_grid:
    """Context manager: make the reference's ``get_grid.get_grid`` return ``grid`` for any name.

    The env looks the function up on the module at call time (MA-env:80), so replacing the
    module attribute is enough; nothing in the reference is edited.
    """

    def __init__(self, grid: np.ndarray):
        self.grid = np.asarray(grid, dtype=np.uint8)
        self._saved = None

    def __enter__(self):
        _, gg = load_reference()
        self._saved = gg.get_grid
        grid = self.grid
        gg.get_grid = lambda name: grid.copy()
        return self

    def __exit__(self, *exc):
        _, gg = load_reference()
        gg.get_grid = self._saved
        return False


def make_reference_single_agent_env(env_config: dict, grid: np.ndarray | None = None):
    """The reference's single-agent (CTE) env, optionally on an injected synthetic grid."""
    load_reference()
    from src.environments.reference_model_single_agent import ReferenceModel as SingleRef  # type: ignore

    if grid is None:
        return SingleRef(dict(env_config))
    cfg = dict(env_config)
    cfg.setdefault("env_name", "synthetic")
    with injected_grid(grid):
        return SingleRef(cfg)


def make_reference_env(env_config: dict, grid: np.ndarray | None = None):
    """Construct the reference env, optionally on an injected synthetic grid."""
    RefModel, _ = load_reference()
    if grid is None:
        return RefModel(dict(env_config))
    cfg = dict(env_config)
    cfg.setdefault("env_name", "synthetic")
    with injected_grid(grid):
        return RefModel(cfg)
