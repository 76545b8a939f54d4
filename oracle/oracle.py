"""ctypes binding of the C oracle (oracle/mapf_oracle.c).  TEST INFRASTRUCTURE ONLY.

Only tests/, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
module; the product package never does.  ``build()`` compiles the shared object with gcc if it
is missing or stale.
"""

from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libmapf_oracle.so")

FLAG_NORMALIZE_GOAL_DELTA = 1
FLAG_GOAL_DISTANCE = 2
FLAG_ACTION_MASK = 4
FLAG_BLOCKING_PRESSURE = 8
FLAG_LIFELONG = 16
FLAG_LOCK_METRICS = 32
FLAG_DETERMINISTIC = 64

INFO_ALL = 14
INFO_ALL_KEYS = (
    "goals_reached_step",
    "goals_reached_total",
    "blocking_count_step",
    "blocking_count_total",
    "deadlock_step",
    "livelock_step",
    "deadlock_event_step",
    "livelock_event_step",
    "deadlock_events_total",
    "livelock_events_total",
    "deadlock_steps_total",
    "livelock_steps_total",
    "completion_ratio",
    "throughput",
)

OK, ERR_BAD_ACTION, ERR_FEW_FREE, ERR_NO_RESPAWN, ERR_CONFIG = 0, -1, -2, -3, -4


class MoConfig(C.Structure):
    _fields_ = [
        ("height", C.c_int32),
        ("width", C.c_int32),
        ("num_agents", C.c_int32),
        ("sensor_range", C.c_int32),
        ("steps_per_episode", C.c_int32),
        ("flags", C.c_uint32),
        ("deadlock_window_steps", C.c_int32),
        ("livelock_window_steps", C.c_int32),
        ("lock_nearby_manhattan", C.c_int32),
        ("lock_min_neighbors", C.c_int32),
        ("lock_progress_epsilon", C.c_double),
    ]


class MoStateView(C.Structure):
    _fields_ = [
        ("positions", C.POINTER(C.c_int16)),
        ("goals", C.POINTER(C.c_int16)),
        ("starts", C.POINTER(C.c_int16)),
        ("reached", C.POINTER(C.c_uint8)),
        ("completed_once", C.POINTER(C.c_uint8)),
        ("pressure_prev", C.POINTER(C.c_float)),
        ("occupancy_owner", C.POINTER(C.c_int16)),
        ("goal_owner", C.POINTER(C.c_int16)),
        ("hist_goal_progress", C.POINTER(C.c_uint8)),
        ("hist_moved", C.POINTER(C.c_uint8)),
        ("hist_failed_move", C.POINTER(C.c_uint8)),
        ("hist_distance", C.POINTER(C.c_int16)),
        ("hist_count", C.POINTER(C.c_int32)),
        ("hist_head", C.POINTER(C.c_int32)),
        ("step_count", C.POINTER(C.c_int32)),
        ("episode_blocking_count", C.POINTER(C.c_double)),
        ("episode_goals_reached_total", C.POINTER(C.c_double)),
        ("episode_deadlock_events", C.POINTER(C.c_double)),
        ("episode_livelock_events", C.POINTER(C.c_double)),
        ("episode_deadlock_steps", C.POINTER(C.c_double)),
        ("episode_livelock_steps", C.POINTER(C.c_double)),
        ("deadlock_state_prev", C.POINTER(C.c_uint8)),
        ("livelock_state_prev", C.POINTER(C.c_uint8)),
        ("n_free", C.c_int32),
        ("free_positions", C.POINTER(C.c_int16)),
        ("hist_size", C.c_int32),
    ]


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "mapf_oracle.c")
    hdr = os.path.join(_HERE, "mapf_oracle.h")
    stale = (
        force
        or not os.path.exists(_SO)
        or os.path.getmtime(_SO) < max(os.path.getmtime(src), os.path.getmtime(hdr))
    )
    if stale:
        subprocess.run(["make", "-C", _HERE, "-s", "-B"], check=True)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    so = os.environ.get("MAPF_ORACLE_SO") or build()  # MAPF_ORACLE_SO: the sanitizer build (make asan), tests/test_oracle_asan.py
    L = C.CDLL(so)
    vp, i32, u64, u32 = C.c_void_p, C.c_int32, C.c_uint64, C.c_uint32
    L.mo_obs_len.restype = C.c_int
    L.mo_obs_len.argtypes = [C.POINTER(MoConfig)]
    L.mo_create.restype = vp
    L.mo_create.argtypes = [C.POINTER(MoConfig), vp]
    L.mo_destroy.argtypes = [vp]
    L.mo_set_rng.argtypes = [vp, u64, u64, u64, u64, i32, u32]
    L.mo_get_rng.argtypes = [vp, vp]
    L.mo_generate_starts_goals.restype = C.c_int
    L.mo_generate_starts_goals.argtypes = [vp]
    L.mo_set_fixed_starts_goals.argtypes = [vp, vp, vp]
    L.mo_reset.restype = C.c_int
    L.mo_reset.argtypes = [vp, vp]
    L.mo_step.restype = C.c_int
    L.mo_step.argtypes = [vp, vp, vp, vp, vp, vp, vp]
    L.mo_get_obs.argtypes = [vp, C.c_int, vp]
    L.mo_get_action_mask.argtypes = [vp, vp, vp]
    L.mo_assign_new_goal.restype = C.c_int
    L.mo_assign_new_goal.argtypes = [vp, C.c_int]
    L.mo_view.argtypes = [vp, C.POINTER(MoStateView)]
    L.mo_rebuild_owner_maps.argtypes = [vp]
    L.mo_reset_lock_tracking.argtypes = [vp]
    L.mo_batch_create.restype = vp
    L.mo_batch_create.argtypes = [C.POINTER(MoConfig), i32, vp]
    L.mo_batch_destroy.argtypes = [vp]
    L.mo_batch_env.restype = vp
    L.mo_batch_env.argtypes = [vp, i32]
    L.mo_batch_step.restype = C.c_int
    L.mo_batch_step.argtypes = [vp, vp, C.c_int, vp, vp, vp, vp, vp, vp, vp, vp]
    L.mo_batch_reset.restype = C.c_int
    L.mo_batch_reset.argtypes = [vp, vp]
    L.mo_rng_bounded.restype = u64
    L.mo_rng_bounded.argtypes = [vp, u64]
    L.mo_rng_choice_noreplace.argtypes = [vp, C.c_int64, C.c_int64, vp]
    L.moc_create.restype = vp
    L.moc_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, vp]
    L.moc_destroy.argtypes = [vp]
    L.moc_obs_len.restype = C.c_int
    L.moc_obs_len.argtypes = [vp]
    L.moc_set_rng.argtypes = [vp, u64, u64, u64, u64, i32, u32]
    L.moc_get_rng.argtypes = [vp, vp]
    L.moc_set_fixed_starts_goals.argtypes = [vp, vp, vp]
    L.moc_generate_starts_goals.argtypes = [vp]
    L.moc_reset.argtypes = [vp, vp]
    L.moc_step.restype = C.c_int
    L.moc_step.argtypes = [vp, vp, vp, vp, vp, vp]
    L.moc_view.argtypes = [vp] + [C.POINTER(C.POINTER(C.c_int32))] * 3 + [C.POINTER(C.POINTER(C.c_uint8)),
                                                                       C.POINTER(C.POINTER(C.c_int32)),
                                                                       C.POINTER(C.POINTER(C.c_double))]
    _lib = L
    return L


def flags_from_env_config(cfg: dict) -> int:
    """Flag word from a reference-style env_config dict (defaults: MA-env:38-61)."""
    f = 0
    if cfg.get("normalize_goal_delta", True):
        f |= FLAG_NORMALIZE_GOAL_DELTA
    if cfg.get("include_goal_distance", False):
        f |= FLAG_GOAL_DISTANCE
    if cfg.get("include_action_mask_in_obs", False):
        f |= FLAG_ACTION_MASK
    if cfg.get("include_blocking_pressure_in_obs", True):
        f |= FLAG_BLOCKING_PRESSURE
    if cfg.get("lifelong_mapf", False):
        f |= FLAG_LIFELONG
    if cfg.get("enable_lock_metrics", True):
        f |= FLAG_LOCK_METRICS
    if cfg.get("deterministic", False):
        f |= FLAG_DETERMINISTIC
    return f


def make_config(grid_shape, env_config: dict) -> MoConfig:
    h, w = int(grid_shape[0]), int(grid_shape[1])
    return MoConfig(
        h,
        w,
        int(env_config.get("num_agents", 2)),
        int(env_config.get("sensor_range", 1)),
        int(env_config.get("steps_per_episode", 100)),
        flags_from_env_config(env_config),
        int(env_config.get("deadlock_window_steps", 8)),
        int(env_config.get("livelock_window_steps", 16)),
        int(env_config.get("lock_nearby_manhattan", 2)),
        int(env_config.get("lock_min_neighbors", 1)),
        float(env_config.get("lock_progress_epsilon", 1)),
    )


def pcg64_words(seed) -> np.ndarray:
    """uint64[6] = (state_hi, state_lo, inc_hi, inc_lo, has_uint32, uinteger) of default_rng(seed)."""
    st = np.random.default_rng(seed).bit_generator.state
    return pcg64_words_from_state(st)


def pcg64_words_from_state(st: dict) -> np.ndarray:
    s, inc = int(st["state"]["state"]), int(st["state"]["inc"])
    m = (1 << 64) - 1
    return np.array([s >> 64, s & m, inc >> 64, inc & m, int(st["has_uint32"]), int(st["uinteger"])], dtype=np.uint64)


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class _EnvHandle:
    """Operations shared by a stand-alone env and an env borrowed from a batch."""

    def __init__(self, handle, cfg: MoConfig, owns: bool):
        self._h = handle
        self.cfg = cfg
        self._owns = owns
        self.N = cfg.num_agents
        self.H, self.W = cfg.height, cfg.width
        self.V = 2 * cfg.sensor_range + 1
        self.L = lib().mo_obs_len(C.byref(cfg))
        v = MoStateView()
        lib().mo_view(handle, C.byref(v))
        self._view = v
        self.hist_size = v.hist_size

    # -- numpy views aliasing the C arrays (mutating them mutates the env, like the reference tests do)
    def _arr(self, ptr, shape, dtype):
        n = int(np.prod(shape))
        ct = np.ctypeslib.as_ctypes_type(np.dtype(dtype))
        buf = C.cast(ptr, C.POINTER(ct * n)).contents
        return np.frombuffer(buf, dtype=dtype).reshape(shape)

    @property
    def positions(self):
        return self._arr(self._view.positions, (self.N, 2), np.int16)

    @property
    def goals(self):
        return self._arr(self._view.goals, (self.N, 2), np.int16)

    @property
    def starts(self):
        return self._arr(self._view.starts, (self.N, 2), np.int16)

    @property
    def reached(self):
        return self._arr(self._view.reached, (self.N,), np.uint8)

    @property
    def completed_once(self):
        return self._arr(self._view.completed_once, (self.N,), np.uint8)

    @property
    def pressure_prev(self):
        return self._arr(self._view.pressure_prev, (self.N,), np.float32)

    @property
    def occupancy_owner(self):
        return self._arr(self._view.occupancy_owner, (self.H, self.W), np.int16)

    @property
    def goal_owner(self):
        return self._arr(self._view.goal_owner, (self.H, self.W), np.int16)

    @property
    def free_positions(self):
        return self._arr(self._view.free_positions, (self._view.n_free, 2), np.int16)

    @property
    def hist(self):
        s = (self.hist_size, self.N)
        return {
            "goal_progress": self._arr(self._view.hist_goal_progress, s, np.uint8),
            "moved": self._arr(self._view.hist_moved, s, np.uint8),
            "failed_move": self._arr(self._view.hist_failed_move, s, np.uint8),
            "distance": self._arr(self._view.hist_distance, s, np.int16),
            "count": int(self._view.hist_count[0]),
            "head": int(self._view.hist_head[0]),
        }

    @property
    def step_count(self):
        return int(self._view.step_count[0])

    @step_count.setter
    def step_count(self, v):
        self._view.step_count[0] = int(v)

    def counters(self) -> dict:
        v = self._view
        return {
            "step_count": int(v.step_count[0]),
            "episode_blocking_count": float(v.episode_blocking_count[0]),
            "episode_goals_reached_total": float(v.episode_goals_reached_total[0]),
            "episode_deadlock_events": float(v.episode_deadlock_events[0]),
            "episode_livelock_events": float(v.episode_livelock_events[0]),
            "episode_deadlock_steps": float(v.episode_deadlock_steps[0]),
            "episode_livelock_steps": float(v.episode_livelock_steps[0]),
            "deadlock_state_prev": int(v.deadlock_state_prev[0]),
            "livelock_state_prev": int(v.livelock_state_prev[0]),
        }

    def set_goals_reached_total(self, v: float):
        self._view.episode_goals_reached_total[0] = float(v)

    # -- RNG
    def set_rng_words(self, words):
        w = [int(x) for x in words]
        lib().mo_set_rng(self._h, w[0], w[1], w[2], w[3], w[4], w[5])

    def seed(self, seed):
        self.set_rng_words(pcg64_words(seed))

    def rng_words(self) -> np.ndarray:
        out = np.zeros(6, dtype=np.uint64)
        lib().mo_get_rng(self._h, _ptr(out))
        return out

    def rng_bounded(self, rng_inclusive: int) -> int:
        return int(lib().mo_rng_bounded(self._h, int(rng_inclusive)))

    def rng_choice(self, pop: int, size: int) -> np.ndarray:
        out = np.zeros(size, dtype=np.int64)
        lib().mo_rng_choice_noreplace(self._h, pop, size, _ptr(out))
        return out

    # -- env API
    def generate_starts_goals(self) -> int:
        return lib().mo_generate_starts_goals(self._h)

    def set_fixed_starts_goals(self, starts, goals):
        s = np.ascontiguousarray(starts, dtype=np.int16)
        g = np.ascontiguousarray(goals, dtype=np.int16)
        lib().mo_set_fixed_starts_goals(self._h, _ptr(s), _ptr(g))

    def rebuild_owner_maps(self):
        lib().mo_rebuild_owner_maps(self._h)

    def reset_lock_tracking(self):
        lib().mo_reset_lock_tracking(self._h)

    def assign_new_goal(self, agent: int) -> int:
        return lib().mo_assign_new_goal(self._h, int(agent))

    def reset(self):
        obs = np.zeros((self.N, self.L), dtype=np.float32)
        rc = lib().mo_reset(self._h, _ptr(obs))
        return rc, obs

    def step(self, actions):
        a = np.ascontiguousarray(actions, dtype=np.int32)
        assert a.shape == (self.N,)
        obs = np.zeros((self.N, self.L), dtype=np.float32)
        rew = np.zeros(self.N, dtype=np.float32)
        done = np.zeros(2, dtype=np.uint8)
        info_all = np.zeros(INFO_ALL, dtype=np.float32)
        info_agent = np.zeros((self.N, 2), dtype=np.uint8)
        rc = lib().mo_step(self._h, _ptr(a), _ptr(obs), _ptr(rew), _ptr(done), _ptr(info_all), _ptr(info_agent))
        return rc, obs, rew, bool(done[0]), bool(done[1]), info_all, info_agent

    def get_obs(self, agent: int) -> np.ndarray:
        out = np.zeros((self.V, self.V), dtype=np.uint8)
        lib().mo_get_obs(self._h, int(agent), _ptr(out))
        return out

    def get_action_mask(self, local: np.ndarray) -> np.ndarray:
        loc = np.ascontiguousarray(local, dtype=np.uint8)
        out = np.zeros(5, dtype=np.int8)
        lib().mo_get_action_mask(self._h, _ptr(loc), _ptr(out))
        return out


class OracleEnv(_EnvHandle):
    """One env.  Mirrors the reference ctor: seeds PCG64 and (non-deterministic) draws the ctor's
    ``generate_starts_goals`` (MA-env:133-134) so the RNG stream lines up with ``ReferenceModel``."""

    def __init__(self, grid, env_config: dict, *, fixed_starts=None, fixed_goals=None, rng_words=None):
        grid = np.ascontiguousarray(grid, dtype=np.uint8)
        cfg = make_config(grid.shape, env_config)
        h = lib().mo_create(C.byref(cfg), _ptr(grid))
        if not h:
            raise ValueError("oracle: bad config")
        super().__init__(h, cfg, True)
        self.grid = grid
        if rng_words is not None:
            self.set_rng_words(rng_words)
        else:
            self.seed(env_config.get("seed", None))
        if cfg.flags & FLAG_DETERMINISTIC:
            self.set_fixed_starts_goals(fixed_starts, fixed_goals)
        else:
            rc = self.generate_starts_goals()
            if rc != OK:
                raise ValueError(f"oracle: generate_starts_goals failed rc={rc}")

    def __del__(self):
        try:
            if self._owns and self._h:
                lib().mo_destroy(self._h)
                self._h = None
        except Exception:
            pass


class OracleBatch:
    """B independent envs stepped by the scalar C loop (parity at size, cpu_baseline)."""

    def __init__(self, grids, env_config: dict, seeds=None, rng_words=None, ctor_draw: bool = True):
        grids = np.ascontiguousarray(grids, dtype=np.uint8)
        assert grids.ndim == 3
        self.B = grids.shape[0]
        self.cfg = make_config(grids.shape[1:], env_config)
        self._h = lib().mo_batch_create(C.byref(self.cfg), self.B, _ptr(grids))
        if not self._h:
            raise ValueError("oracle: bad config")
        self.N = self.cfg.num_agents
        self.L = lib().mo_obs_len(C.byref(self.cfg))
        self.envs = [_EnvHandle(lib().mo_batch_env(self._h, i), self.cfg, False) for i in range(self.B)]
        if rng_words is None:
            if seeds is None:
                seeds = np.arange(self.B)
            rng_words = np.stack([pcg64_words(int(s)) for s in seeds])
        for e, w in zip(self.envs, rng_words):
            e.set_rng_words(w)
        if ctor_draw and not (self.cfg.flags & FLAG_DETERMINISTIC):
            for e in self.envs:
                rc = e.generate_starts_goals()
                if rc != OK:
                    raise ValueError(f"oracle: generate_starts_goals failed rc={rc}")

    def __del__(self):
        try:
            if self._h:
                lib().mo_batch_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def reset(self):
        obs = np.zeros((self.B, self.N, self.L), dtype=np.float32)
        rc = lib().mo_batch_reset(self._h, _ptr(obs))
        return rc, obs

    def step(self, actions, auto_reset: bool = True, want_final_obs: bool = False, outputs: bool = True):
        a = np.ascontiguousarray(actions, dtype=np.int8)
        assert a.shape == (self.B, self.N)
        B, N, L = self.B, self.N, self.L
        if outputs:
            obs = np.zeros((B, N, L), dtype=np.float32)
            rew = np.zeros((B, N), dtype=np.float32)
            info_all = np.zeros((B, INFO_ALL), dtype=np.float32)
            info_agent = np.zeros((B, N, 2), dtype=np.uint8)
        else:
            obs = rew = info_all = info_agent = None
        term = np.zeros(B, dtype=np.uint8)
        trunc = np.zeros(B, dtype=np.uint8)
        final_obs = np.zeros((B, N, L), dtype=np.float32) if want_final_obs else None
        err_env = C.c_int32(-1)
        rc = lib().mo_batch_step(
            self._h, _ptr(a), int(auto_reset), _ptr(obs), _ptr(rew), _ptr(term), _ptr(trunc), _ptr(info_all),
            _ptr(info_agent), _ptr(final_obs), C.byref(err_env),
        )
        return {
            "rc": rc, "err_env": err_env.value, "obs": obs, "rewards": rew, "terminated": term, "truncated": trunc,
            "info_all": info_all, "info_agent": info_agent, "final_obs": final_obs,
        }

    def state(self) -> dict:
        """Snapshot of the per-env state arrays, stacked over envs."""
        es = self.envs
        return {
            "positions": np.stack([e.positions.copy() for e in es]),
            "goals": np.stack([e.goals.copy() for e in es]),
            "starts": np.stack([e.starts.copy() for e in es]),
            "reached": np.stack([e.reached.copy() for e in es]),
            "completed_once": np.stack([e.completed_once.copy() for e in es]),
            "pressure_prev": np.stack([e.pressure_prev.copy() for e in es]),
            "step_count": np.array([e.step_count for e in es], dtype=np.int32),
            "rng": np.stack([e.rng_words() for e in es]),
        }


class OracleCteEnv:
    """CPU restatement of the single-agent (CTE) sibling env (reference reference_model_single_agent.py).
    Mirrors the reference ctor: seeds PCG64 and, unless deterministic, draws generate_starts_goals once."""

    INFO_KEYS = ("blocking_count_step", "goals_reached_step", "goals_reached_total", "blocking_count_total")

    def __init__(self, grid, env_config: dict, *, fixed_starts=None, fixed_goals=None, rng_words=None):
        grid = np.ascontiguousarray(grid, dtype=np.uint8)
        self.grid = grid
        self.H, self.W = grid.shape
        self.N = int(env_config.get("num_agents", 2))
        self.deterministic = bool(env_config.get("deterministic", False))
        self._h = lib().moc_create(self.H, self.W, self.N, int(env_config.get("steps_per_episode", 100)),
                                   int(self.deterministic), float(env_config.get("blocking_penalty", -0.2)),
                                   float(env_config.get("move_after_goal_penalty", -0.05)), _ptr(grid))
        if not self._h:
            raise ValueError("oracle: bad CTE config")
        self.L = lib().moc_obs_len(self._h)
        ptrs = [C.POINTER(C.c_int32)() for _ in range(3)] + [C.POINTER(C.c_uint8)(), C.POINTER(C.c_int32)(),
                                                              C.POINTER(C.c_double)()]
        lib().moc_view(self._h, *[C.byref(p) for p in ptrs])
        self._pos, self._goals, self._starts, self._reached, self._step, self._blk = ptrs
        w = rng_words if rng_words is not None else pcg64_words(env_config.get("seed", None))
        w = [int(x) for x in w]
        lib().moc_set_rng(self._h, w[0], w[1], w[2], w[3], w[4], w[5])
        if self.deterministic:
            s = np.ascontiguousarray(fixed_starts, dtype=np.int32)
            g = np.ascontiguousarray(fixed_goals, dtype=np.int32)
            lib().moc_set_fixed_starts_goals(self._h, _ptr(s), _ptr(g))
        else:
            lib().moc_generate_starts_goals(self._h)

    def __del__(self):
        try:
            if self._h:
                lib().moc_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def _arr(self, ptr, n, dtype):
        ct = np.ctypeslib.as_ctypes_type(np.dtype(dtype))
        return np.frombuffer(C.cast(ptr, C.POINTER(ct * n)).contents, dtype=dtype)

    positions = property(lambda self: self._arr(self._pos, 2 * self.N, np.int32).reshape(self.N, 2))
    goals = property(lambda self: self._arr(self._goals, 2 * self.N, np.int32).reshape(self.N, 2))
    starts = property(lambda self: self._arr(self._starts, 2 * self.N, np.int32).reshape(self.N, 2))
    reached_once = property(lambda self: self._arr(self._reached, self.N, np.uint8))
    step_count = property(lambda self: int(self._step[0]))

    def rng_words(self):
        out = np.zeros(6, dtype=np.uint64)
        lib().moc_get_rng(self._h, _ptr(out))
        return out

    def reset(self):
        obs = np.zeros(self.L, dtype=np.float32)
        lib().moc_reset(self._h, _ptr(obs))
        return obs

    def step(self, action):
        a = np.ascontiguousarray(action, dtype=np.int32)
        obs = np.zeros(self.L, dtype=np.float32)
        rew = C.c_double(0.0)
        done = np.zeros(2, dtype=np.uint8)
        info = np.zeros(4, dtype=np.float32)
        rc = lib().moc_step(self._h, _ptr(a), _ptr(obs), C.byref(rew), _ptr(done), _ptr(info))
        return rc, obs, float(rew.value), bool(done[0]), bool(done[1]), info


class OracleCteBatch:
    """B single-agent envs stepped by one C loop (moc_run): bench.py's cpu_baseline for the single-agent workloads."""

    def __init__(self, grids, env_config: dict, seeds=None):
        grids = np.ascontiguousarray(grids, dtype=np.uint8)
        self.B = grids.shape[0]
        seeds = list(range(self.B)) if seeds is None else list(seeds)
        self.envs = [OracleCteEnv(grids[b], dict(env_config, seed=int(seeds[b]))) for b in range(self.B)]
        self.N = self.envs[0].N
        self._handles = (C.c_void_p * self.B)(*[e._h for e in self.envs])
        self._scratch = np.zeros(self.envs[0].L, dtype=np.float32)
        lib().moc_run.restype = C.c_long
        lib().moc_run.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p]

    def reset(self):
        for e in self.envs:
            e.reset()

    def set_step_counts(self, counts):
        for e, c in zip(self.envs, counts):
            e._step[0] = int(c)

    def run(self, actions, steps: int) -> int:
        a = np.ascontiguousarray(actions, dtype=np.int8)
        assert a.ndim == 3 and a.shape[1:] == (self.B, self.N)
        rc = lib().moc_run(self._handles, self.B, _ptr(a), int(a.shape[0]), int(steps), _ptr(self._scratch))
        if rc < 0:
            raise ValueError(f"oracle: moc_run failed rc={rc}")
        return int(rc)
