"""Batched tensor API over the HIP step engine: B independent ``ReferenceModel`` envs on one GPU.

``VecReferenceModel(env_config)`` takes the reference's ``env_config`` keys with the reference's
defaults (MA-env = reference src/environments/reference_model_multi_agent.py:38-61) plus these
extension keys:

    num_envs        B (default 1)
    device          torch device (default "cuda:0")
    grid            uint8 [H,W] (shared by every env) or [B,H,W]; default: named grid ``env_name``
    seeds           one NumPy seed per env; default ``seed + b`` when ``seed`` is given, else OS entropy
    rng_words       uint64 [B,6] explicit PCG64 states (overrides seeds)
    fixed_starts / fixed_goals   int16 [B,N,2] or [N,2] for ``deterministic`` on synthetic grids
    lanes_per_env   engine tuning knob (0 = auto)
    jit_specialize  opt-in: when no prebuilt specialisation matches, compile the step kernels for exactly this
                    configuration when the engine is created (hiprtc, 2-5 s, cached per process); ``launch_info()``
                    tells whether it took (``jit``) and why not (``jit_note``)
    force_generic_kernel   engine knob: use the runtime-config step kernel even when a compile-time
                    specialisation (BASELINE.json shapes) matches
    force_sequential_reset   engine knob (tests): in-kernel resets always use the sequential restatement of
                    ``generate_starts_goals`` instead of the lane-parallel one

All hot-path calls are asynchronous on the current torch stream; outputs are preallocated device
tensors that are overwritten by the next call (clone them to keep them).
"""

from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib as L
from . import get_grid as grid_tables

INFO_ALL_KEYS = (
    "goals_reached_step", "goals_reached_total", "blocking_count_step", "blocking_count_total",
    "deadlock_step", "livelock_step", "deadlock_event_step", "livelock_event_step",
    "deadlock_events_total", "livelock_events_total", "deadlock_steps_total", "livelock_steps_total",
    "completion_ratio", "throughput",
)


def pcg64_words(seed) -> np.ndarray:
    """uint64[6] PCG64 state of ``np.random.default_rng(seed)`` (SeedSequence expansion stays in NumPy)."""
    st = np.random.default_rng(seed).bit_generator.state
    s, inc = int(st["state"]["state"]), int(st["state"]["inc"])
    m = (1 << 64) - 1
    return np.array([s >> 64, s & m, inc >> 64, inc & m, int(st["has_uint32"]), int(st["uinteger"])], dtype=np.uint64)


def metrics_from_sums(s, num_agents: int, lifelong: bool) -> dict:
    """Means over finished episodes from the int64 accumulator vector (exact rational arithmetic in float64)."""
    n = float(s[L.ACC_EPISODES])
    if n == 0:
        return {"episodes": 0}
    m = {
        "episodes": int(s[L.ACC_EPISODES]),
        "goals_reached": s[L.ACC_GOALS_REACHED] / n,
        "blocking_count": s[L.ACC_BLOCKING_COUNT] / n,
        "deadlock_count": s[L.ACC_DEADLOCK_COUNT] / n,
        "livelock_count": s[L.ACC_LIVELOCK_COUNT] / n,
        "deadlock_steps": s[L.ACC_DEADLOCK_STEPS] / n,
        "livelock_steps": s[L.ACC_LIVELOCK_STEPS] / n,
        "episode_len_mean": s[L.ACC_EPISODE_STEPS] / n,
    }
    completion = s[L.ACC_COMPLETED_AGENTS] / (n * num_agents)
    if lifelong:  # SuccessRateCallback logs the completion ratio as success in lifelong mode (callbacks.py:150-155)
        m["success_rate"] = completion
        m["completion_ratio"] = completion
        # every lifelong episode runs to the step limit, so the mean of goals/steps is the ratio of the sums
        m["throughput"] = s[L.ACC_GOALS_REACHED] / max(float(s[L.ACC_EPISODE_STEPS]), 1.0)
    else:
        m["success_rate"] = s[L.ACC_SUCCESSES] / n
    return m


def config_flags(cfg: dict) -> int:
    f = 0
    if cfg.get("normalize_goal_delta", True):
        f |= L.FLAG_NORMALIZE_GOAL_DELTA
    if cfg.get("include_goal_distance", False):
        f |= L.FLAG_GOAL_DISTANCE
    if bool(cfg.get("include_action_mask_in_obs", False)):
        f |= L.FLAG_ACTION_MASK
    if bool(cfg.get("include_blocking_pressure_in_obs", True)):
        f |= L.FLAG_BLOCKING_PRESSURE
    if bool(cfg.get("lifelong_mapf", False)):
        f |= L.FLAG_LIFELONG
    if bool(cfg.get("enable_lock_metrics", True)):
        f |= L.FLAG_LOCK_METRICS
    if cfg.get("deterministic", False):
        f |= L.FLAG_DETERMINISTIC
    if cfg.get("force_pair_walk", False):  # engine knob: all-pairs walk instead of the LDS cell map at N > 16
        f |= L.FLAG_NO_CELL_MAP
    if cfg.get("force_sequential_reset", False):  # engine knob: in-kernel resets through the sequential sampler only
        f |= L.FLAG_SEQUENTIAL_RESET
    if cfg.get("jit_specialize", False):  # opt-in: compile the step kernels for exactly this configuration (hiprtc)
        f |= L.FLAG_JIT_SPECIALIZE
    if cfg.get("force_generic_kernel", False):  # engine knob: skip the compile-time specialised step kernel
        f |= L.FLAG_GENERIC_KERNEL
    # engine knobs (tests): which build of the small-group step kernels -- "dense" = the 128-register one mapf_create
    # otherwise picks for grids of more than three waves per SIMD, "sparse" = never that one; and where the
    # runtime-config kernels pre-draw placements ("sampler_workgroups" instead of slices in the env workgroups)
    budget = cfg.get("register_budget")
    if budget not in (None, "dense", "sparse"):
        raise ValueError(f"register_budget must be None, 'dense' or 'sparse', got {budget!r}")
    if budget == "dense":
        f |= L.FLAG_FORCE_DENSE
    if budget == "sparse":
        f |= L.FLAG_FORCE_SPARSE
    if cfg.get("background_draw") == "sampler_workgroups":
        f |= L.FLAG_SAMPLER_WORKGROUPS
    if cfg.get("wide_kernel") == "two_wave":  # engine knob: 64-lane groups on the two-wave kernel with the LDS cell map
        f |= L.FLAG_TWO_WAVE_WIDE
    if cfg.get("small_group_observation") == "table_walk":  # engine knob: round 3's observation wave (A/B, tests)
        f |= L.FLAG_TABLE_WALK_OBS
    if cfg.get("small_group_rows") == "off":  # engine knob: round 3's three-wave kernel (no bit rows at all)
        f |= L.FLAG_NO_BIT_ROWS
    return f


# the current stream's raw handle without building a torch.cuda.Stream object per call (0.5 us of a 7 us Python step)
_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None) or (lambda idx: torch.cuda.current_stream(idx).cuda_stream)


class VecReferenceModel:
    def __init__(self, env_config: dict):
        cfg = dict(env_config)
        self.env_config = cfg
        self._lib = L.load()  # raises if the HIP library is not built: no CPU fallback
        self.device = torch.device(cfg.get("device", "cuda:0"))
        if self.device.type != "cuda":
            raise ValueError("VecReferenceModel runs on a HIP device only (device='cuda:N')")
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.num_envs = B = int(cfg.get("num_envs", 1))
        self.num_agents = N = int(cfg.get("num_agents", 2))
        self.sensor_range = int(cfg.get("sensor_range", 1))
        self.steps_per_episode = int(cfg.get("steps_per_episode", 100))
        self.lifelong_mapf = bool(cfg.get("lifelong_mapf", False))
        self.deterministic = bool(cfg.get("deterministic", False))
        self.info_mode = str(cfg.get("info_mode", "lite")).lower()
        if self.info_mode not in {"lite", "full"}:
            raise ValueError(f"Unsupported info_mode '{self.info_mode}'. Expected 'lite' or 'full'.")

        grid = cfg.get("grid", None)
        if grid is None:
            grid = grid_tables.get_grid(cfg["env_name"])
        grid = np.ascontiguousarray(grid, dtype=np.uint8)
        if grid.ndim == 2:
            shared, self.grids = 1, grid[None]
        elif grid.ndim == 3 and grid.shape[0] == B:
            shared, self.grids = 0, grid
        else:
            raise ValueError("grid must be [H,W] or [num_envs,H,W]")
        H, W = int(self.grids.shape[1]), int(self.grids.shape[2])
        self.grid_shape = (H, W)

        c = L.MapfConfig(
            B, H, W, N, self.sensor_range, self.steps_per_episode, config_flags(cfg),
            int(cfg.get("deadlock_window_steps", 8)), int(cfg.get("livelock_window_steps", 16)),
            int(cfg.get("lock_nearby_manhattan", 2)), int(cfg.get("lock_min_neighbors", 1)),
            float(cfg.get("lock_progress_epsilon", 1)), int(self.device.index), int(cfg.get("lanes_per_env", 0)),
        )
        self._cfg = c
        self.obs_len = int(self._lib.mapf_obs_len(C.byref(c)))
        self.livelock_window = max(1, int(cfg.get("livelock_window_steps", 16)))
        h = C.c_void_p()
        rc = self._lib.mapf_create(C.byref(c), C.byref(h))
        if rc != L.MAPF_OK:
            raise ValueError(f"mapf_create failed ({rc}): {self._lib.mapf_last_error(None).decode()}")
        self._h = h
        self._check(self._lib.mapf_set_grids(h, self.grids.ctypes.data_as(C.c_void_p), shared), ValueError)

        # RNG: one NumPy PCG64 stream per env (MA-env:74-78)
        if cfg.get("rng_words", None) is not None:
            words = np.ascontiguousarray(cfg["rng_words"], dtype=np.uint64).reshape(B, 6)
        else:
            seeds = cfg.get("seeds", None)
            if seeds is None:
                seed = cfg.get("seed", None)
                seeds = [None] * B if seed is None else [int(seed) + b for b in range(B)]
            if len(seeds) != B:
                raise ValueError("need one seed per env")
            words = np.stack([pcg64_words(s) for s in seeds])
        self._check(self._lib.mapf_set_rng_state(h, words.ctypes.data_as(C.c_void_p)))

        with torch.cuda.device(self.device):
            dev = self.device
            Lo = self.obs_len
            self._obs = torch.zeros((B, N, Lo), dtype=torch.float32, device=dev)
            self._final_obs = torch.zeros((B, N, Lo), dtype=torch.float32, device=dev)
            # the small per-step outputs live in ONE allocation (256-byte aligned sections): a wave's stores to them
            # then share address translations instead of touching five separately mapped tensors
            shapes = (("_rewards", (B, N), torch.float32), ("_info_all", (B, L.INFO_ALL), torch.float32),
                      ("_info_agent", (B, N, 2), torch.uint8), ("_terminated", (B,), torch.uint8),
                      ("_truncated", (B,), torch.uint8))
            if cfg.get("separate_output_tensors", False):  # (A/B knob)
                for name, shape, dt in shapes:
                    setattr(self, name, torch.zeros(shape, dtype=dt, device=dev))
            else:
                sizes = [int(np.prod(shape)) * torch.empty((), dtype=dt).element_size() for _, shape, dt in shapes]
                offs, total = [], 0
                for sz in sizes:
                    offs.append(total)
                    total += (sz + 255) & ~255
                self._out_blob = torch.zeros((total,), dtype=torch.uint8, device=dev)
                for (name, shape, dt), off, sz in zip(shapes, offs, sizes):
                    setattr(self, name, self._out_blob[off:off + sz].view(dt).view(shape))

        self._act_shape = torch.Size((B, N))
        self._dev_index = int(self.device.index)
        self._step_fn = self._lib.mapf_step_bound  # (four arguments per call instead of eleven: mapf_bind_outputs)
        self._check(self._lib.mapf_bind_outputs(self._h, self._obs.data_ptr(), self._rewards.data_ptr(), self._terminated.data_ptr(),
                                                self._truncated.data_ptr(), self._info_all.data_ptr(), self._info_agent.data_ptr()))
        self._step_out = {"obs": self._obs, "rewards": self._rewards, "terminated": self._terminated, "truncated": self._truncated,
                          "info_all": self._info_all, "info_agent": self._info_agent, "final_obs": None}
        if self.deterministic:
            # fixed start/goal tables (MA-env:124-132)
            fs, fg = cfg.get("fixed_starts", None), cfg.get("fixed_goals", None)
            if fs is None or fg is None:
                s = grid_tables.get_start_positions(cfg["env_name"], N)
                g = grid_tables.get_goal_positions(cfg["env_name"], N)
                fs = np.array([s[f"agent_{i}"] for i in range(N)], dtype=np.int16)
                fg = np.array([g[f"agent_{i}"] for i in range(N)], dtype=np.int16)
            fs = np.ascontiguousarray(np.broadcast_to(np.asarray(fs, np.int16).reshape(-1, N, 2), (B, N, 2)))
            fg = np.ascontiguousarray(np.broadcast_to(np.asarray(fg, np.int16).reshape(-1, N, 2), (B, N, 2)))
            self._check(self._lib.mapf_set_fixed_starts_goals(
                h, fs.ctypes.data_as(C.c_void_p), fg.ctypes.data_as(C.c_void_p)), ValueError)
        else:
            # the reference ctor draws one generate_starts_goals() (MA-env:133-134): same RNG consumption
            self._check(self._lib.mapf_reset(h, None, None, self._stream()))

    def set_grids(self, grid) -> None:
        """New obstacle grids for the handle's envs ([H,W] shared or [num_envs,H,W]; same shape as at creation), e.g. per
        curriculum stage.  State, streams and counters stay; pre-drawn placements are voided (their free-cell tables
        changed) and the agents' pass bits follow the new rows.  Positions must lie on free cells of the new grids before
        the next step (reset, or set_state)."""
        grid = np.ascontiguousarray(grid, dtype=np.uint8)
        shared = 1 if grid.ndim == 2 else 0
        g = grid[None] if shared else grid
        if g.shape[1:] != self.grid_shape or (not shared and g.shape[0] != self.num_envs):
            raise ValueError("grid must be [H,W] or [num_envs,H,W] with the handle's H, W")
        self._check(self._lib.mapf_set_grids(self._h, g.ctypes.data_as(C.c_void_p), shared), ValueError)
        self.grids = g

    # ------------------------------------------------------------------------------------------
    def _stream(self):
        return C.c_void_p(_raw_stream(int(self.device.index)))

    def _check(self, rc: int, exc=RuntimeError):
        if rc != L.MAPF_OK:
            raise exc(f"{self._lib.mapf_last_error(self._h).decode()} (code {rc})")

    def close(self):
        if getattr(self, "_h", None):
            self._lib.mapf_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def launch_info(self) -> dict:
        b, t, l, p = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
        special = self._lib.mapf_launch_info(self._h, C.byref(b), C.byref(t), C.byref(l), C.byref(p))
        why = C.c_char_p()
        jit = self._lib.mapf_jit_status(self._h, C.byref(why))
        return {"blocks": b.value, "threads": t.value, "lds_bytes": l.value, "lanes_per_env": p.value,
                "specialized_kernel": int(special), "jit": bool(jit), "jit_note": (why.value or b"").decode(errors="replace")}

    # ------------------------------------------------------------------------------------------
    def reset(self, env_mask: torch.Tensor | None = None) -> torch.Tensor:
        """reset() of every env (or those with env_mask != 0).  Returns obs [B,N,L] (device)."""
        mptr = None
        if env_mask is not None:
            env_mask = env_mask.to(device=self.device, dtype=torch.uint8).contiguous()
            mptr = C.c_void_p(env_mask.data_ptr())
        self._check(self._lib.mapf_reset(self._h, mptr, C.c_void_p(self._obs.data_ptr()), self._stream()))
        return self._obs

    def step(self, actions: torch.Tensor, auto_reset: bool = True, want_final_obs: bool = False,
             env_mask: torch.Tensor | None = None) -> dict:
        """One step of every env (or of those with env_mask != 0: the others are not touched at all -- state, generator,
        counters, statistics, outputs).  actions: int8 [B,N] on the env's device."""
        if actions.dtype != torch.int8 or actions.device != self.device or not actions.is_contiguous():
            actions = actions.to(device=self.device, dtype=torch.int8).contiguous()
        if actions.shape != self._act_shape:
            raise ValueError(f"actions must have shape {tuple(self._act_shape)}")
        if env_mask is None and not want_final_obs:
            # the common call of a rollout loop: every pointer but the actions' is fixed for the life of the handle and bound to
            # it once (mapf_bind_outputs); the per-call cost is one data_ptr(), the stream and a four-argument ctypes call
            rc = self._step_fn(self._h, actions.data_ptr(), 1 if auto_reset else 0, _raw_stream(self._dev_index))
            if rc != 0:
                self._check(rc)
            return self._step_out
        fo = C.c_void_p(self._final_obs.data_ptr()) if (want_final_obs and auto_reset) else None
        if env_mask is not None:
            env_mask = env_mask.to(device=self.device, dtype=torch.uint8).contiguous()
            if tuple(env_mask.shape) != (self.num_envs,):
                raise ValueError(f"env_mask must have shape {(self.num_envs,)}")
            self._check(self._lib.mapf_step_masked(
                self._h, C.c_void_p(actions.data_ptr()), C.c_void_p(env_mask.data_ptr()), C.c_void_p(self._obs.data_ptr()),
                C.c_void_p(self._rewards.data_ptr()), C.c_void_p(self._terminated.data_ptr()),
                C.c_void_p(self._truncated.data_ptr()), C.c_void_p(self._info_all.data_ptr()),
                C.c_void_p(self._info_agent.data_ptr()), fo, 1 if auto_reset else 0, self._stream()))
        else:
            self._check(self._lib.mapf_step(
                self._h, C.c_void_p(actions.data_ptr()), C.c_void_p(self._obs.data_ptr()),
                C.c_void_p(self._rewards.data_ptr()), C.c_void_p(self._terminated.data_ptr()),
                C.c_void_p(self._truncated.data_ptr()), C.c_void_p(self._info_all.data_ptr()),
                C.c_void_p(self._info_agent.data_ptr()), fo, 1 if auto_reset else 0, self._stream()))
        return {
            "obs": self._obs, "rewards": self._rewards, "terminated": self._terminated, "truncated": self._truncated,
            "info_all": self._info_all, "info_agent": self._info_agent,
            "final_obs": self._final_obs if fo is not None else None,
        }

    def step_many(self, actions: torch.Tensor, obs_mode: int = 1, outputs: bool = True) -> dict:
        """T fused steps in one launch.  actions: int8 [T,B,N] on the device.  Returns fresh tensors:
        obs ([B,N,L] for obs_mode 1, [T,B,N,L] for 2, None for 0) and, when `outputs`, per-step rewards [T,B,N],
        terminated / truncated [T,B], info_all [T,B,14], info_agent [T,B,N,2]."""
        if actions.dtype != torch.int8 or actions.device != self.device or not actions.is_contiguous():
            actions = actions.to(device=self.device, dtype=torch.int8).contiguous()
        T = int(actions.shape[0])
        B, N, Lo = self.num_envs, self.num_agents, self.obs_len
        if tuple(actions.shape) != (T, B, N):
            raise ValueError(f"actions must have shape (T, {B}, {N})")
        dev = self.device
        obs = None
        if obs_mode == 1:
            obs = torch.empty((B, N, Lo), dtype=torch.float32, device=dev)
        elif obs_mode == 2:
            obs = torch.empty((T, B, N, Lo), dtype=torch.float32, device=dev)
        out = {"obs": obs, "rewards": None, "terminated": None, "truncated": None, "info_all": None, "info_agent": None}
        if outputs:
            out["rewards"] = torch.empty((T, B, N), dtype=torch.float32, device=dev)
            out["terminated"] = torch.empty((T, B), dtype=torch.uint8, device=dev)
            out["truncated"] = torch.empty((T, B), dtype=torch.uint8, device=dev)
            out["info_all"] = torch.empty((T, B, L.INFO_ALL), dtype=torch.float32, device=dev)
            out["info_agent"] = torch.empty((T, B, N, 2), dtype=torch.uint8, device=dev)

        def ptr(t):
            return None if t is None else C.c_void_p(t.data_ptr())

        self._check(self._lib.mapf_step_many(
            self._h, T, C.c_void_p(actions.data_ptr()), ptr(obs), int(obs_mode), ptr(out["rewards"]),
            ptr(out["terminated"]), ptr(out["truncated"]), ptr(out["info_all"]), ptr(out["info_agent"]), self._stream()))
        return out

    def step_many_sampled(self, T: int, seed: int, obs_in: torch.Tensor | None = None) -> dict:
        """T fused steps in one launch with the DEVICE-SIDE masked-random policy (the reference benchmark's "masked"
        mode): every agent picks uniformly among the actions its action mask allows, evaluated in-kernel on the
        observation of the previous step.  ``obs_in`` [B,N,L] is the current observation (default: the tensor the last
        ``reset`` / ``step`` returned).  Returns fresh tensors: actions [T,B,N] (what was taken), obs [T,B,N,L], rewards,
        terminated, truncated, info_all, info_agent."""
        B, N, Lo = self.num_envs, self.num_agents, self.obs_len
        dev = self.device
        src = self._obs if obs_in is None else obs_in.to(device=dev, dtype=torch.float32).contiguous()
        if tuple(src.shape) != (B, N, Lo):
            raise ValueError(f"obs_in must have shape {(B, N, Lo)}")
        out = {
            "actions": torch.empty((T, B, N), dtype=torch.int8, device=dev),
            "obs": torch.empty((T, B, N, Lo), dtype=torch.float32, device=dev),
            "rewards": torch.empty((T, B, N), dtype=torch.float32, device=dev),
            "terminated": torch.empty((T, B), dtype=torch.uint8, device=dev),
            "truncated": torch.empty((T, B), dtype=torch.uint8, device=dev),
            "info_all": torch.empty((T, B, L.INFO_ALL), dtype=torch.float32, device=dev),
            "info_agent": torch.empty((T, B, N, 2), dtype=torch.uint8, device=dev),
        }
        self._check(self._lib.mapf_step_many_sampled(
            self._h, int(T), C.c_void_p(src.data_ptr()), C.c_uint64(int(seed) & (2**64 - 1)),
            C.c_void_p(out["actions"].data_ptr()), C.c_void_p(out["obs"].data_ptr()), C.c_void_p(out["rewards"].data_ptr()),
            C.c_void_p(out["terminated"].data_ptr()), C.c_void_p(out["truncated"].data_ptr()),
            C.c_void_p(out["info_all"].data_ptr()), C.c_void_p(out["info_agent"].data_ptr()), self._stream()), ValueError)
        return out

    def observe(self) -> torch.Tensor:
        """Observation of every agent from the current state (no state change).  Returns a fresh tensor."""
        out = torch.empty_like(self._obs)
        self._check(self._lib.mapf_observe(self._h, C.c_void_p(out.data_ptr()), self._stream()))
        return out

    def assign_new_goal(self, env: int, agent: int) -> np.ndarray:
        """`_assign_new_goal(agent_idx)` of one env (MA-env:284-304) by itself, on the device: a new goal among the free
        cells that hold neither an agent nor a goal, chosen with ``rng.integers(k)`` on the env's stream.  Returns the
        new goal int16 [row, col]; raises RuntimeError when there is no such cell (the reference's error, :296-298)."""
        goal = np.zeros(2, dtype=np.int16)
        rc = self._lib.mapf_assign_new_goal(self._h, int(env), int(agent), goal.ctypes.data_as(C.c_void_p), self._stream())
        if rc == L.MAPF_ERR_NO_RESPAWN:
            self._lib.mapf_poll_error(self._h, self._stream(), None, None, None)  # (clears the latched record)
            raise RuntimeError("No valid cell available for lifelong goal reassignment.")
        self._check(rc)
        return goal

    def step_raw(self, actions_ptr: int, stream_ptr: int, auto_reset: int = 1) -> int:
        """Lowest-overhead launch for benchmarks: raw device pointer in, preallocated outputs."""
        return self._lib.mapf_step(
            self._h, actions_ptr, self._obs.data_ptr(), self._rewards.data_ptr(), self._terminated.data_ptr(),
            self._truncated.data_ptr(), self._info_all.data_ptr(), self._info_agent.data_ptr(), None, auto_reset,
            stream_ptr)

    def episode_sums(self, reset: bool = False) -> np.ndarray:
        """int64[12] sums over all finished episodes of all envs (columns: _lib.ACC_*)."""
        out = np.zeros(L.NUM_EPISODE_ACC, dtype=np.int64)
        self._check(self._lib.mapf_get_episode_stats(self._h, out.ctypes.data_as(C.c_void_p), 1 if reset else 0))
        return out

    def episode_sums_device(self, out: torch.Tensor | None = None) -> torch.Tensor:
        """The same sums as a device tensor (int64[12]), added up by one small launch on the current stream: no host
        round trip, nothing synchronized, nothing cleared (mapf_episode_stats_async)."""
        if out is None:
            out = torch.empty(L.NUM_EPISODE_ACC, dtype=torch.int64, device=self.device)
        if out.dtype != torch.int64 or out.device != self.device or out.numel() != L.NUM_EPISODE_ACC or not out.is_contiguous():
            raise ValueError(f"out must be a contiguous int64[{L.NUM_EPISODE_ACC}] tensor on {self.device}")
        self._check(self._lib.mapf_episode_stats_async(self._h, C.c_void_p(out.data_ptr()), self._stream()))
        return out

    def episode_metrics(self, reset: bool = False, sums: np.ndarray | None = None) -> dict:
        """Mean per-episode metrics under the names the reference's RLlib callbacks log
        (src/trainers/callbacks.py: success_rate :138-181, goals_reached ... livelock_steps :325-330,
        throughput / completion_ratio :331-335).  `sums` lets a multi-GPU job pass the all-reduced vector."""
        return metrics_from_sums(self.episode_sums(reset) if sums is None else sums, self.num_agents,
                                 self.lifelong_mapf)

    def poll_error(self):
        """Synchronize and raise the Python exception the reference would have raised inside step()."""
        env, agent, value = C.c_int32(-1), C.c_int32(-1), C.c_int32(0)
        rc = self._lib.mapf_poll_error(self._h, self._stream(), C.byref(env), C.byref(agent), C.byref(value))
        if rc == L.MAPF_OK:
            return
        if rc == L.MAPF_ERR_BAD_ACTION:
            err = ValueError(f"Invalid action {value.value} for agent_{agent.value}")
        elif rc == L.MAPF_ERR_NO_RESPAWN:
            err = RuntimeError("No valid cell available for lifelong goal reassignment.")
        else:
            err = RuntimeError(self._lib.mapf_last_error(self._h).decode())
        err.env_index = env.value
        raise err

    # ------------------------------------------------------------------------------------------
    def get_state(self) -> dict:
        B, N = self.num_envs, self.num_agents
        out = {
            "positions": np.zeros((B, N, 2), np.int16), "goals": np.zeros((B, N, 2), np.int16),
            "starts": np.zeros((B, N, 2), np.int16), "reached": np.zeros((B, N), np.uint8),
            "completed_once": np.zeros((B, N), np.uint8), "pressure_prev": np.zeros((B, N), np.uint8),
            "counters": np.zeros((B, L.NUM_COUNTERS), np.int32), "rng_words": np.zeros((B, 6), np.uint64),
            "lock_history": np.zeros((B, N, 3), np.uint64),
            "distance_ring": np.zeros((B, self.livelock_window, N), np.int16),
        }
        s = L.MapfState(**{k: v.ctypes.data_as(C.c_void_p) for k, v in out.items()})
        self._check(self._lib.mapf_get_state(self._h, C.byref(s)))
        return out

    def set_state(self, *, positions=None, goals=None, starts=None, reached=None, completed_once=None,
                  pressure_prev=None, counters=None, rng_words=None, lock_history=None, distance_ring=None,
                  clear_episode: bool = False):
        """Overwrite parts of the env state (what the reference's tests do by poking private arrays).
        clear_episode=True also zeroes flags, counters and lock tracking, like the tests' _set_state helpers."""
        B, N = self.num_envs, self.num_agents
        if clear_episode:
            reached = np.zeros((B, N), np.uint8) if reached is None else reached
            completed_once = np.zeros((B, N), np.uint8) if completed_once is None else completed_once
            pressure_prev = np.zeros((B, N), np.uint8) if pressure_prev is None else pressure_prev
            lock_history = np.zeros((B, N, 3), np.uint64) if lock_history is None else lock_history
            if counters is None:
                counters = self.get_state()["counters"]
                counters[:, : L.CTR_EPISODES_DONE] = 0
        keep = []

        def ptr(a, dtype, shape):
            if a is None:
                return None
            arr = np.ascontiguousarray(np.asarray(a, dtype=dtype).reshape(shape))
            keep.append(arr)
            return arr.ctypes.data_as(C.c_void_p)

        s = L.MapfState(
            ptr(positions, np.int16, (B, N, 2)), ptr(goals, np.int16, (B, N, 2)), ptr(starts, np.int16, (B, N, 2)),
            ptr(reached, np.uint8, (B, N)), ptr(completed_once, np.uint8, (B, N)), ptr(pressure_prev, np.uint8, (B, N)),
            ptr(counters, np.int32, (B, L.NUM_COUNTERS)), ptr(rng_words, np.uint64, (B, 6)),
            ptr(lock_history, np.uint64, (B, N, 3)), ptr(distance_ring, np.int16, (B, self.livelock_window, N)),
        )
        self._check(self._lib.mapf_set_state(self._h, C.byref(s)), ValueError)
