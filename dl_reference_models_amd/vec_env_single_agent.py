"""Batched tensor API of the single-agent (CTE) sibling env: B independent copies of the reference's
``src/environments/reference_model_single_agent.py::ReferenceModel`` ("SA-env") on one GPU.

Same construction keys as the reference (``steps_per_episode``, ``num_agents``, ``deterministic``,
``blocking_penalty`` -0.2, ``move_after_goal_penalty`` -0.05, ``seed``, ``env_name``; SA-env:84-114) plus the
extension keys of ``VecReferenceModel`` (``num_envs``, ``device``, ``grid``, ``seeds``, ``rng_words``,
``fixed_starts`` / ``fixed_goals``).
"""

from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib as L
from . import get_grid as grid_tables
from .vec_env import _raw_stream, pcg64_words

INFO_KEYS = ("blocking_count_step", "goals_reached_step", "goals_reached_total", "blocking_count_total")


class VecSingleAgentReferenceModel:
    def __init__(self, env_config: dict):
        cfg = dict(env_config)
        self._lib = L.load()
        self.device = torch.device(cfg.get("device", "cuda:0"))
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self._dev_index = int(self.device.index)
        self._out_ptrs = None
        self.num_envs = B = int(cfg.get("num_envs", 1))
        self.num_agents = N = int(cfg.get("num_agents", 2))
        self.steps_per_episode = int(cfg.get("steps_per_episode", 100))
        self.deterministic = bool(cfg.get("deterministic", False))
        grid = cfg.get("grid", None)
        if grid is None:
            grid = grid_tables.get_grid(cfg["env_name"])
        grid = np.ascontiguousarray(grid, dtype=np.uint8)
        shared, self.grids = (1, grid[None]) if grid.ndim == 2 else (0, grid)
        H, W = int(self.grids.shape[1]), int(self.grids.shape[2])
        self.grid_shape = (H, W)
        flags = L.FLAG_SINGLE_AGENT | (L.FLAG_DETERMINISTIC if self.deterministic else 0)
        c = L.MapfConfig(B, H, W, N, 0, self.steps_per_episode, flags, 1, 1, 1, 1, 0.0, int(self.device.index),
                         int(cfg.get("lanes_per_env", 0)))
        self.obs_len = int(self._lib.mapf_obs_len(C.byref(c)))
        h = C.c_void_p()
        rc = self._lib.mapf_create(C.byref(c), C.byref(h))
        if rc != L.MAPF_OK:
            raise ValueError(f"mapf_create failed ({rc}): {self._lib.mapf_last_error(None).decode()}")
        self._h = h
        self._check(self._lib.mapf_cte_configure(h, float(cfg.get("blocking_penalty", -0.2)),
                                                 float(cfg.get("move_after_goal_penalty", -0.05))))
        self._check(self._lib.mapf_set_grids(h, self.grids.ctypes.data_as(C.c_void_p), shared), ValueError)
        if cfg.get("rng_words", None) is not None:
            words = np.ascontiguousarray(cfg["rng_words"], dtype=np.uint64).reshape(B, 6)
        else:
            seeds = cfg.get("seeds", None)
            if seeds is None:
                seed = cfg.get("seed", None)
                seeds = [None] * B if seed is None else [int(seed) + b for b in range(B)]
            words = np.stack([pcg64_words(s) for s in seeds])
        self._check(self._lib.mapf_set_rng_state(h, words.ctypes.data_as(C.c_void_p)))
        dev = self.device
        with torch.cuda.device(dev):
            self._obs = torch.zeros((B, self.obs_len), dtype=torch.float32, device=dev)
            self._final_obs = torch.zeros((B, self.obs_len), dtype=torch.float32, device=dev)
            self._reward = torch.zeros((B,), dtype=torch.float64, device=dev)
            self._terminated = torch.zeros((B,), dtype=torch.uint8, device=dev)
            self._truncated = torch.zeros((B,), dtype=torch.uint8, device=dev)
            self._info = torch.zeros((B, 4), dtype=torch.float32, device=dev)
        if self.deterministic:  # fixed tables (SA-env:109-112)
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
        else:  # the ctor's own generate_starts_goals() draw (SA-env:113-114)
            self._check(self._lib.mapf_cte_reset(h, None, None, self._stream()))

    def _stream(self):
        return C.c_void_p(_raw_stream(int(self.device.index)))

    def _check(self, rc, exc=RuntimeError):
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

    def reset(self, env_mask=None) -> torch.Tensor:
        mptr = None
        if env_mask is not None:
            env_mask = env_mask.to(device=self.device, dtype=torch.uint8).contiguous()
            mptr = C.c_void_p(env_mask.data_ptr())
        self._check(self._lib.mapf_cte_reset(self._h, mptr, C.c_void_p(self._obs.data_ptr()), self._stream()))
        return self._obs

    def step(self, actions: torch.Tensor, auto_reset: bool = True, want_final_obs: bool = False) -> dict:
        """actions: int8 [B, N] (the reference's MultiDiscrete([5]*N) action per env)."""
        if actions.dtype != torch.int8 or actions.device != self.device or not actions.is_contiguous():
            actions = actions.to(device=self.device, dtype=torch.int8).contiguous()
        if tuple(actions.shape) != (self.num_envs, self.num_agents):
            raise ValueError(f"actions must have shape {(self.num_envs, self.num_agents)}")
        # (the output tensors live as long as the handle: their pointers are marshalled once, the per-call cost is the
        #  actions' pointer, the raw handle of the current stream and one ctypes call)
        po = self._out_ptrs
        if po is None:
            po = self._out_ptrs = tuple(C.c_void_p(t.data_ptr()) for t in (self._obs, self._reward, self._terminated, self._truncated,
                                                                          self._info, self._final_obs))
            self._out_plain = {"obs": self._obs, "reward": self._reward, "terminated": self._terminated, "truncated": self._truncated,
                               "info": self._info, "final_obs": None}
            self._out_final = dict(self._out_plain, final_obs=self._final_obs)
            self._act_shape = (self.num_envs, self.num_agents)
        final = want_final_obs and auto_reset
        rc = self._lib.mapf_cte_step(self._h, actions.data_ptr(), po[0], po[1], po[2], po[3], po[4], po[5] if final else None,
                                     1 if auto_reset else 0, _raw_stream(self._dev_index))
        if rc != 0:
            self._check(rc)
        return self._out_final if final else self._out_plain

    def step_many(self, actions: torch.Tensor, obs_mode: int = 1) -> dict:
        """T fused steps in one launch (mapf_cte_step_many).  actions: int8 [T, B, N].  Returns fresh tensors: obs
        ([B, row] for obs_mode 1, [T, B, row] for 2, None for 0), reward [T, B] float64, terminated / truncated [T, B],
        info [T, B, 4].  Finished envs are reset inside the launch (their row of that step is the reset observation).
        Call poll_error() afterwards: an invalid action is latched there, and the rows of that env from that step on are
        zeros, not results."""
        if actions.dtype != torch.int8 or actions.device != self.device or not actions.is_contiguous():
            actions = actions.to(device=self.device, dtype=torch.int8).contiguous()
        T, B = int(actions.shape[0]), self.num_envs
        if tuple(actions.shape) != (T, B, self.num_agents):
            raise ValueError(f"actions must have shape (T, {B}, {self.num_agents})")
        dev = self.device
        obs = None
        if obs_mode == 1:
            obs = torch.empty((B, self.obs_len), dtype=torch.float32, device=dev)
        elif obs_mode == 2:
            obs = torch.empty((T, B, self.obs_len), dtype=torch.float32, device=dev)
        # zero-filled, not empty: the kernel skips the output stores of a step in which an env hit an invalid action
        # (SA-env:401-403 raises there), so those [t, env] rows would otherwise hold uninitialised memory.  Such an env is
        # reported by poll_error(), which a caller of step_many must consult (it synchronises, so it is not done here).
        out = {"obs": obs, "reward": torch.zeros((T, B), dtype=torch.float64, device=dev),
               "terminated": torch.zeros((T, B), dtype=torch.uint8, device=dev),
               "truncated": torch.zeros((T, B), dtype=torch.uint8, device=dev),
               "info": torch.zeros((T, B, 4), dtype=torch.float32, device=dev)}
        self._check(self._lib.mapf_cte_step_many(
            self._h, T, C.c_void_p(actions.data_ptr()), None if obs is None else C.c_void_p(obs.data_ptr()), int(obs_mode),
            C.c_void_p(out["reward"].data_ptr()), C.c_void_p(out["terminated"].data_ptr()),
            C.c_void_p(out["truncated"].data_ptr()), C.c_void_p(out["info"].data_ptr()), self._stream()))
        return out

    def set_step_counts(self, counts) -> None:
        """Put env b `counts[b]` steps into its episode (staggered episode boundaries for benchmarks and tests)."""
        c = self.get_state()["counters"]
        c[:, L.CTR_STEP_COUNT] = np.asarray(counts, dtype=np.int32)
        s = L.MapfState(counters=c.ctypes.data_as(C.c_void_p))
        self._check(self._lib.mapf_set_state(self._h, C.byref(s)), ValueError)

    def launch_info(self, fused: bool = False) -> dict:
        """Launch shape of ``step()`` (``fused=True``: of ``step_many()`` with T > 1, which picks its own group width)."""
        b, t, l, p = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
        fn = self._lib.mapf_cte_many_launch_info if fused else self._lib.mapf_launch_info
        fn(self._h, C.byref(b), C.byref(t), C.byref(l), C.byref(p))
        return {"blocks": b.value, "threads": 128, "lds_bytes": l.value, "lanes_per_env": p.value, "specialized_kernel": 0,
                "jit": False, "jit_note": "single-agent env"}

    def poll_error(self):
        env, agent, value = C.c_int32(-1), C.c_int32(-1), C.c_int32(0)
        rc = self._lib.mapf_poll_error(self._h, self._stream(), C.byref(env), C.byref(agent), C.byref(value))
        if rc == L.MAPF_OK:
            return
        if rc == L.MAPF_ERR_BAD_ACTION:
            raise ValueError("Invalid action")  # SA-env:401-403
        raise RuntimeError(self._lib.mapf_last_error(self._h).decode())

    def get_state(self) -> dict:
        B, N = self.num_envs, self.num_agents
        out = {"positions": np.zeros((B, N, 2), np.int16), "goals": np.zeros((B, N, 2), np.int16),
               "starts": np.zeros((B, N, 2), np.int16), "reached": np.zeros((B, N), np.uint8),
               "counters": np.zeros((B, L.NUM_COUNTERS), np.int32), "rng_words": np.zeros((B, 6), np.uint64)}
        s = L.MapfState(positions=out["positions"].ctypes.data_as(C.c_void_p), goals=out["goals"].ctypes.data_as(C.c_void_p),
                        starts=out["starts"].ctypes.data_as(C.c_void_p), reached=out["reached"].ctypes.data_as(C.c_void_p),
                        counters=out["counters"].ctypes.data_as(C.c_void_p),
                        rng_words=out["rng_words"].ctypes.data_as(C.c_void_p))
        self._check(self._lib.mapf_get_state(self._h, C.byref(s)))
        return out
