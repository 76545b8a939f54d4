"""Drop-in ``ReferenceModel``: the reference's RLlib ``MultiAgentEnv`` / gymnasium dict API served
by the HIP step engine (one env per object, i.e. a B=1 engine handle).

Mirrors reference ``src/environments/reference_model_multi_agent.py`` ("MA-env"): same constructor
keys and defaults (:38-61), same ``reset`` / ``step`` signatures and return structure (:440, :474,
:695), same attribute surface the callers read (callbacks.py:111-131,265-307; main.py:154,265,
290,314; benchmark script :39,:47) and the private arrays its tests poke.  All arithmetic of the
step / reset path runs on the GPU; this class only converts between dicts and device tensors.

State arrays (``_positions_arr`` ...) are host mirrors: they are refreshed from the device after
every ``reset``/``step`` and pushed back to the device before the next call when a caller has
modified them (that is how the reference's tests inject states).

``_assign_new_goal(agent_idx)`` is callable by itself (one respawn on the device, the env's stream advances as in
the reference); monkeypatching it does not reach the respawns ``step()`` performs, which run inside the kernel.
Not provided: matplotlib rendering (``render`` is a no-op; GUI is out of scope).

The dict API costs a kernel launch plus small device<->host copies per call, like any per-env
Python env; throughput work should use ``VecReferenceModel`` (tensor API) instead.
"""

from __future__ import annotations

import logging

import numpy as np
import torch

from . import _lib as L
from . import get_grid
from .actions import DOWN, LEFT, NO_OP, RIGHT, UP
from .spaces import Box, Discrete, MultiAgentEnv, MultiBinary
from .vec_env import VecReferenceModel

logger = logging.getLogger(__name__)

_COUNTER_ATTRS = (
    ("step_count", L.CTR_STEP_COUNT, int),
    ("_episode_blocking_count", L.CTR_BLOCKING_COUNT, float),
    ("_episode_goals_reached_total", L.CTR_GOALS_REACHED_TOTAL, float),
    ("_episode_deadlock_events", L.CTR_DEADLOCK_EVENTS, float),
    ("_episode_livelock_events", L.CTR_LIVELOCK_EVENTS, float),
    ("_episode_deadlock_steps", L.CTR_DEADLOCK_STEPS, float),
    ("_episode_livelock_steps", L.CTR_LIVELOCK_STEPS, float),
)


class ReferenceModel(MultiAgentEnv):
    """Multi-agent grid world with flat per-agent observations (layout: local_obs, goal_delta,
    optional goal_distance, optional blocking_pressure_prev, optional action_mask)."""

    EMPTY_CELL = 0
    OBSTACLE_CELL = 1
    OTHER_AGENT_CELL = 2
    OWN_GOAL_CELL = 3
    OTHER_GOAL_CELL = 4
    TRAVERSABLE_LOCAL_VALUES = (EMPTY_CELL, OWN_GOAL_CELL, OTHER_GOAL_CELL)
    UNASSIGNED_OWNER = -1

    def __init__(self, env_config):
        super().__init__()
        cfg = dict(env_config)
        # ---- configuration, same keys / defaults / clamps as MA-env:37-61 --------------------
        self.step_count = 0
        self.steps_per_episode = cfg.get("steps_per_episode", 100)
        self._num_agents = int(cfg.get("num_agents", 2))
        self.sensor_range = cfg.get("sensor_range", 1)
        self.deterministic = cfg.get("deterministic", False)
        self.normalize_goal_delta = cfg.get("normalize_goal_delta", True)
        self.include_goal_distance = cfg.get("include_goal_distance", False)
        self.include_action_mask_in_obs = bool(cfg.get("include_action_mask_in_obs", False))
        self.include_blocking_pressure_in_obs = bool(cfg.get("include_blocking_pressure_in_obs", True))
        self.validate_observation_space = bool(cfg.get("validate_observation_space", False))
        self.possible_agents = [f"agent_{i}" for i in range(self._num_agents)]
        self.agents = self.possible_agents.copy()
        self.render_env = cfg.get("render_env", False)
        self.info_mode = str(cfg.get("info_mode", "lite")).lower()
        self.lifelong_mapf = bool(cfg.get("lifelong_mapf", False))
        self._needs_action_mask = self.include_action_mask_in_obs or self.info_mode == "full"
        if self.info_mode not in {"lite", "full"}:
            msg = f"Unsupported info_mode '{self.info_mode}'. Expected 'lite' or 'full'."
            raise ValueError(msg)
        self.deadlock_window_steps = max(1, int(cfg.get("deadlock_window_steps", 8)))
        self.livelock_window_steps = max(1, int(cfg.get("livelock_window_steps", 16)))
        self.lock_nearby_manhattan = max(1, int(cfg.get("lock_nearby_manhattan", 2)))
        self.lock_progress_epsilon = float(cfg.get("lock_progress_epsilon", 1))
        self.lock_min_neighbors = max(1, int(cfg.get("lock_min_neighbors", 1)))
        self.enable_lock_metrics = bool(cfg.get("enable_lock_metrics", True))
        self.goal_reached_once = dict.fromkeys(self.agents, False)
        self._agent_index = {agent_id: idx for idx, agent_id in enumerate(self.agents)}
        self._coord_dtype = np.int16
        self.seed = cfg.get("seed", None)

        grid = cfg.get("grid", None)
        self.grid = get_grid.get_grid(cfg["env_name"]) if grid is None else np.array(grid, dtype=np.uint8)
        self._free_positions = np.argwhere(self.grid == self.EMPTY_CELL).astype(self._coord_dtype, copy=False)
        self._action_deltas = np.array([[0, 0], [-1, 0], [0, 1], [1, 0], [0, -1]], dtype=self._coord_dtype)

        # ---- engine (B = 1) -------------------------------------------------------------------
        ecfg = dict(cfg)
        ecfg["num_envs"] = 1
        ecfg["grid"] = self.grid
        self._engine = VecReferenceModel(ecfg)
        self._lock_history_size = max(self.deadlock_window_steps, self.livelock_window_steps)

        # ---- host mirrors of the device state ----------------------------------------------------
        n = self._num_agents
        self._starts_arr = np.zeros((n, 2), dtype=self._coord_dtype)
        self._positions_arr = np.zeros((n, 2), dtype=self._coord_dtype)
        self._goals_arr = np.zeros((n, 2), dtype=self._coord_dtype)
        self._reached_arr = np.zeros(n, dtype=np.bool_)
        self._completed_once_arr = np.zeros(n, dtype=np.bool_)
        self._blocking_pressure_prev_arr = np.zeros(n, dtype=np.float32)
        self._deadlock_state_prev = False
        self._livelock_state_prev = False
        self._shadow = {}
        self._bind_public_state_views()
        self._pull()

        # ---- spaces (MA-env:136-184) --------------------------------------------------------------
        self._view_side = self.sensor_range * 2 + 1
        self._local_obs_space = Box(low=0, high=self.OTHER_GOAL_CELL, shape=(self._view_side, self._view_side),
                                    dtype=np.uint8)
        high = np.array([self.grid.shape[0] - 1, self.grid.shape[1] - 1], dtype=np.float32)
        self._goal_delta_denominator = np.array(
            [max(self.grid.shape[0] - 1, 1), max(self.grid.shape[1] - 1, 1)], dtype=np.float32)
        low = -high
        if self.normalize_goal_delta:
            low = low / self._goal_delta_denominator
            high = high / self._goal_delta_denominator
        self._goal_delta_space = Box(low=np.asarray(low, np.float32), high=np.asarray(high, np.float32), shape=(2,),
                                     dtype=np.float32)
        self._single_act_space = Discrete(5)
        self._action_mask_space = MultiBinary(int(self._single_act_space.n))
        self._blocking_pressure_space = Box(low=np.zeros(1, np.float32), high=np.ones(1, np.float32), dtype=np.float32)
        self._single_obs_space, self._obs_slices = self._build_obs_layout()
        self._single_obs_len = int(np.prod(self._single_obs_space.shape))
        assert self._single_obs_len == self._engine.obs_len
        self.observation_spaces = dict.fromkeys(self.possible_agents, self._single_obs_space)
        self.action_spaces = dict.fromkeys(self.possible_agents, self._single_act_space)
        self.observation_space = self._single_obs_space
        self.action_space = self._single_act_space

    # ---------------------------------------------------------------------------------------------
    def _bind_public_state_views(self):
        """dict views of the state arrays, as the reference exposes them (MA-env:194-198)."""
        self.starts = {aid: self._starts_arr[i] for aid, i in self._agent_index.items()}
        self.positions = {aid: self._positions_arr[i] for aid, i in self._agent_index.items()}
        self.goals = {aid: self._goals_arr[i] for aid, i in self._agent_index.items()}

    def _build_obs_component_spaces(self):
        comps = [("local_obs", self._local_obs_space), ("goal_delta", self._goal_delta_space)]
        if self.include_goal_distance:
            max_manhattan = float(np.abs(self._goal_delta_space.high).sum())
            comps.append(("goal_distance", Box(low=np.zeros(1, np.float32),
                                               high=np.asarray([max_manhattan], np.float32), dtype=np.float32)))
        if self.include_blocking_pressure_in_obs:
            comps.append(("blocking_pressure_prev", self._blocking_pressure_space))
        if self.include_action_mask_in_obs:
            comps.append(("action_mask", self._action_mask_space))
        return comps

    def _build_obs_layout(self):
        """Flat observation space and the slice of each component in it (the order of MA-env:306-328)."""
        names, lows, highs = [], [], []
        for name, space in self._build_obs_component_spaces():
            size = int(np.prod(space.shape))
            binary = isinstance(space, MultiBinary)
            names.append((name, size))
            lows.append(np.zeros(size, np.float32) if binary else np.asarray(space.low, np.float32).ravel())
            highs.append(np.ones(size, np.float32) if binary else np.asarray(space.high, np.float32).ravel())
        ends = np.cumsum([size for _, size in names])
        slices = {name: slice(int(end - size), int(end)) for (name, size), end in zip(names, ends)}
        return Box(low=np.concatenate(lows), high=np.concatenate(highs), dtype=np.float32), slices

    # ---- device <-> host mirror -------------------------------------------------------------------
    def _pull(self):
        s = self._engine.get_state()
        np.copyto(self._positions_arr, s["positions"][0])
        np.copyto(self._goals_arr, s["goals"][0])
        np.copyto(self._starts_arr, s["starts"][0])
        np.copyto(self._reached_arr, s["reached"][0].astype(np.bool_))
        np.copyto(self._completed_once_arr, s["completed_once"][0].astype(np.bool_))
        np.copyto(self._blocking_pressure_prev_arr, s["pressure_prev"][0].astype(np.float32))
        ctr = s["counters"][0]
        for name, idx, typ in _COUNTER_ATTRS:
            setattr(self, name, typ(ctr[idx]))
        self._deadlock_state_prev = bool(ctr[L.CTR_LOCK_STATE_PREV] & 1)
        self._livelock_state_prev = bool(ctr[L.CTR_LOCK_STATE_PREV] & 2)
        self._lock_hist_count = int(min(ctr[L.CTR_HIST_ROWS], self._lock_history_size))
        self._lock_hist_head = int(ctr[L.CTR_HIST_ROWS] % self._lock_history_size)
        self._counters_raw = ctr.copy()
        self.goal_reached_once = {aid: bool(self._completed_once_arr[i]) for aid, i in self._agent_index.items()}
        self._shadow = self._snapshot()

    def _snapshot(self):
        return {
            "positions": self._positions_arr.copy(), "goals": self._goals_arr.copy(), "starts": self._starts_arr.copy(),
            "reached": self._reached_arr.copy(), "completed": self._completed_once_arr.copy(),
            "pressure": self._blocking_pressure_prev_arr.copy(),
            "counters": tuple(getattr(self, name) for name, _, _ in _COUNTER_ATTRS)
            + (self._deadlock_state_prev, self._livelock_state_prev),
        }

    def _push_if_dirty(self):
        """Upload whatever a caller changed in the host mirrors since the last pull."""
        now, old = self._snapshot(), self._shadow
        kw = {}
        if not np.array_equal(now["positions"], old["positions"]):
            kw["positions"] = now["positions"][None]
        if not np.array_equal(now["goals"], old["goals"]):
            kw["goals"] = now["goals"][None]
        if not np.array_equal(now["starts"], old["starts"]):
            kw["starts"] = now["starts"][None]
        if not np.array_equal(now["reached"], old["reached"]):
            kw["reached"] = now["reached"].astype(np.uint8)[None]
        if not np.array_equal(now["completed"], old["completed"]):
            kw["completed_once"] = now["completed"].astype(np.uint8)[None]
        if not np.array_equal(now["pressure"], old["pressure"]):
            kw["pressure_prev"] = (now["pressure"] != 0).astype(np.uint8)[None]
        if now["counters"] != old["counters"]:
            ctr = self._counters_raw.copy()
            for name, idx, _ in _COUNTER_ATTRS:
                ctr[idx] = int(getattr(self, name))
            ctr[L.CTR_LOCK_STATE_PREV] = int(bool(self._deadlock_state_prev)) | (int(bool(self._livelock_state_prev)) << 1)
            kw["counters"] = ctr[None]
        if getattr(self, "_pending_lock_reset", False):
            ctr = kw.get("counters", self._counters_raw.copy()[None])[0]
            for idx in (L.CTR_HIST_ROWS, L.CTR_DEADLOCK_EVENTS, L.CTR_LIVELOCK_EVENTS, L.CTR_DEADLOCK_STEPS,
                        L.CTR_LIVELOCK_STEPS, L.CTR_LOCK_STATE_PREV):
                ctr[idx] = 0
            kw["counters"] = ctr[None]
            kw["lock_history"] = np.zeros((1, self._num_agents, 3), np.uint64)
            self._pending_lock_reset = False
        if kw:
            self._engine.set_state(**kw)
            self._shadow = now

    # ---- helpers the reference's tests call ------------------------------------------------------------
    def _rebuild_occupancy_owner(self):
        """Owner maps are derived data here (the kernel needs none); kept as a sync point for callers."""
        self._push_if_dirty()

    def _rebuild_goal_owner(self):
        self._push_if_dirty()

    @property
    def _occupancy_owner(self):
        m = np.full(self.grid.shape, self.UNASSIGNED_OWNER, dtype=np.int16)
        for i in range(self._num_agents):
            m[self._positions_arr[i, 0], self._positions_arr[i, 1]] = i
        return m

    @property
    def _goal_owner(self):
        m = np.full(self.grid.shape, self.UNASSIGNED_OWNER, dtype=np.int16)
        for i in range(self._num_agents):
            m[self._goals_arr[i, 0], self._goals_arr[i, 1]] = i
        return m

    def _reset_lock_tracking(self):
        """MA-env:360-372: clear history + lock counters (applied on the device before the next call)."""
        self._episode_deadlock_events = 0.0
        self._episode_livelock_events = 0.0
        self._episode_deadlock_steps = 0.0
        self._episode_livelock_steps = 0.0
        self._deadlock_state_prev = False
        self._livelock_state_prev = False
        self._pending_lock_reset = True

    def generate_starts_goals(self):
        """MA-env:267-282: draw new starts/goals from the env's RNG stream (device-side PCG64)."""
        self._push_if_dirty()
        keep = {name: getattr(self, name) for name, _, _ in _COUNTER_ATTRS}
        flags = (self._reached_arr.copy(), self._completed_once_arr.copy(), self._blocking_pressure_prev_arr.copy())
        self._engine.reset()  # same draw; then restore the episode bookkeeping reset() would have cleared
        self._pull()
        for name, v in keep.items():
            setattr(self, name, v)
        np.copyto(self._reached_arr, flags[0])
        np.copyto(self._completed_once_arr, flags[1])
        np.copyto(self._blocking_pressure_prev_arr, flags[2])
        self._push_if_dirty()

    def _assign_new_goal(self, agent_idx: int) -> np.ndarray:
        """MA-env:284-304: assign a new unique, currently unoccupied goal to one agent (device-side: the candidates in
        row-major order, ``rng.integers(k)`` on the env's PCG64 stream).  Returns the new goal like the reference."""
        self._push_if_dirty()
        new_goal = self._engine.assign_new_goal(0, int(agent_idx)).astype(self._coord_dtype, copy=False)
        self._pull()
        return new_goal

    def get_agent_ids(self):
        return set(self.agents)

    def get_next_position(self, action: int, pos):
        """MA-env:697-705"""
        action = int(action)
        if action < NO_OP or action > LEFT:
            msg = "Invalid action"
            raise ValueError(msg)
        return np.asarray(pos, dtype=self._coord_dtype) + self._action_deltas[action]

    def _observe_all(self) -> np.ndarray:
        self._push_if_dirty()
        return self._engine.observe().cpu().numpy()[0]

    def get_obs(self, agent_id: str):
        """Local V x V observation of one agent from the current state (MA-env:707-747), computed on device."""
        row = self._observe_all()[self._agent_index[agent_id]]
        return row[self._obs_slices["local_obs"]].astype(np.uint8).reshape(self._view_side, self._view_side)

    def get_action_mask(self, obs):
        """[no-op, up, right, down, left] from a local observation: a move is allowed iff the neighbouring window cell
        exists and holds a traversable code (MA-env:749-773).  (step() / reset() take the mask the kernel computed.)"""
        obs = np.asarray(obs)
        c = self.sensor_range
        open_cell = np.isin(obs, self.TRAVERSABLE_LOCAL_VALUES)
        mask = np.zeros(self._action_mask_space.shape, dtype=self._action_mask_space.dtype)
        mask[NO_OP] = 1
        for action, (dr, dc) in ((UP, (-1, 0)), (RIGHT, (0, 1)), (DOWN, (1, 0)), (LEFT, (0, -1))):
            r, q = c + dr, c + dc
            if 0 <= r < obs.shape[0] and 0 <= q < obs.shape[1]:
                mask[action] = int(open_cell[r, q])
        return mask

    def _get_goal_delta(self, agent_id: str) -> np.ndarray:
        i = self._agent_index[agent_id]
        gd = (self._goals_arr[i] - self._positions_arr[i]).astype(np.float32)
        if self.normalize_goal_delta:
            gd = np.asarray(gd / self._goal_delta_denominator, dtype=np.float32)
        return gd

    def _full_info(self, idx: int, obs_row: np.ndarray) -> dict:
        """info_mode='full' payload (MA-env:350-358), sliced out of the device observation."""
        sl = self._obs_slices
        local = obs_row[sl["local_obs"]].astype(np.uint8).reshape(self._view_side, self._view_side)
        mask = (obs_row[sl["action_mask"]].astype(self._action_mask_space.dtype) if "action_mask" in sl
                else self.get_action_mask(local))
        return {
            "position": np.asarray(self._positions_arr[idx]),
            "goal": np.asarray(self._goals_arr[idx]),
            "goal_delta": np.asarray(obs_row[sl["goal_delta"]], dtype=np.float32),
            "action_mask": mask,
            "local_obs": local,
        }

    def _check_obs(self, agent_id: str, obs: np.ndarray, where: str) -> np.ndarray:
        obs = np.asarray(obs, dtype=np.float32)
        if self.validate_observation_space and not self.observation_space.contains(obs):
            msg = (f"{where} produced invalid observation for {agent_id} "
                   f"(dtype={obs.dtype}, min={float(np.min(obs))}, max={float(np.max(obs))}).")
            raise ValueError(msg)
        return obs

    # ---- gymnasium / RLlib API -------------------------------------------------------------------
    def reset(self, *, seed=None, options=None):
        """MA-env:440-472 (``seed`` and ``options`` are ignored, exactly like the reference)."""
        self._push_if_dirty()
        obs_t = self._engine.reset()
        obs_np = obs_t.cpu().numpy()[0]
        self._pull()
        obs, infos = {}, {aid: {} for aid in self.agents}
        for i, aid in enumerate(self.agents):
            obs[aid] = self._check_obs(aid, obs_np[i].copy(), "reset")
            if self.info_mode == "full":
                infos[aid] = self._full_info(i, obs_np[i])
        return obs, infos

    def step(self, action_dict):
        """MA-env:474-695"""
        if not action_dict or any(aid not in action_dict for aid in self.agents):
            action_dict = dict.fromkeys(self.agents, NO_OP)
            logger.warning("No actions provided or missing agent actions. Defaulting to no-op actions: %s", action_dict)
        acts = np.zeros((1, self._num_agents), dtype=np.int8)
        first_bad = None
        for i, aid in enumerate(self.agents):
            a = int(action_dict[aid])
            if a < NO_OP or a > LEFT:
                if first_bad is None:
                    first_bad = (a, aid)
                a = 5  # any out-of-range code: the kernel stops the agent loop there, like the reference
            acts[0, i] = a
        self._push_if_dirty()
        out = self._engine.step(torch.from_numpy(acts).to(self._engine.device), auto_reset=False)
        if first_bad is not None:
            try:
                self._engine.poll_error()
            except ValueError:
                pass
            self._pull()
            msg = f"Invalid action {first_bad[0]} for {first_bad[1]}"
            raise ValueError(msg)
        self._engine.poll_error()
        obs_np = out["obs"].cpu().numpy()[0]
        rew = out["rewards"].cpu().numpy()[0]
        term, trunc = bool(out["terminated"].cpu().numpy()[0]), bool(out["truncated"].cpu().numpy()[0])
        ia = out["info_all"].cpu().numpy()[0]
        iag = out["info_agent"].cpu().numpy()[0]
        self._pull()

        obs, rewards, info = {}, {}, {aid: {} for aid in self.agents}
        if self.lifelong_mapf:
            goals_reached_total = float(self._episode_goals_reached_total)
        else:
            goals_reached_total = float(np.sum(self._reached_arr))
        blocking_count_total = float(self._episode_blocking_count)
        for i, aid in enumerate(self.agents):
            obs[aid] = self._check_obs(aid, obs_np[i].copy(), "step")
            rewards[aid] = float(rew[i])
            if self.info_mode == "full":
                info[aid] = self._full_info(i, obs_np[i])
            info[aid]["blocking"] = float(iag[i, 0])
            info[aid]["goal_reached_step"] = float(iag[i, 1])
        for aid in self.agents:
            info[aid]["goals_reached_total"] = goals_reached_total
            info[aid]["blocking_count_total"] = blocking_count_total
        info_all = {
            "goals_reached_step": float(ia[0]),
            "goals_reached_total": goals_reached_total,
            "blocking_count_step": float(ia[2]),
            "blocking_count_total": blocking_count_total,
            "deadlock_step": float(ia[4]),
            "livelock_step": float(ia[5]),
            "deadlock_event_step": float(ia[6]),
            "livelock_event_step": float(ia[7]),
            "deadlock_events_total": float(self._episode_deadlock_events),
            "livelock_events_total": float(self._episode_livelock_events),
            "deadlock_steps_total": float(self._episode_deadlock_steps),
            "livelock_steps_total": float(self._episode_livelock_steps),
        }
        if self.lifelong_mapf:  # float64 like the reference (MA-env:638, :653-655)
            info_all["completion_ratio"] = float(np.mean(self._completed_once_arr))
            info_all["throughput"] = goals_reached_total / float(max(self.step_count, 1))
        info["__all__"] = info_all

        terminated = dict.fromkeys(self.agents, term)
        truncated = dict.fromkeys(self.agents, trunc)
        terminated["__all__"] = term
        truncated["__all__"] = trunc
        return obs, rewards, terminated, truncated, info

    def render(self, mode="human"):
        """Rendering (matplotlib GUI, MA-env:775-916) is outside this engine's scope."""
        return None

    def close(self):
        self._engine.close()
