"""Drop-in for the reference's single-agent (CTE) env, ``src/environments/reference_model_single_agent.py``
("SA-env"): a ``gym.Env`` whose one policy drives all agents, served by the HIP engine (a B = 1 handle).

Same constructor keys / defaults (SA-env:84-114), ``reset() -> (obs, {"action_mask"})`` (:222-244),
``step(action) -> (obs, reward, terminated, truncated, info)`` (:246-363) with the reference's info keys,
``split_flat_observation`` (:216-220), ``get_next_position`` (:366-405), position / goal dict views.
"""

from __future__ import annotations

import numpy as np
import torch

from . import get_grid
from .spaces import Box, GymEnv, MultiBinary, MultiDiscrete
from .vec_env_single_agent import VecSingleAgentReferenceModel


class ReferenceModel(GymEnv):
    def __init__(self, env_config):
        super().__init__()
        cfg = dict(env_config)
        self.step_count = 0
        self.steps_per_episode = cfg.get("steps_per_episode", 100)
        self.num_agents = cfg.get("num_agents", 2)
        self.sensor_range = cfg.get("sensor_range", 1)
        self.deterministic = cfg.get("deterministic", False)
        self._agent_ids = {f"agent_{i}" for i in range(self.num_agents)}
        self.render_env = cfg.get("render_env", False)
        self.goal_reached_once = {f"agent_{i}": False for i in range(self.num_agents)}
        self.blocking_penalty = cfg.get("blocking_penalty", -0.2)
        self.move_after_goal_penalty = cfg.get("move_after_goal_penalty", -0.05)
        self._episode_blocking_count = 0.0
        self.validate_observation_space = bool(cfg.get("validate_observation_space", False))
        self.seed = cfg.get("seed", None)
        grid = cfg.get("grid", None)
        self.grid = get_grid.get_grid(cfg["env_name"]) if grid is None else np.array(grid, dtype=np.uint8)
        ecfg = dict(cfg, num_envs=1, grid=self.grid)
        self._engine = VecSingleAgentReferenceModel(ecfg)
        n = self.num_agents
        self._grid_obs_space = Box(low=0, high=2 * n + 1, shape=self.grid.shape, dtype=np.uint8)
        self._action_mask_space = MultiBinary(5 * n)
        flat_grid_len, flat_mask_len = int(self.grid.size), 5 * n
        low = np.zeros(flat_grid_len + flat_mask_len, dtype=np.float32)
        high = np.concatenate([np.full(flat_grid_len, 2 * n + 1, dtype=np.float32), np.ones(flat_mask_len, np.float32)])
        self._obs_slices = {"grid": slice(0, flat_grid_len),
                            "action_mask": slice(flat_grid_len, flat_grid_len + flat_mask_len)}
        self.observation_space = Box(low=low, high=high, dtype=np.float32)
        self.action_space = MultiDiscrete([5] * n)
        self._pull()

    def _pull(self):
        s = self._engine.get_state()
        ids = [f"agent_{i}" for i in range(self.num_agents)]
        self.starts = {a: s["starts"][0, i].astype(np.int64) for i, a in enumerate(ids)}
        self.positions = {a: s["positions"][0, i].astype(np.int64) for i, a in enumerate(ids)}
        self.goals = {a: s["goals"][0, i].astype(np.int64) for i, a in enumerate(ids)}
        self.goal_reached_once = {a: bool(s["reached"][0, i]) for i, a in enumerate(ids)}
        self.step_count = int(s["counters"][0, 0])
        self._episode_blocking_count = float(s["counters"][0, 2])

    def _check_obs(self, obs, where):
        obs = np.asarray(obs, dtype=np.float32)
        if self.validate_observation_space and not self.observation_space.contains(obs):
            raise ValueError(f"{where} produced observation outside observation_space "
                             f"(dtype={obs.dtype}, min={float(np.min(obs))}, max={float(np.max(obs))}).")
        return obs

    def split_flat_observation(self, flat_obs: np.ndarray):
        grid = flat_obs[self._obs_slices["grid"]].reshape(self._grid_obs_space.shape)
        return {"observations": grid, "action_mask": flat_obs[self._obs_slices["action_mask"]]}

    def reset(self, *, seed=None, options=None):
        obs = self._engine.reset().cpu().numpy()[0].copy()
        self._pull()
        mask = obs[self._obs_slices["action_mask"]].astype(self._action_mask_space.dtype)
        return self._check_obs(obs, "reset"), {"action_mask": mask}

    def step(self, action):
        acts = np.zeros((1, self.num_agents), dtype=np.int8)
        bad = False
        for i in range(self.num_agents):
            a = int(action[i])
            if a < 0 or a > 4:
                bad = True
                a = 5
            acts[0, i] = a
        out = self._engine.step(torch.from_numpy(acts).to(self._engine.device), auto_reset=False)
        if bad:
            try:
                self._engine.poll_error()
            except ValueError:
                pass
            self._pull()
            raise ValueError("Invalid action")
        self._engine.poll_error()
        obs = out["obs"].cpu().numpy()[0].copy()
        reward = float(out["reward"].cpu().numpy()[0])
        terminated, truncated = bool(out["terminated"].cpu().numpy()[0]), bool(out["truncated"].cpu().numpy()[0])
        inf = out["info"].cpu().numpy()[0]
        self._pull()
        info = {
            "action_mask": obs[self._obs_slices["action_mask"]].astype(self._action_mask_space.dtype),
            "blocking_count_step": float(inf[0]),
            "goals_reached_step": float(inf[1]),
            "goals_reached_total": float(inf[2]),
            "blocking_count_total": float(inf[3]),
        }
        return self._check_obs(obs, "step"), reward, terminated, truncated, info

    def get_next_position(self, action, pos):
        pos = np.asarray(pos, dtype=np.int32)
        deltas = {0: (0, 0), 1: (-1, 0), 2: (0, 1), 3: (1, 0), 4: (0, -1)}
        if action not in deltas:
            raise ValueError("Invalid action")
        return np.array([pos[0] + deltas[action][0], pos[1] + deltas[action][1]], dtype=np.int32)

    def render(self):
        return None

    def close(self):
        self._engine.close()
