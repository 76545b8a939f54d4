"""Synthetic workloads of BASELINE.json / SURVEY 8(d): build-defined random-obstacle grids (the
reference ships only 8 named grids) and the env flags the headline metric is quoted on.

Every quantity is a function of the GLOBAL env index, so a shard of envs is the same whichever
rank / GPU count runs it.
"""

from __future__ import annotations

import numpy as np

# common flags: sensor_range 2, mask + blocking pressure in obs (L = 33), lock metrics on
COMMON = {
    "env_name": "synthetic", "sensor_range": 2, "steps_per_episode": 100, "deterministic": False,
    "normalize_goal_delta": True, "include_goal_distance": False, "include_blocking_pressure_in_obs": True,
    "include_action_mask_in_obs": True, "enable_lock_metrics": True, "info_mode": "lite",
}

WORKLOADS = {
    # name: (envs per GPU, H, W, agents, obstacle density, config overrides)
    "c1_10x10_n2": (1, 10, 10, 2, 0.10, {"sensor_range": 1}),
    "c2_1024x16x16_n4": (1024, 16, 16, 4, 0.20, {}),
    "c3_8192x32x32_n8": (8192, 32, 32, 8, 0.40, {}),
    "c5_1024x64x64_n64_lifelong": (1024, 64, 64, 64, 0.20, {"lifelong_mapf": True, "steps_per_episode": 256}),
    # not a BASELINE.json config: the env settings of the reference's own training run (/root/reference/main.py:55-67:
    # 16 agents, sensor_range 3, no action mask in the observation) on synthetic 32x32 grids, for the secondary numbers
    "ref_training_4096x32x32_n16": (4096, 32, 32, 16, 0.20, {"sensor_range": 3, "include_action_mask_in_obs": False,
                                                              "steps_per_episode": 256}),
    # the single-agent (CTE) sibling env: full-grid observation, one policy moves all agents
    "cte_8192x16x16_n4": (8192, 16, 16, 4, 0.20, {"single_agent": True}),
    "cte_1024x32x32_n8": (1024, 32, 32, 8, 0.20, {"single_agent": True}),
}
HEADLINE = "c3_8192x32x32_n8"  # BASELINE.json metric: agent-steps/sec at 8192 envs x 8 agents on 32x32


def synthetic_grid(env_id: int, h: int, w: int, density: float, n_agents: int) -> np.ndarray:
    """grid = default_rng(10_000 + env_id).random((h, w)) < density; reseed (+100_000) while free cells < 2N."""
    s = 10_000 + int(env_id)
    while True:
        g = (np.random.default_rng(s).random((h, w)) < density).astype(np.uint8)
        if int((g == 0).sum()) >= 2 * n_agents:
            return g
        s += 100_000


def synthetic_grids(env_ids, h, w, density, n_agents) -> np.ndarray:
    return np.stack([synthetic_grid(i, h, w, density, n_agents) for i in env_ids])


def workload_config(name: str, env_ids) -> dict:
    """env_config for VecReferenceModel: grids + one NumPy seed per env (= its global env index)."""
    _b, h, w, n, density, over = WORKLOADS[name]
    cfg = dict(COMMON)
    cfg.update(over)
    cfg["num_agents"] = n
    cfg["num_envs"] = len(env_ids)
    cfg["grid"] = synthetic_grids(env_ids, h, w, density, n)
    cfg["seeds"] = [int(i) for i in env_ids]
    return cfg


def is_single_agent(name: str) -> bool:
    return bool(WORKLOADS[name][5].get("single_agent", False))


def obs_len(cfg: dict) -> int:
    v = 2 * int(cfg.get("sensor_range", 1)) + 1
    return (v * v + 2 + (1 if cfg.get("include_goal_distance", False) else 0)
            + (1 if cfg.get("include_blocking_pressure_in_obs", True) else 0)
            + (5 if cfg.get("include_action_mask_in_obs", False) else 0))


def algorithmic_bytes_per_env_step(n_agents: int, obs_floats: int, h: int, w: int) -> int:
    """N*(41 + 4L) + H*W + 122 bytes per env-step."""
    return n_agents * (41 + 4 * obs_floats) + h * w + 122


def cte_algorithmic_bytes_per_env_step(n_agents: int, h: int, w: int) -> int:
    """The single-agent env's counterpart of the figure above (SURVEY 8(d) has none for it; stated in DESIGN.md): the
    observation row written (H*W + 5N floats), the 8-byte hot plane of every agent and the 64-byte env scalars read and
    written, obstacle rows (8 B per grid row) and the action bytes read, reward (f64), two done flags and four info
    floats written."""
    return 4 * (h * w + 5 * n_agents) + 2 * 8 * n_agents + 2 * 64 + 8 * h + n_agents + 8 + 2 + 16
