"""Named grids and fixed start/goal tables (reference: src/environments/get_grid.py).

The tables are data, shipped as ``data/named_grids.npz`` (exported from the reference's own
accessors by oracle/gen_golden.py); this module mirrors the three accessor functions
(get_grid.py:6, :735, :805) including their error behaviour.
"""

from __future__ import annotations

import os

import numpy as np

_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "named_grids.npz")
_cache = None


def _tables() -> dict:
    global _cache
    if _cache is None:
        with np.load(_DATA, allow_pickle=False) as z:
            _cache = {k: z[k] for k in z.files}
    return _cache


def grid_names() -> list[str]:
    return sorted(k[: -len(".grid")] for k in _tables() if k.endswith(".grid"))


def get_grid(env_name: str) -> np.ndarray:
    """uint8[H,W], 0 = free, 1 = obstacle.  Unknown name -> ValueError (get_grid.py:728-732)."""
    try:
        return _tables()[env_name + ".grid"].astype(np.uint8, copy=True)
    except KeyError as exc:
        raise ValueError(f"Unknown environment name: {env_name}") from exc


def _positions(kind: str, env_name: str, num_agents: int) -> dict:
    key = f"{env_name}.{kind}"
    t = _tables()
    if key not in t:
        raise ValueError(f"Unknown environment name: {env_name}")
    table = t[key]
    if num_agents > len(table):
        what = "positions" if kind == "starts" else "goal positions"
        raise ValueError(f"Requested number of agents ({num_agents}) exceeds available {what} in {env_name}")
    return {f"agent_{i}": (int(table[i][0]), int(table[i][1])) for i in range(num_agents)}


def get_start_positions(env_name: str, num_agents: int) -> dict:
    """get_grid.py:735-802"""
    return _positions("starts", env_name, num_agents)


def get_goal_positions(env_name: str, num_agents: int) -> dict:
    """get_grid.py:805-872"""
    return _positions("goals", env_name, num_agents)
