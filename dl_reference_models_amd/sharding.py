"""One process per GPU: env instances shard embarrassingly, no collective on the hot path.

Rank r of G owns the contiguous global env range [r*B/G, (r+1)*B/G) (strong scaling) or
[r*B_per, (r+1)*B_per) (weak scaling).  The only communication is an optional, off-the-timed-path
all-reduce of a small episode-statistics vector (RCCL on GPUs, gloo in CPU tests).
"""

from __future__ import annotations

import os

import numpy as np

EPISODE_STAT_KEYS = (
    "env_steps", "episodes", "goals_reached", "blocking_count", "deadlock_events", "livelock_events",
    "deadlock_steps", "livelock_steps",
)


def dist_env():
    """(rank, local_rank, world_size) from the torchrun environment (defaults: single process)."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def shard_range(total_envs: int, world_size: int, rank: int) -> range:
    """Contiguous split of `total_envs` over ranks; the first (total % world) ranks get one extra."""
    base, extra = divmod(int(total_envs), int(world_size))
    start = rank * base + min(rank, extra)
    return range(start, start + base + (1 if rank < extra else 0))


def weak_range(envs_per_rank: int, rank: int) -> range:
    return range(rank * envs_per_rank, (rank + 1) * envs_per_rank)


class EpisodeStats:
    """Accumulates per-step info_all rows ([B,14], columns of include/mapf_step.h MAPF_INFO_*) into the
    totals the reference's RLlib callbacks report per episode (callbacks.py:135-345)."""

    def __init__(self):
        self.v = np.zeros(len(EPISODE_STAT_KEYS), dtype=np.float64)

    def update(self, info_all: np.ndarray, done: np.ndarray):
        self.v[0] += info_all.shape[0]
        self.v[1] += float(np.count_nonzero(done))
        self.v[2] += float(info_all[:, 0].sum())
        self.v[3] += float(info_all[:, 2].sum())
        self.v[4] += float(info_all[:, 6].sum())
        self.v[5] += float(info_all[:, 7].sum())
        self.v[6] += float(info_all[:, 4].sum())
        self.v[7] += float(info_all[:, 5].sum())

    def as_dict(self) -> dict:
        return dict(zip(EPISODE_STAT_KEYS, (float(x) for x in self.v)))


def all_reduce_stats(vec: np.ndarray, device=None) -> np.ndarray:
    """Sum a small stats vector over all ranks (one fused buffer; latency-bound, so a single collective).
    No-op when torch.distributed is not initialised."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()):
        return np.asarray(vec, dtype=np.float64)
    t = torch.as_tensor(np.asarray(vec, dtype=np.float64), device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.cpu().numpy()
