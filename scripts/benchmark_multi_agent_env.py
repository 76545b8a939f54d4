#!/usr/bin/env python3
"""Benchmark throughput of the multi-agent grid environment (HIP engine).

Counterpart of the reference's ``scripts/benchmark_multi_agent_env.py``: same CLI flags and
defaults (:138-158), same loop structure -- warm-up with reset-on-done, timed loop counting
finished episodes (:59-107) -- same result fields and JSON + CSV writers (:110-135), same
``--assert-min-steps-per-s`` check.  Extensions: ``--num-envs`` / ``--device`` run B vectorised envs
through the tensor API (one kernel launch per step, auto-reset in-kernel, actions sampled on the
host exactly like the reference for ``--num-envs 1``), and agent-steps/s is reported next to
env-steps/s.

    python scripts/benchmark_multi_agent_env.py --env-name ReferenceModel-2-1 --num-agents 2 \
        --deterministic --modes random
"""

from __future__ import annotations

import argparse
import csv
import json
import sys
import time
from datetime import datetime, timezone
from pathlib import Path

import numpy as np

PROJECT_ROOT = Path(__file__).resolve().parents[1]
if str(PROJECT_ROOT) not in sys.path:
    sys.path.insert(0, str(PROJECT_ROOT))

NO_OP = 0


def _build_env_config(args: argparse.Namespace) -> dict:
    cfg = {
        "env_name": args.env_name,
        "seed": args.env_seed,
        "deterministic": args.deterministic,
        "num_agents": args.num_agents,
        "steps_per_episode": args.steps_per_episode,
        "sensor_range": args.sensor_range,
        "info_mode": args.info_mode,
        "training_execution_mode": "CTDE",
        "render_env": False,
    }
    if "masked" in args.modes:
        # the reference script forgets this key and crashes in masked mode on its own snapshot
        # (KeyError 'action_mask', scripts/benchmark_multi_agent_env.py:47); masked sampling needs the mask
        cfg["include_action_mask_in_obs"] = True
    return cfg


class _DictPolicy:
    """Action source of the single-env leg.  Draws from one NumPy generator in the reference script's order (one draw per
    agent, agents in env order: scripts/benchmark_multi_agent_env.py:36-57), so a seeded run chooses the same actions."""

    def __init__(self, env, mode: str, seed: int):
        if mode not in ("random", "masked"):
            raise ValueError(f"Unsupported mode: {mode}")
        self.agents, self.n_actions = list(env.agents), int(env.action_space.n)
        self.mask = env._obs_slices["action_mask"] if mode == "masked" else None
        self.rng = np.random.default_rng(seed)

    def __call__(self, obs: dict) -> dict:
        if self.mask is None:
            return {a: int(self.rng.integers(0, self.n_actions)) for a in self.agents}
        picks = {}
        for a in self.agents:
            allowed = np.flatnonzero(obs[a][self.mask] > 0.5)
            picks[a] = int(self.rng.choice(allowed)) if allowed.size else NO_OP
        return picks


def _count_episodes(advance, steps: int) -> int:
    """Calls advance() `steps` times; advance returns how many episodes ended in that step."""
    ended = 0
    for _ in range(steps):
        ended += advance()
    return ended


def run_benchmark(env_config: dict, mode: str, steps: int, warmup_steps: int, action_seed: int) -> dict:
    """Single env through the drop-in dict API (the reference's run_benchmark: warm-up with reset-on-done, then the
    timed loop counting finished episodes)."""
    from dl_reference_models_amd.reference_model_multi_agent import ReferenceModel

    env = ReferenceModel(env_config)
    policy = _DictPolicy(env, mode, action_seed)
    state = {"obs": env.reset()[0]}

    def advance() -> int:
        obs, _rew, terminated, truncated, _info = env.step(policy(state["obs"]))
        over = bool(terminated.get("__all__", False) or truncated.get("__all__", False))
        state["obs"] = env.reset()[0] if over else obs
        return int(over)

    _count_episodes(advance, warmup_steps)
    t0 = time.perf_counter()
    episodes = _count_episodes(advance, steps)
    elapsed_s = time.perf_counter() - t0
    return _result(mode, steps, warmup_steps, episodes, elapsed_s, env_config, 1, env_config["num_agents"],
                   {"final_positions": env._positions_arr.tolist()})


def run_benchmark_fused(env_config: dict, steps: int, warmup_steps: int, action_seed: int, num_envs: int, device: str,
                        fused: int) -> dict:
    """Masked mode, B envs, `fused` steps per launch: the masked-random policy runs inside the kernel
    (VecReferenceModel.step_many_sampled), so the loop has no per-step host or torch work at all."""
    import torch

    from dl_reference_models_amd.vec_env import VecReferenceModel

    env = VecReferenceModel(dict(env_config, num_envs=num_envs, device=device))
    obs = env.reset()

    def launches(count: int, seed0: int) -> int:
        nonlocal obs
        done = torch.zeros((), dtype=torch.int64, device=env.device)
        for i in range(count):
            out = env.step_many_sampled(fused, seed=seed0 + i, obs_in=obs)
            obs = out["obs"][-1]
            done += ((out["terminated"] | out["truncated"]) != 0).sum()
        torch.cuda.synchronize(env.device)
        return int(done.item())

    launches(max(warmup_steps // fused, 1), action_seed)
    n_launch = max(steps // fused, 1)
    t0 = time.perf_counter()
    episodes = launches(n_launch, action_seed + 1_000_000)
    elapsed_s = time.perf_counter() - t0
    env.poll_error()
    return _result("masked", n_launch * fused, warmup_steps, episodes, elapsed_s, env_config, num_envs, env.num_agents,
                   {"fused_steps_per_launch": fused})


def run_benchmark_vectorized(env_config: dict, mode: str, steps: int, warmup_steps: int, action_seed: int,
                             num_envs: int, device: str) -> dict:
    """B envs through the tensor API; actions are sampled on the device (inputs, not part of the path)."""
    import torch

    from dl_reference_models_amd.vec_env import VecReferenceModel

    cfg = dict(env_config, num_envs=num_envs, device=device)
    env = VecReferenceModel(cfg)
    n = env.num_agents
    gen = torch.Generator(device=env.device)
    gen.manual_seed(action_seed)
    obs = env.reset()
    episodes = 0
    mask_lo = env.obs_len - 5

    def sample():
        if mode == "random":
            return torch.randint(0, 5, (num_envs, n), generator=gen, device=env.device, dtype=torch.int8)
        if mode == "masked":
            m = obs[:, :, mask_lo:] > 0.5  # NO_OP is always valid
            w = torch.rand((num_envs, n, 5), generator=gen, device=env.device) * m
            return w.argmax(dim=2).to(torch.int8)
        raise ValueError(f"Unsupported mode: {mode}")

    done_count = torch.zeros((), dtype=torch.int64, device=env.device)
    for _ in range(warmup_steps):
        obs = env.step(sample())["obs"]
    torch.cuda.synchronize(env.device)
    t0 = time.perf_counter()
    for _ in range(steps):
        out = env.step(sample())
        obs = out["obs"]
        done_count += ((out["terminated"] | out["truncated"]) != 0).sum()
    torch.cuda.synchronize(env.device)
    elapsed_s = time.perf_counter() - t0
    env.poll_error()
    episodes = int(done_count.item())
    return _result(mode, steps, warmup_steps, episodes, elapsed_s, env_config, num_envs, n, {})


def _result(mode, steps, warmup_steps, episodes, elapsed_s, env_config, num_envs, num_agents, extra) -> dict:
    env_steps = steps * num_envs
    res = {
        "mode": mode,
        "steps": steps,
        "warmup_steps": warmup_steps,
        "episodes_completed": episodes,
        "elapsed_s": elapsed_s,
        "steps_per_s": env_steps / elapsed_s,
        "episodes_per_s": episodes / elapsed_s,
        "mean_step_ms": 1000.0 * elapsed_s / steps,
        "env_config": env_config,
        # extensions
        "num_envs": num_envs,
        "agent_steps_per_s": env_steps * num_agents / elapsed_s,
    }
    res.update(extra)
    return res


def save_results(results: list, output_dir: Path):
    output_dir.mkdir(parents=True, exist_ok=True)
    timestamp = datetime.now(timezone.utc).strftime("%Y-%m-%d_%H-%M-%S")
    json_path = output_dir / f"multi_agent_env_benchmark_{timestamp}.json"
    csv_path = output_dir / f"multi_agent_env_benchmark_{timestamp}.csv"
    with json_path.open("w", encoding="utf-8") as f:
        json.dump(results, f, indent=2)
    fieldnames = ["mode", "steps", "warmup_steps", "episodes_completed", "elapsed_s", "steps_per_s", "episodes_per_s",
                  "mean_step_ms"]
    with csv_path.open("w", encoding="utf-8", newline="") as f:
        writer = csv.DictWriter(f, fieldnames=fieldnames)
        writer.writeheader()
        for row in results:
            writer.writerow({k: row[k] for k in fieldnames})
    return json_path, csv_path


def parse_args(argv=None) -> argparse.Namespace:
    p = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    p.add_argument("--env-name", default="ReferenceModel-2-1")
    p.add_argument("--num-agents", type=int, default=4)
    p.add_argument("--sensor-range", type=int, default=2)
    p.add_argument("--steps-per-episode", type=int, default=100)
    p.add_argument("--steps", type=int, default=40000)
    p.add_argument("--warmup-steps", type=int, default=5000)
    p.add_argument("--modes", default="random,masked", help="Comma-separated modes: random, masked")
    p.add_argument("--env-seed", type=int, default=123)
    p.add_argument("--action-seed", type=int, default=999)
    p.add_argument("--deterministic", action="store_true")
    p.add_argument("--info-mode", choices=["lite", "full"], default="lite")
    p.add_argument("--output-dir", type=Path, default=Path("experiments/results/benchmarks"))
    p.add_argument("--assert-min-steps-per-s", type=float, default=None,
                   help="If set, assert that the random-mode throughput reaches this threshold.")
    # extensions
    p.add_argument("--num-envs", type=int, default=1, help="> 1: vectorised tensor API (B envs per launch)")
    p.add_argument("--device", default="cuda:0")
    p.add_argument("--fused", type=int, default=0,
                   help="with --num-envs > 1 and mode 'masked': steps per launch, policy evaluated in-kernel")
    return p.parse_args(argv)


def main(argv=None):
    args = parse_args(argv)
    modes = [m.strip() for m in args.modes.split(",") if m.strip()]
    args.modes = modes
    env_config = _build_env_config(args)
    results = []
    for mode in modes:
        if args.num_envs > 1 and args.fused > 0 and mode == "masked":
            r = run_benchmark_fused(env_config, args.steps, args.warmup_steps, args.action_seed, args.num_envs,
                                    args.device, args.fused)
        elif args.num_envs > 1:
            r = run_benchmark_vectorized(env_config, mode, args.steps, args.warmup_steps, args.action_seed,
                                         args.num_envs, args.device)
        else:
            r = run_benchmark(dict(env_config, device=args.device), mode, args.steps, args.warmup_steps,
                              args.action_seed)
        results.append(r)
        print(f"[{mode}] steps/s={r['steps_per_s']:.2f}, episodes/s={r['episodes_per_s']:.3f}, "
              f"mean_step_ms={r['mean_step_ms']:.3f}, agent-steps/s={r['agent_steps_per_s']:.0f}")
    json_path, csv_path = save_results(results, args.output_dir)
    print(f"Saved benchmark JSON to {json_path}")
    print(f"Saved benchmark CSV to {csv_path}")
    if args.assert_min_steps_per_s is not None:
        rr = next((row for row in results if row["mode"] == "random"), None)
        if rr is None:
            raise ValueError("Assertion requested but random mode is missing from --modes.")
        if rr["steps_per_s"] < args.assert_min_steps_per_s:
            raise AssertionError(f"Random-mode throughput {rr['steps_per_s']:.2f} steps/s is below "
                                 f"required {args.assert_min_steps_per_s:.2f} steps/s.")
    return results


if __name__ == "__main__":
    main()
