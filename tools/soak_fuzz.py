#!/usr/bin/env python3
"""One-off soak (not part of the test suite): the randomized engine-vs-oracle fuzz of
tests/test_engine_parity_gpu.py with another master seed and many more cases.
Usage: python tools/soak_fuzz.py [master_seed] [cases] [fused]
SOAK_JIT=1 in the environment: every engine is created with jit_specialize (kernels compiled for the case's configuration).
With `fused`, every case also runs mapf_step_many (observations every step) on a fresh engine and compares it, step
for step and in its final state, with the single-step engine."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from trace_util import EngineStepper, OracleStepper, compare_steppers, synth_grids

master = int(sys.argv[1]) if len(sys.argv) > 1 else 777
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 300
fused = len(sys.argv) > 3 and sys.argv[3] == "fused"


def check_fused(grids, cfg, seeds, extra, acts):
    import torch
    one, many = EngineStepper(grids, cfg, seeds=seeds, **extra), EngineStepper(grids, cfg, seeds=seeds, **extra)
    one.reset(); many.reset()
    out = many.env.step_many(torch.from_numpy(acts).to(many.env.device), obs_mode=2)
    out = {k: v.cpu().numpy() for k, v in out.items()}
    for t in range(acts.shape[0]):
        o = one.step(acts[t], auto_reset=True)
        for k in ("obs", "rewards", "terminated", "truncated", "info_all", "info_agent"):
            if not np.array_equal(o[k], out[k][t]):
                raise AssertionError(f"fused {k} differs at step {t}")
    sa, sb = one.env.get_state(), many.env.get_state()
    for k in sa:
        if not np.array_equal(sa[k], sb[k]):
            raise AssertionError(f"fused final state {k} differs")

rng = np.random.default_rng(master)
t0 = time.time(); done = 0
for case in range(cases):
    H, W = int(rng.integers(1, 65)), int(rng.integers(2, 65))
    N = int(rng.integers(1, min(64, max(1, (H * W) // 3)) + 1))
    cfg = {
        "env_name": "synthetic", "num_agents": N, "sensor_range": int(rng.integers(0, 6)),
        "steps_per_episode": int(rng.integers(3, 70)),
        "normalize_goal_delta": bool(rng.integers(0, 2)), "include_goal_distance": bool(rng.integers(0, 2)),
        "include_action_mask_in_obs": bool(rng.integers(0, 2)),
        "include_blocking_pressure_in_obs": bool(rng.integers(0, 2)),
        "lifelong_mapf": bool(rng.integers(0, 2)), "enable_lock_metrics": bool(rng.integers(0, 4) > 0),
        "deadlock_window_steps": int(rng.integers(1, 65)), "livelock_window_steps": int(rng.integers(1, 65)),
        "lock_nearby_manhattan": int(rng.integers(1, 6)), "lock_min_neighbors": int(rng.integers(1, 4)),
        "lock_progress_epsilon": float(rng.choice([0, 0.5, 1, 2, -1, 3.7])),
    }
    B = int(rng.integers(1, 40))
    density = float(rng.choice([0.0, 0.1, 0.3]))
    grids = synth_grids(B, H, W, density, N, base_seed=int(rng.integers(0, 10**6)))
    seeds = [int(x) for x in rng.integers(0, 10**6, size=B)]
    lanes = [l for l in (4, 8, 16, 32, 64) if l >= N]
    extra = {"lanes_per_env": int(rng.choice(lanes))} if rng.random() < 0.5 else {}
    if rng.random() < 0.3:
        extra["force_pair_walk"] = True
    if rng.random() < 0.2:
        extra["force_sequential_reset"] = True
    if rng.random() < 0.2:
        extra["force_generic_kernel"] = True
    if os.environ.get("SOAK_JIT"):  # every case through step kernels compiled for its configuration (2-5 s per case)
        extra["jit_specialize"] = True
    p = rng.dirichlet(np.ones(5))
    acts = rng.choice(5, size=(90, B, N), p=p).astype(np.int8)
    try:
        compare_steppers(EngineStepper(grids, cfg, seeds=seeds, **extra), OracleStepper(grids, cfg, seeds=seeds), acts)
        if fused:
            check_fused(grids, cfg, seeds, extra, acts)
    except AssertionError as exc:
        print(f"FAIL case {case}: cfg={cfg} B={B} HxW={H}x{W} density={density} extra={extra}: {exc}", flush=True)
        sys.exit(1)
    done += 1
    if case % 25 == 0:
        print(f"case {case} ok ({time.time() - t0:.0f} s)", flush=True)
print(f"soak ok: {done} cases, master seed {master}, {time.time() - t0:.0f} s")
