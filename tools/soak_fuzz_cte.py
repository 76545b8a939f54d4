#!/usr/bin/env python3
"""One-off soak for the single-agent (CTE) env: tests/test_single_agent_gpu.py's fuzz with another master seed, more
cases and the lanes_per_env knob.  Usage: python tools/soak_fuzz_cte.py [master_seed] [cases]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from trace_util import CteEngineStepper, CteOracleStepper, synth_grids

def eq(what, x, y, t=None):
    if not np.array_equal(np.asarray(x), np.asarray(y)):
        raise AssertionError(f"{what} differs" + (f" at step {t}" if t is not None else ""))

master = int(sys.argv[1]) if len(sys.argv) > 1 else 778
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 200
rng = np.random.default_rng(master); t0 = time.time()
for case in range(cases):
    H, W = int(rng.integers(2, 65)), int(rng.integers(2, 65))
    N = int(rng.integers(1, min(64, max(1, (H * W) // 4)) + 1))
    cfg = {"env_name": "synthetic", "num_agents": N, "steps_per_episode": int(rng.integers(3, 50)),
           "blocking_penalty": float(rng.choice([-0.2, -0.3, -1.0])),
           "move_after_goal_penalty": float(rng.choice([-0.05, -0.07, 0.0]))}
    B = int(rng.integers(1, 30))
    grids = synth_grids(B, H, W, float(rng.choice([0.0, 0.2])), N, base_seed=int(rng.integers(0, 10**6)))
    seeds = [int(x) for x in rng.integers(0, 10**6, size=B)]
    extra = {}
    if rng.random() < 0.5:
        lanes = [l for l in (4, 8, 16, 32, 64) if l >= N and (64 // l) * (H * W + 5 * N) * 4 <= 56 * 1024]
        if lanes:
            extra["lanes_per_env"] = int(rng.choice(lanes))
    try:
        try:
            a = CteEngineStepper(grids, dict(cfg, **extra), seeds=seeds)
        except ValueError as exc:  # the forced group width does not fit 64 KiB of LDS: a legitimate refusal
            if "LDS" not in str(exc):
                raise
            extra = {}
            a = CteEngineStepper(grids, cfg, seeds=seeds)
        b = CteOracleStepper(grids, cfg, seeds=seeds)
        eq("reset obs", a.reset(), b.reset())
        p = rng.dirichlet(np.ones(5))
        for t in range(70):
            acts = rng.choice(5, size=(B, N), p=p).astype(np.int8)
            ra, rb = a.step(acts), b.step(acts)
            for k in ("obs", "reward", "terminated", "truncated", "info"):
                eq(k, ra[k], rb[k], t)
        eq("rng", a.rng_words(), b.rng_words())
    except AssertionError as exc:
        print(f"FAIL case {case}: cfg={cfg} B={B} HxW={H}x{W} extra={extra}: {exc}", flush=True)
        sys.exit(1)
    if case % 25 == 0:
        print(f"case {case} ok ({time.time() - t0:.0f} s)", flush=True)
print(f"cte soak ok: {cases} cases, master seed {master}, {time.time() - t0:.0f} s")
