"""Record golden input/output vectors from the UNMODIFIED reference env.  TEST INFRASTRUCTURE.

Runs only in the build container (needs /root/reference, imported through oracle/ref_harness.py).
Writes small ``.npz`` fixtures to tests/golden/ -- data only: inputs (grids, RNG words, actions,
injected state) and the reference's outputs (observations, rewards, done flags, info, state).
Also exports the reference's named-grid tables (get_grid.py data) to the package data file.

Usage:  python oracle/gen_golden.py            regenerates everything (deterministic: every env is seeded)
        python oracle/gen_golden.py --check    regenerates into a temporary directory and compares every array with
                                               the committed fixtures (exit status 1 on any difference)
"""

from __future__ import annotations

import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)

import ref_harness as rh  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")
NAMED_GRIDS = os.path.join(ROOT, "dl_reference_models_amd", "data", "named_grids.npz")
INFO_ALL_KEYS = (
    "goals_reached_step", "goals_reached_total", "blocking_count_step", "blocking_count_total",
    "deadlock_step", "livelock_step", "deadlock_event_step", "livelock_event_step",
    "deadlock_events_total", "livelock_events_total", "deadlock_steps_total", "livelock_steps_total",
    "completion_ratio", "throughput",
)
NAMED = ["ReferenceModel-1-1", "ReferenceModel-1-2", "ReferenceModel-1-3", "ReferenceModel-1-4",
         "ReferenceModel-2-1", "ReferenceModel-2-1-b", "ReferenceModel-2-2", "ReferenceModel-3-1"]


def synth_grid(seed: int, h: int, w: int, density: float, need_free: int) -> np.ndarray:
    """Build-defined synthetic grid: reseed (+100000) until enough free cells."""
    s = seed
    while True:
        g = (np.random.default_rng(s).random((h, w)) < density).astype(np.uint8)
        if int((g == 0).sum()) >= need_free:
            return g
        s += 100_000


def pcg_words(state: dict) -> np.ndarray:
    s, inc = int(state["state"]["state"]), int(state["state"]["inc"])
    m = (1 << 64) - 1
    return np.array([s >> 64, s & m, inc >> 64, inc & m, int(state["has_uint32"]), int(state["uinteger"])], dtype=np.uint64)


def info_all_vector(env, info_all: dict) -> np.ndarray:
    """14 floats; the two lifelong-only keys are derived from env state when the reference omits them
    (same expressions as MA-env:638,655)."""
    out = np.zeros(14, dtype=np.float32)
    for k, key in enumerate(INFO_ALL_KEYS[:12]):
        out[k] = np.float32(info_all[key])
    out[12] = np.float32(info_all.get("completion_ratio", float(np.mean(env._completed_once_arr))))
    out[13] = np.float32(info_all.get("throughput", info_all["goals_reached_total"] / float(max(env.step_count, 1))))
    return out


def greedy_actions(env, rng, p_greedy: float) -> np.ndarray:
    """Build-defined action stream (SURVEY 8(d) 'greedy'): with prob. p step along the larger-magnitude
    goal-delta axis, else uniform.  Actions are inputs; they are recorded in the fixture."""
    n = env._num_agents
    out = rng.integers(0, 5, size=n)
    for a in range(n):
        if rng.random() < p_greedy:
            dr = int(env._goals_arr[a, 0]) - int(env._positions_arr[a, 0])
            dc = int(env._goals_arr[a, 1]) - int(env._positions_arr[a, 1])
            if abs(dr) >= abs(dc) and dr != 0:
                out[a] = 3 if dr > 0 else 1
            elif dc != 0:
                out[a] = 2 if dc > 0 else 4
    return out.astype(np.int8)


def record_trace(cfg: dict, grids, seeds, T: int, actions: np.ndarray | None = None, action_seed: int = 999,
                 auto_reset: bool = True, greedy: float | None = None) -> dict:
    """Run B reference envs in lockstep for T steps.  Returns arrays (see keys below)."""
    B = len(seeds)
    N = int(cfg["num_agents"])
    envs, words = [], []
    for b in range(B):
        c = dict(cfg, seed=int(seeds[b]))
        words.append(pcg_words(np.random.default_rng(int(seeds[b])).bit_generator.state))
        envs.append(rh.make_reference_env(c, None if grids is None else grids[b]))
    grid_arr = np.stack([e.grid for e in envs]).astype(np.uint8)
    L = envs[0]._single_obs_len
    policy_rng = np.random.default_rng(action_seed)
    if actions is None and greedy is None:
        actions = policy_rng.integers(0, 5, size=(T, B, N)).astype(np.int8)
    elif actions is None:
        actions = np.zeros((T, B, N), np.int8)  # filled in step by step below
    fixed_starts = np.stack([e._starts_arr.copy() for e in envs])
    fixed_goals = np.stack([e._goals_arr.copy() for e in envs])
    out = {
        "config": np.array(json.dumps(cfg)),
        "grids": grid_arr,
        "rng_words": np.stack(words),
        "seeds": np.asarray(seeds, dtype=np.int64),
        "actions": actions,
        "ctor_starts": fixed_starts,
        "ctor_goals": fixed_goals,
        "reset0_obs": np.zeros((B, N, L), np.float32),
        "obs": np.zeros((T, B, N, L), np.float32),
        "rewards": np.zeros((T, B, N), np.float32),
        "terminated": np.zeros((T, B), np.uint8),
        "truncated": np.zeros((T, B), np.uint8),
        "info_all": np.zeros((T, B, 14), np.float32),
        "info_agent": np.zeros((T, B, N, 2), np.uint8),
        "positions": np.zeros((T, B, N, 2), np.int16),
        "goals": np.zeros((T, B, N, 2), np.int16),
        "did_reset": np.zeros((T, B), np.uint8),
        "reset_obs": np.zeros((T, B, N, L), np.float32),
        "reset_positions": np.zeros((T, B, N, 2), np.int16),
        "reset_goals": np.zeros((T, B, N, 2), np.int16),
        "reset0_positions": np.zeros((B, N, 2), np.int16),
        "reset0_goals": np.zeros((B, N, 2), np.int16),
        "final_rng_words": np.zeros((B, 6), np.uint64),
    }
    for b, e in enumerate(envs):
        o, _ = e.reset()
        out["reset0_obs"][b] = np.stack([o[f"agent_{a}"] for a in range(N)])
        out["reset0_positions"][b] = e._positions_arr
        out["reset0_goals"][b] = e._goals_arr
    for t in range(T):
        for b, e in enumerate(envs):
            if greedy is not None:
                actions[t, b] = greedy_actions(e, policy_rng, greedy)
            act = {f"agent_{a}": int(actions[t, b, a]) for a in range(N)}
            o, r, term, trunc, info = e.step(act)
            out["obs"][t, b] = np.stack([o[f"agent_{a}"] for a in range(N)])
            out["rewards"][t, b] = [np.float32(r[f"agent_{a}"]) for a in range(N)]
            out["terminated"][t, b] = term["__all__"]
            out["truncated"][t, b] = trunc["__all__"]
            out["info_all"][t, b] = info_all_vector(e, info["__all__"])
            for a in range(N):
                out["info_agent"][t, b, a, 0] = info[f"agent_{a}"]["blocking"]
                out["info_agent"][t, b, a, 1] = info[f"agent_{a}"]["goal_reached_step"]
            out["positions"][t, b] = e._positions_arr
            out["goals"][t, b] = e._goals_arr
            if auto_reset and (term["__all__"] or trunc["__all__"]):
                o, _ = e.reset()
                out["did_reset"][t, b] = 1
                out["reset_obs"][t, b] = np.stack([o[f"agent_{a}"] for a in range(N)])
                out["reset_positions"][t, b] = e._positions_arr
                out["reset_goals"][t, b] = e._goals_arr
    for b, e in enumerate(envs):
        out["final_rng_words"][b] = pcg_words(e.rng.bit_generator.state)
    return out


def save(name: str, data: dict) -> None:
    path = os.path.join(GOLDEN, name + ".npz")
    np.savez_compressed(path, **data)
    print(f"  {name}.npz  {os.path.getsize(path) / 1024:.1f} KiB")


# ---------------------------------------------------------------------------------------------
def g1_parity_digest():
    """Replica of the reference's tests/test_reference_model_multi_agent_parity.py run: per-step
    tensors plus the SHA-256 digest over the reference's outputs, hashed with our own helper
    (must equal the constants in that test, which are also stored in the fixture)."""
    expected = {
        False: "d58a9e9e0e383f29c5d7f96a1338dfd73c9035dc5335f66b2ce11a0d8e0452de",
        True: "2612dc3eeab5b4fd69d8cbe7fb4f01e35cf52a6f05b07bc955a306ae765c2595",
    }
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from digest_util import TraceHasher  # our own restatement of the digest protocol

    for det in (False, True):
        cfg = {
            "env_name": "ReferenceModel-2-1", "seed": 123, "deterministic": det, "num_agents": 4,
            "steps_per_episode": 100, "sensor_range": 2, "info_mode": "full", "training_execution_mode": "CTDE",
            "render_env": False, "include_action_mask_in_obs": True, "include_blocking_pressure_in_obs": False,
        }
        env = rh.make_reference_env(cfg)
        words = pcg_words(np.random.default_rng(123).bit_generator.state)
        arng = np.random.default_rng(999)
        th = TraceHasher()
        N, L = 4, env._single_obs_len
        rec = {k: [] for k in ("actions", "obs", "rewards", "terminated", "truncated", "info_all", "info_agent",
                               "positions", "goals", "episode_start")}
        reset_obs, reset_pos, reset_goals, summary = [], [], [], []
        for ep in range(3):
            obs, infos = env.reset()
            th.reset_record(ep, obs, infos)
            reset_obs.append(np.stack([obs[f"agent_{a}"] for a in range(N)]))
            reset_pos.append(env._positions_arr.copy())
            reset_goals.append(env._goals_arr.copy())
            rsum = 0.0
            for st in range(140):
                actions = {f"agent_{i}": int(arng.integers(0, 5)) for i in range(4)}
                obs, rewards, term, trunc, infos = env.step(actions)
                rsum += float(sum(rewards.values()))
                th.step_record(ep, st, actions, obs, rewards, term, trunc, infos)
                rec["actions"].append([actions[f"agent_{a}"] for a in range(N)])
                rec["obs"].append(np.stack([obs[f"agent_{a}"] for a in range(N)]))
                rec["rewards"].append([np.float32(rewards[f"agent_{a}"]) for a in range(N)])
                rec["terminated"].append(term["__all__"])
                rec["truncated"].append(trunc["__all__"])
                rec["info_all"].append(info_all_vector(env, infos["__all__"]))
                rec["info_agent"].append([[infos[f"agent_{a}"]["blocking"], infos[f"agent_{a}"]["goal_reached_step"]]
                                          for a in range(N)])
                rec["positions"].append(env._positions_arr.copy())
                rec["goals"].append(env._goals_arr.copy())
                rec["episode_start"].append(st == 0)
                if term["__all__"] or trunc["__all__"]:
                    summary.append([ep, st + 1, round(rsum, 6)])
                    break
        digest = th.hexdigest()
        assert digest == expected[det], (det, digest)
        save(
            "g1_parity_" + ("deterministic" if det else "stochastic"),
            {
                "config": np.array(json.dumps(cfg)),
                "grid": env.grid.astype(np.uint8),
                "rng_words": words,
                "ctor_starts": env._starts_arr.copy() if det else np.zeros((N, 2), np.int16),
                "fixed_goals": np.array([[6, 6], [9, 3], [6, 0], [3, 3]], np.int16) if det else np.zeros((N, 2), np.int16),
                "actions": np.asarray(rec["actions"], np.int8),
                "obs": np.asarray(rec["obs"], np.float32),
                "rewards": np.asarray(rec["rewards"], np.float32),
                "terminated": np.asarray(rec["terminated"], np.uint8),
                "truncated": np.asarray(rec["truncated"], np.uint8),
                "info_all": np.asarray(rec["info_all"], np.float32),
                "info_agent": np.asarray(rec["info_agent"], np.uint8),
                "positions": np.asarray(rec["positions"], np.int16),
                "goals": np.asarray(rec["goals"], np.int16),
                "reset_obs": np.asarray(reset_obs, np.float32),
                "reset_positions": np.asarray(reset_pos, np.int16),
                "reset_goals": np.asarray(reset_goals, np.int16),
                "summary": np.asarray(summary, np.float64),
                "digest": np.array(digest),
            },
        )


def g2_g3_g4_batches():
    # G2: c2-like, all four obs flags on
    cfg = {"env_name": "synthetic", "num_agents": 4, "sensor_range": 2, "steps_per_episode": 100,
           "include_action_mask_in_obs": True, "include_goal_distance": True, "include_blocking_pressure_in_obs": True}
    B = 8
    grids = [synth_grid(10_000 + b, 16, 16, 0.20, 8) for b in range(B)]
    save("g2_c2_16x16_n4", record_trace(cfg, grids, list(range(B)), 300))

    tr = record_trace(cfg, grids, list(range(50, 50 + B)), 300, greedy=0.85)
    succ = int((tr["terminated"].astype(bool) & ~tr["truncated"].astype(bool)).sum())
    print(f"  g2 greedy: success terminations={succ} blocking={tr['info_all'][:, :, 2].sum():.0f} "
          f"deadlock events={tr['info_all'][:, :, 6].sum():.0f}")
    assert succ >= 1
    save("g2_c2_16x16_n4_greedy", tr)

    # G3: c3-like, deadlock-heavy density, lock metrics on; must contain both event kinds
    cfg = {"env_name": "synthetic", "num_agents": 8, "sensor_range": 2, "steps_per_episode": 100,
           "include_action_mask_in_obs": True}
    B = 16
    grids = [synth_grid(10_000 + b, 32, 32, 0.40, 16) for b in range(B)]
    tr = record_trace(cfg, grids, list(range(B)), 300)
    dl, ll = tr["info_all"][:, :, 6].sum(), tr["info_all"][:, :, 7].sum()
    print(f"  g3 events: deadlock={dl:.0f} livelock={ll:.0f}")
    assert dl >= 1 and ll >= 1
    save("g3_c3_32x32_n8", tr)

    # G3b: greedy-ish action stream on a tight corridor grid to get many deadlocks + blocking
    cfg = {"env_name": "synthetic", "num_agents": 6, "sensor_range": 1, "steps_per_episode": 60,
           "include_action_mask_in_obs": True, "deadlock_window_steps": 3, "livelock_window_steps": 6}
    B = 6
    grids = [synth_grid(20_000 + b, 6, 7, 0.25, 12) for b in range(B)]
    acts = np.random.default_rng(5).choice(5, size=(400, B, 6), p=[0.1, 0.1, 0.6, 0.1, 0.1]).astype(np.int8)
    tr = record_trace(cfg, grids, list(range(100, 100 + B)), 400, actions=acts)
    print(f"  g3b events: deadlock={tr['info_all'][:, :, 6].sum():.0f} livelock={tr['info_all'][:, :, 7].sum():.0f} "
          f"blocking={tr['info_all'][:, :, 2].sum():.0f} goals={tr['info_all'][:, :, 0].sum():.0f} "
          f"success_terms={(tr['terminated'] & ~tr['truncated'].astype(bool)).sum()}")
    assert tr["info_all"][:, :, 6].sum() >= 1 and tr["info_all"][:, :, 2].sum() >= 1
    save("g3b_tight_6x7_n6", tr)

    # G4: c5-like lifelong (pins _assign_new_goal + integers stream)
    cfg = {"env_name": "synthetic", "num_agents": 64, "sensor_range": 2, "steps_per_episode": 256,
           "include_action_mask_in_obs": True, "lifelong_mapf": True}
    B = 2
    grids = [synth_grid(10_000 + b, 64, 64, 0.20, 128) for b in range(B)]
    tr = record_trace(cfg, grids, list(range(B)), 300, greedy=0.8)
    print(f"  g4 respawns: {tr['info_all'][:, :, 0].sum():.0f} max/step {tr['info_all'][:, :, 0].max():.0f}")
    assert tr["info_all"][:, :, 0].sum() >= 20
    save("g4_c5_64x64_n64_lifelong", tr)

    # G4b: small dense lifelong -> many respawns incl. multiple per step and k small
    cfg = {"env_name": "synthetic", "num_agents": 10, "sensor_range": 3, "steps_per_episode": 50,
           "include_action_mask_in_obs": True, "lifelong_mapf": True, "include_goal_distance": True,
           "normalize_goal_delta": False, "deadlock_window_steps": 3, "livelock_window_steps": 5,
           "lock_nearby_manhattan": 3, "lock_min_neighbors": 2, "lock_progress_epsilon": 0.5}
    B = 6
    grids = [synth_grid(30_000 + b, 5, 9, 0.15, 22) for b in range(B)]
    tr = record_trace(cfg, grids, list(range(40, 40 + B)), 400)
    print(f"  g4b respawns: {tr['info_all'][:, :, 0].sum():.0f} max/step {tr['info_all'][:, :, 0].max():.0f}")
    assert tr["info_all"][:, :, 0].max() >= 2
    save("g4b_lifelong_5x9_n10", tr)

    # G7: c1-like synthetic 10x10, 2 agents, sr 1 (default flags)
    cfg = {"env_name": "synthetic", "num_agents": 2, "sensor_range": 1, "steps_per_episode": 100}
    grids = [synth_grid(10_000 + b, 10, 10, 0.10, 4) for b in range(4)]
    save("g7_c1_10x10_n2", record_trace(cfg, grids, list(range(4)), 250))

    # G8: lock metrics off, no pressure, no normalisation, N=1 and N=3, sr=0
    cfg = {"env_name": "synthetic", "num_agents": 3, "sensor_range": 0, "steps_per_episode": 30,
           "include_action_mask_in_obs": True, "include_blocking_pressure_in_obs": False,
           "normalize_goal_delta": False, "enable_lock_metrics": False}
    grids = [synth_grid(40_000 + b, 4, 5, 0.10, 6) for b in range(4)]
    save("g8_sr0_nolock_4x5_n3", record_trace(cfg, grids, list(range(4)), 200))
    cfg = {"env_name": "synthetic", "num_agents": 1, "sensor_range": 4, "steps_per_episode": 25,
           "include_action_mask_in_obs": True, "include_goal_distance": True}
    grids = [synth_grid(50_000 + b, 3, 4, 0.10, 2) for b in range(4)]
    save("g8_sr4_3x4_n1", record_trace(cfg, grids, list(range(4)), 200))
    # wide windows (> 32 steps) exercise the long-history path
    cfg = {"env_name": "synthetic", "num_agents": 5, "sensor_range": 1, "steps_per_episode": 150,
           "deadlock_window_steps": 40, "livelock_window_steps": 50}
    grids = [synth_grid(60_000 + b, 5, 5, 0.10, 10) for b in range(4)]
    acts = np.random.default_rng(11).integers(0, 5, size=(320, 4, 5)).astype(np.int8)
    acts[(np.arange(320) % 75) >= 15] = 2  # long runs of RIGHT: everybody ends up pushing a wall
    tr = record_trace(cfg, grids, list(range(4)), 320, actions=acts)
    assert tr["info_all"][:, :, 6].sum() >= 1 and tr["info_all"][:, :, 7].sum() >= 1
    print(f"  g8 wide windows events: dl={tr['info_all'][:, :, 6].sum():.0f} ll={tr['info_all'][:, :, 7].sum():.0f}")
    save("g8_widewin_5x5_n5", tr)


def g10_wide_groups():
    """Wide groups recorded from the reference itself: N = 20 finite episodes (32-lane groups: LDS cell map, B1 after
    the map) with a goal-seeking stream so that episodes also end in success, and N = 40 lifelong on a dense small
    grid (64-lane groups, many respawns per step, small candidate sets)."""
    cfg = {"env_name": "synthetic", "num_agents": 20, "sensor_range": 2, "steps_per_episode": 40,
           "include_action_mask_in_obs": True, "deadlock_window_steps": 4, "livelock_window_steps": 8}
    B = 3
    grids = [synth_grid(70_000 + b, 12, 12, 0.12, 40) for b in range(B)]
    tr = record_trace(cfg, grids, list(range(200, 200 + B)), 130, greedy=0.7)
    print(f"  g10 n20: goals={tr['info_all'][:, :, 0].sum():.0f} blocking={tr['info_all'][:, :, 2].sum():.0f} "
          f"dl={tr['info_all'][:, :, 6].sum():.0f} ll={tr['info_all'][:, :, 7].sum():.0f}")
    assert tr["info_all"][:, :, 0].sum() >= 10
    save("g10_n20_finite_12x12", tr)

    cfg = {"env_name": "synthetic", "num_agents": 40, "sensor_range": 2, "steps_per_episode": 45,
           "include_action_mask_in_obs": True, "lifelong_mapf": True, "deadlock_window_steps": 4,
           "livelock_window_steps": 8}
    B = 2
    grids = [synth_grid(80_000 + b, 11, 13, 0.10, 90) for b in range(B)]
    tr = record_trace(cfg, grids, list(range(300, 300 + B)), 120, greedy=0.7)
    print(f"  g10 n40 lifelong respawns: {tr['info_all'][:, :, 0].sum():.0f} max/step {tr['info_all'][:, :, 0].max():.0f}")
    assert tr["info_all"][:, :, 0].max() >= 2
    save("g10_n40_lifelong_11x13", tr)


def g11_exactly_2n_free_cells():
    """F == 2N: rng.choice(F, 2N, replace=False) starts Floyd at j = 0, and a bounded draw with bound 0 consumes no
    random number -- the alignment of the whole stream depends on it.  Several resets per env."""
    n, h, w = 3, 4, 6
    rng = np.random.default_rng(8)
    grids = []
    for _ in range(4):
        g = np.ones((h, w), np.uint8)
        g.reshape(-1)[rng.choice(h * w, size=2 * n, replace=False)] = 0
        grids.append(g)
    cfg = {"env_name": "synthetic", "num_agents": n, "sensor_range": 1, "steps_per_episode": 7,
           "include_action_mask_in_obs": True}
    tr = record_trace(cfg, grids, list(range(700, 704)), 60)
    save("g11_f_equals_2n_4x6_n3", tr)


def g12_lifelong_exactly_2n_free_cells():
    """Lifelong mode with F == 2N: a respawn has k = F - N - (N - 1) + overlap = 1 + overlap candidates, so whenever
    no agent stands on another agent's goal rng.integers(1) is a bounded draw with bound 0 that consumes NOTHING
    (and the episode resets start Floyd at j = 0 like g11).  Connected free cells so that arrivals happen."""
    n = 3
    grids = []
    g = np.ones((4, 5), np.uint8)
    g[1:3, 1:4] = 0  # 2x3 open block: 6 free cells
    grids.append(g)
    g = np.ones((4, 5), np.uint8)
    g[0, 0:3] = 0
    g[1, 0] = g[1, 2] = 0
    g[2, 2] = 0  # a bent corridor, 6 free cells
    grids.append(g)
    g = np.ones((4, 5), np.uint8)
    g[3, 0:5] = 0
    g[2, 4] = 0  # an L, 6 free cells
    grids.append(g)
    cfg = {"env_name": "synthetic", "num_agents": n, "sensor_range": 1, "steps_per_episode": 25, "lifelong_mapf": True,
           "include_action_mask_in_obs": True}
    tr = record_trace(cfg, grids, [900, 901, 902], 120)
    assert float(tr["info_all"][:, :, 0].sum()) >= 10, "the fixture must contain respawns"
    save("g12_lifelong_f_equals_2n", tr)


def g5_named_and_deterministic():
    """Every named grid with its fixed start/goal table (deterministic), 4 or 2 agents."""
    for name in NAMED:
        if name == "ReferenceModel-2-1-b":
            continue  # no fixed table (get_grid.py:790-792); covered stochastic below
        n = 2 if name in ("ReferenceModel-1-1", "ReferenceModel-1-2", "ReferenceModel-1-3") else 4
        cfg = {"env_name": name, "num_agents": n, "sensor_range": 2, "steps_per_episode": 60, "deterministic": True,
               "include_action_mask_in_obs": True}
        save("g5_named_" + name.split("-", 1)[1].replace("-", "_"), record_trace(cfg, None, [123], 200))
    cfg = {"env_name": "ReferenceModel-2-1-b", "num_agents": 5, "sensor_range": 2, "steps_per_episode": 60}
    save("g5_named_2_1_b", record_trace(cfg, None, [123], 200))
    # deterministic + lifelong: goals persist across reset (MA-env:452-455 keeps _goals_arr)
    cfg = {"env_name": "ReferenceModel-1-4", "num_agents": 4, "sensor_range": 1, "steps_per_episode": 30,
           "deterministic": True, "lifelong_mapf": True}
    tr = record_trace(cfg, None, [7], 300, greedy=0.7)
    print(f"  g5 det+lifelong respawns: {tr['info_all'][:, :, 0].sum():.0f}")
    assert tr["info_all"][:, :, 0].sum() >= 5
    save("g5_det_lifelong_1_4", tr)


def g5_micro_cases():
    """Hand-set states (the same private arrays the reference's tests poke,
    tests/test_reference_model_multi_agent_invariants.py:28-38) + scripted actions."""
    cases = {}

    def run_case(name, cfg, grid, positions, goals, action_rows, reset_lock=True, zero_goals_total=True):
        env = rh.make_reference_env(cfg, grid)
        env.reset()
        N = len(positions)
        np.copyto(env._positions_arr, np.asarray(positions, np.int16))
        np.copyto(env._starts_arr, np.asarray(positions, np.int16))
        np.copyto(env._goals_arr, np.asarray(goals, np.int16))
        env._rebuild_goal_owner()
        env._rebuild_occupancy_owner()
        env._reached_arr[:] = False
        env._completed_once_arr[:] = False
        env.goal_reached_once = dict.fromkeys(env.agents, False)
        env._blocking_pressure_prev_arr.fill(0.0)
        if zero_goals_total:
            env._episode_goals_reached_total = 0.0
        if reset_lock:
            env._reset_lock_tracking()
        env.step_count = 0
        words = pcg_words(env.rng.bit_generator.state)
        L = env._single_obs_len
        T = len(action_rows)
        rec = {"obs": np.zeros((T, N, L), np.float32), "rewards": np.zeros((T, N), np.float32),
               "terminated": np.zeros(T, np.uint8), "truncated": np.zeros(T, np.uint8),
               "info_all": np.zeros((T, 14), np.float32), "info_agent": np.zeros((T, N, 2), np.uint8),
               "positions": np.zeros((T, N, 2), np.int16), "goals": np.zeros((T, N, 2), np.int16)}
        for t, row in enumerate(action_rows):
            o, r, term, trunc, info = env.step({f"agent_{a}": int(row[a]) for a in range(N)})
            rec["obs"][t] = np.stack([o[f"agent_{a}"] for a in range(N)])
            rec["rewards"][t] = [np.float32(r[f"agent_{a}"]) for a in range(N)]
            rec["terminated"][t], rec["truncated"][t] = term["__all__"], trunc["__all__"]
            rec["info_all"][t] = info_all_vector(env, info["__all__"])
            for a in range(N):
                rec["info_agent"][t, a] = [info[f"agent_{a}"]["blocking"], info[f"agent_{a}"]["goal_reached_step"]]
            rec["positions"][t] = env._positions_arr
            rec["goals"][t] = env._goals_arr
        cases[name + ".config"] = np.array(json.dumps(cfg))
        cases[name + ".grid"] = env.grid.astype(np.uint8)
        cases[name + ".positions0"] = np.asarray(positions, np.int16)
        cases[name + ".goals0"] = np.asarray(goals, np.int16)
        cases[name + ".rng_words"] = words
        cases[name + ".actions"] = np.asarray(action_rows, np.int8)
        for k, v in rec.items():
            cases[name + "." + k] = v

    NO, UP, RT, DN, LT = 0, 1, 2, 3, 4
    open5 = np.zeros((5, 5), np.uint8)
    # (every case is seeded: the recorded RNG words and the lifelong respawns must not depend on OS entropy)
    base = {"env_name": "synthetic", "seed": 11, "sensor_range": 1, "steps_per_episode": 20,
            "include_action_mask_in_obs": True}
    # follow, leader has lower index: both move
    run_case("follow_leader_low", dict(base, num_agents=2), open5, [(2, 2), (2, 1)], [(0, 0), (0, 4)], [[RT, RT]])
    # follow, leader has higher index: follower blocked
    run_case("follow_leader_high", dict(base, num_agents=2), open5, [(2, 1), (2, 2)], [(0, 0), (0, 4)], [[RT, RT]])
    # head-on swap: both stay
    run_case("swap", dict(base, num_agents=2), open5, [(2, 1), (2, 2)], [(0, 0), (0, 4)], [[RT, LT]])
    # 4-cycle rotation: all stay
    run_case("cycle4", dict(base, num_agents=4), open5, [(1, 1), (1, 2), (2, 2), (2, 1)],
             [(4, 4), (4, 3), (4, 2), (4, 1)], [[RT, DN, LT, UP]])
    # contention for one free cell: lower index wins
    run_case("contention", dict(base, num_agents=3), open5, [(2, 1), (2, 3), (1, 2)], [(0, 0), (0, 4), (4, 4)],
             [[RT, LT, DN]])
    # out of bounds + obstacle
    wall = open5.copy()
    wall[1, 1] = 1
    run_case("oob_obstacle", dict(base, num_agents=2), wall, [(0, 0), (1, 0)], [(4, 4), (4, 0)],
             [[UP, RT], [LT, RT], [NO, UP]])
    # both reach goals -> +0.5 +1 each, terminated not truncated
    run_case("both_reach", dict(base, num_agents=2), open5, [(0, 0), (4, 4)], [(0, 1), (4, 3)], [[RT, LT]])
    # one reaches earlier, other later: sticky reached + final success
    run_case("staggered_reach", dict(base, num_agents=2), open5, [(0, 0), (4, 4)], [(0, 1), (4, 2)],
             [[RT, LT], [NO, LT]])
    # leave goal after reaching then truncation: -1 for the one off goal, term and trunc
    run_case("truncation", dict(base, num_agents=2, steps_per_episode=3), open5, [(0, 0), (4, 4)], [(0, 1), (2, 2)],
             [[RT, NO], [LT, NO], [NO, NO]])
    # blocking pressure 0,1,1,0 (tests/...invariants.py:119-147) on ReferenceModel-1-3
    _, gg = rh.load_reference()
    g13 = gg.get_grid("ReferenceModel-1-3")
    cfg13 = {"env_name": "ReferenceModel-1-3", "seed": 123, "num_agents": 2, "steps_per_episode": 20, "sensor_range": 1,
             "include_action_mask_in_obs": False, "include_blocking_pressure_in_obs": True}
    run_case("blocking_pressure", cfg13, None, [(2, 0), (2, 1)], [(2, 2), (2, 1)],
             [[RT, NO], [RT, NO], [NO, NO], [NO, NO]], reset_lock=False)
    # deadlock window 2 (tests/...lock_metrics.py:44-63)
    cfgl = {"env_name": "ReferenceModel-1-3", "seed": 123, "num_agents": 2, "steps_per_episode": 50, "sensor_range": 1,
            "deadlock_window_steps": 2, "livelock_window_steps": 4, "lock_nearby_manhattan": 2,
            "lock_progress_epsilon": 1, "lock_min_neighbors": 1}
    run_case("deadlock_on_goal_blocker", cfgl, None, [(2, 0), (2, 1)], [(2, 2), (2, 1)],
             [[RT, NO], [RT, NO], [RT, NO]], zero_goals_total=False)
    # sticky flags vs current state (tests/...lock_metrics.py:66-86)
    run_case("deadlock_not_sticky", cfgl, None, [(2, 0), (2, 2)], [(2, 1), (4, 2)],
             [[RT, NO], [LT, LT], [RT, NO], [RT, NO]], zero_goals_total=False)
    # livelock: two agents oscillating next to each other, no distance reduction
    cfgv = dict(base, num_agents=2, deadlock_window_steps=2, livelock_window_steps=4, steps_per_episode=30)
    run_case("livelock_oscillation", cfgv, open5, [(2, 1), (2, 3)], [(0, 0), (0, 4)],
             [[UP, UP], [DN, DN], [UP, UP], [DN, DN], [UP, UP], [DN, DN]])
    # lifelong immediate respawn (tests/...lifelong.py:73-94) + completion ratio / throughput
    cfgll = {"env_name": "ReferenceModel-2-1", "seed": 123, "num_agents": 2, "steps_per_episode": 20, "sensor_range": 2,
             "lifelong_mapf": True}
    run_case("lifelong_respawn", cfgll, None, [(0, 0), (0, 2)], [(0, 1), (0, 3)], [[RT, NO], [NO, RT], [NO, NO]])
    # lifelong: agent standing on its goal at NO_OP also respawns; two respawns in one step
    run_case("lifelong_double", dict(base, num_agents=3, lifelong_mapf=True), open5, [(0, 0), (4, 4), (2, 2)],
             [(0, 1), (4, 3), (2, 2)], [[RT, LT, NO], [NO, NO, NO]])
    # lifelong, one agent on a 1x3 strip: two candidate cells after the old goal is released (k = 2; the k = 1 case,
    # a bounded draw that consumes nothing, is pinned by the batch fixture g12_lifelong_f_equals_2n)
    tiny = np.array([[0, 0, 0]], np.uint8)
    run_case("lifelong_single_agent_1x3", dict(base, num_agents=1, lifelong_mapf=True, sensor_range=1), tiny, [(0, 0)],
             [(0, 1)], [[RT], [RT], [LT]])
    np.savez_compressed(os.path.join(GOLDEN, "g5_micro_cases.npz"), **cases)
    print(f"  g5_micro_cases.npz  {os.path.getsize(os.path.join(GOLDEN, 'g5_micro_cases.npz')) / 1024:.1f} KiB "
          f"({len([k for k in cases if k.endswith('.config')])} cases)")

    # known-answer arrays of tests/get_obs.py:141-164 (hand-checked 3x3 local obs + masks)
    np.savez_compressed(
        os.path.join(GOLDEN, "g5_get_obs_known_answer.npz"),
        # scene of tests/get_obs.py:9,34-35; the expectations are 3x3, i.e. sensor_range 1 (the script's
        # own sensor_range=2 object is stale)
        grid=np.array([[1, 1, 0, 1], [0, 0, 0, 1], [1, 0, 1, 1], [0, 0, 0, 0]], np.uint8),
        positions=np.array([[1, 1], [3, 3]], np.int16), goals=np.array([[1, 2], [0, 2]], np.int16),
        sensor_range=np.int32(1),
        expected_obs_agent_0=np.array([[1, 1, 4], [0, 0, 3], [1, 0, 1]], np.uint8),
        expected_mask_agent_0=np.array([1, 0, 1, 1, 1], np.int8),
        expected_obs_agent_1=np.array([[1, 1, 1], [0, 0, 1], [1, 1, 1]], np.uint8),
        expected_mask_agent_1=np.array([1, 0, 0, 0, 1], np.int8),
    )


def g5_error_paths():
    """ValueError mid-loop after partial mutation (MA-env:504-506): state after the exception."""
    cfg = {"env_name": "synthetic", "num_agents": 3, "sensor_range": 1, "steps_per_episode": 20}
    grid = np.zeros((4, 4), np.uint8)
    env = rh.make_reference_env(dict(cfg, seed=3), grid)
    words = pcg_words(np.random.default_rng(3).bit_generator.state)
    env.reset()
    pos0, goals0 = env._positions_arr.copy(), env._goals_arr.copy()
    # find a first action for agent 0 that moves it
    moved_action = None
    for a in (1, 2, 3, 4):
        nxt = env.get_next_position(a, env._positions_arr[0])
        if 0 <= nxt[0] < 4 and 0 <= nxt[1] < 4 and env._occupancy_owner[nxt[0], nxt[1]] == -1:
            moved_action = a
            break
    actions = [moved_action, 7, 2]
    raised = False
    try:
        env.step({f"agent_{i}": actions[i] for i in range(3)})
    except ValueError:
        raised = True
    assert raised
    np.savez_compressed(
        os.path.join(GOLDEN, "g5_bad_action.npz"),
        config=np.array(json.dumps(cfg)), grid=grid, rng_words=words, seed=np.int64(3),
        actions=np.asarray(actions, np.int8), positions0=pos0, goals0=goals0,
        positions_after=env._positions_arr.copy(), step_count_after=np.int32(env.step_count),
    )
    print("  g5_bad_action.npz")


def g6_rng_known_answers():
    rows = []
    for seed in (0, 1, 7, 123, 2**31 - 1, 2**40 + 17):
        for (F, S) in ((4, 4), (8, 2), (30, 16), (150, 8), (614, 16), (1024, 16), (3277, 128), (4096, 128)):
            rng = np.random.default_rng(seed)
            words = pcg_words(rng.bit_generator.state)
            c1 = rng.choice(F, size=S, replace=False)
            ks = [1, 2, 3, 5, 17, 1000, 4095]
            ints = [int(rng.integers(k)) for k in ks]
            c2 = rng.choice(F, size=S, replace=False)
            rows.append((seed, F, S, words, c1, ints, c2, pcg_words(rng.bit_generator.state)))
    np.savez_compressed(
        os.path.join(GOLDEN, "g6_rng.npz"),
        seeds=np.array([r[0] for r in rows], np.int64), F=np.array([r[1] for r in rows], np.int64),
        S=np.array([r[2] for r in rows], np.int64), words=np.stack([r[3] for r in rows]),
        choice1=np.array([np.pad(r[4], (0, 128 - len(r[4])), constant_values=-1) for r in rows], np.int64),
        ks=np.array([1, 2, 3, 5, 17, 1000, 4095], np.int64), ints=np.array([r[5] for r in rows], np.int64),
        choice2=np.array([np.pad(r[6], (0, 128 - len(r[6])), constant_values=-1) for r in rows], np.int64),
        final_words=np.stack([r[7] for r in rows]),
    )
    print("  g6_rng.npz")


def g9_benchmark_pin():
    """Harness pin: the reference benchmark loop on 2-1 / 2 agents / deterministic,
    20000 steps after 3000 warm-up completes exactly this many episodes."""
    cfg = {"env_name": "ReferenceModel-2-1", "seed": 123, "deterministic": True, "num_agents": 2,
           "steps_per_episode": 100, "sensor_range": 2, "info_mode": "lite", "training_execution_mode": "CTDE",
           "render_env": False}
    env = rh.make_reference_env(cfg)
    rng = np.random.default_rng(999)
    env.reset()
    episodes = 0
    for i in range(3000 + 20000):
        acts = {aid: int(rng.integers(0, env.action_space.n)) for aid in env.agents}
        _, _, term, trunc, _ = env.step(acts)
        if term["__all__"] or trunc["__all__"]:
            if i >= 3000:
                episodes += 1
            env.reset()
    np.savez_compressed(os.path.join(GOLDEN, "g9_benchmark_pin.npz"), config=np.array(json.dumps(cfg)),
                        steps=np.int64(20000), warmup=np.int64(3000), episodes_completed=np.int64(episodes),
                        final_positions=env._positions_arr.copy())
    print(f"  g9_benchmark_pin.npz episodes_completed={episodes}")


def record_cte_trace(cfg: dict, grids, seeds, T: int, greedy: float, action_seed: int = 77) -> dict:
    """Single-agent (CTE) sibling env (reference reference_model_single_agent.py): B envs in lockstep."""
    B, N = len(seeds), int(cfg["num_agents"])
    envs, words = [], []
    for b in range(B):
        words.append(pcg_words(np.random.default_rng(int(seeds[b])).bit_generator.state))
        envs.append(rh.make_reference_single_agent_env(dict(cfg, seed=int(seeds[b])), None if grids is None else grids[b]))
    L = int(envs[0].observation_space.shape[0])
    rng = np.random.default_rng(action_seed)
    pos = lambda e, d: np.array([d[f"agent_{i}"] for i in range(N)], np.int16)
    out = {
        "config": np.array(json.dumps(cfg)), "grids": np.stack([e.grid for e in envs]).astype(np.uint8),
        "rng_words": np.stack(words), "seeds": np.asarray(seeds, np.int64),
        "ctor_starts": np.stack([pos(e, e.starts) for e in envs]), "ctor_goals": np.stack([pos(e, e.goals) for e in envs]),
        "actions": np.zeros((T, B, N), np.int8), "obs": np.zeros((T, B, L), np.float32),
        "reward": np.zeros((T, B), np.float64), "terminated": np.zeros((T, B), np.uint8),
        "truncated": np.zeros((T, B), np.uint8), "info": np.zeros((T, B, 4), np.float32),
        "positions": np.zeros((T, B, N, 2), np.int16), "did_reset": np.zeros((T, B), np.uint8),
        "reset_obs": np.zeros((T, B, L), np.float32), "reset0_obs": np.zeros((B, L), np.float32),
        "reset0_positions": np.zeros((B, N, 2), np.int16), "reset0_goals": np.zeros((B, N, 2), np.int16),
        "final_rng_words": np.zeros((B, 6), np.uint64),
    }
    for b, e in enumerate(envs):
        o, info = e.reset()
        out["reset0_obs"][b] = o
        out["reset0_positions"][b] = pos(e, e.positions)
        out["reset0_goals"][b] = pos(e, e.goals)
        assert np.array_equal(info["action_mask"].astype(np.float32), o[-5 * N:])
    for t in range(T):
        for b, e in enumerate(envs):
            acts = rng.integers(0, 5, size=N)
            for a in range(N):
                if rng.random() < greedy:
                    d = np.asarray(e.goals[f"agent_{a}"], int) - np.asarray(e.positions[f"agent_{a}"], int)
                    if abs(d[0]) >= abs(d[1]) and d[0] != 0:
                        acts[a] = 3 if d[0] > 0 else 1
                    elif d[1] != 0:
                        acts[a] = 2 if d[1] > 0 else 4
            out["actions"][t, b] = acts
            o, r, term, trunc, info = e.step([int(x) for x in acts])
            out["obs"][t, b] = o
            out["reward"][t, b] = r
            out["terminated"][t, b], out["truncated"][t, b] = term, trunc
            out["info"][t, b] = [info["blocking_count_step"], info["goals_reached_step"], info["goals_reached_total"],
                                 info["blocking_count_total"]]
            out["positions"][t, b] = pos(e, e.positions)
            if term or trunc:
                o, _ = e.reset()
                out["did_reset"][t, b] = 1
                out["reset_obs"][t, b] = o
    for b, e in enumerate(envs):
        out["final_rng_words"][b] = pcg_words(e.rng.bit_generator.state)
    return out


def gs_single_agent_traces():
    cfg = {"env_name": "ReferenceModel-2-1", "num_agents": 4, "steps_per_episode": 60, "deterministic": True}
    tr = record_cte_trace(cfg, None, [123], 250, greedy=0.7)
    save("gs_cte_named_2_1_det", tr)
    cfg = {"env_name": "ReferenceModel-2-2", "num_agents": 4, "steps_per_episode": 50}
    save("gs_cte_named_2_2", record_cte_trace(cfg, None, [5, 6], 220, greedy=0.8))
    cfg = {"env_name": "synthetic", "num_agents": 5, "steps_per_episode": 40, "blocking_penalty": -0.3,
           "move_after_goal_penalty": -0.07}
    grids = [synth_grid(70_000 + b, 6, 7, 0.20, 10) for b in range(6)]
    tr = record_cte_trace(cfg, grids, list(range(6)), 400, greedy=0.8)
    succ = int((tr["terminated"].astype(bool) & ~tr["truncated"].astype(bool)).sum())
    print(f"  gs small: blocking={tr['info'][:, :, 0].sum():.0f} goals={tr['info'][:, :, 1].sum():.0f} successes={succ}")
    assert tr["info"][:, :, 0].sum() >= 10 and succ >= 1
    save("gs_cte_6x7_n5_penalties", tr)
    cfg = {"env_name": "synthetic", "num_agents": 8, "steps_per_episode": 100}
    grids = [synth_grid(10_000 + b, 16, 16, 0.20, 16) for b in range(4)]
    save("gs_cte_16x16_n8", record_cte_trace(cfg, grids, list(range(4)), 250, greedy=0.3))


def export_named_grids():
    """Named-grid data (get_grid.py:17-857 tables) -> package data file, via the reference's own accessors."""
    _, gg = rh.load_reference()
    data = {}
    for name in NAMED:
        data[name + ".grid"] = gg.get_grid(name).astype(np.uint8)
        for n in (4, 3, 2, 1):
            try:
                s = gg.get_start_positions(name, n)
                g = gg.get_goal_positions(name, n)
            except ValueError:
                continue
            data[name + ".starts"] = np.array([s[f"agent_{i}"] for i in range(n)], np.int16)
            data[name + ".goals"] = np.array([g[f"agent_{i}"] for i in range(n)], np.int16)
            break
    path = NAMED_GRIDS
    os.makedirs(os.path.dirname(path), exist_ok=True)
    np.savez_compressed(path, **data)
    print(f"  named_grids.npz {os.path.getsize(path) / 1024:.1f} KiB")


def main():
    os.makedirs(GOLDEN, exist_ok=True)
    print("recording golden vectors from the unmodified reference:")
    export_named_grids()
    g1_parity_digest()
    g2_g3_g4_batches()
    g10_wide_groups()
    g11_exactly_2n_free_cells()
    g12_lifelong_exactly_2n_free_cells()
    g5_named_and_deterministic()
    g5_micro_cases()
    g5_error_paths()
    g6_rng_known_answers()
    g9_benchmark_pin()
    gs_single_agent_traces()
    sizes = sum(os.path.getsize(os.path.join(GOLDEN, f)) for f in os.listdir(GOLDEN))
    print(f"total golden size: {sizes / 1024:.0f} KiB")


def check() -> int:
    """Regenerate everything into a temporary directory and compare, array by array, with the committed fixtures
    (the .npz containers themselves differ: zip members carry timestamps)."""
    import tempfile

    global GOLDEN, NAMED_GRIDS
    committed, committed_named = GOLDEN, NAMED_GRIDS
    bad = []
    with tempfile.TemporaryDirectory() as tmp:
        GOLDEN, NAMED_GRIDS = os.path.join(tmp, "golden"), os.path.join(tmp, "named_grids.npz")
        main()
        pairs = [(os.path.join(GOLDEN, f), os.path.join(committed, f)) for f in sorted(os.listdir(GOLDEN))]
        pairs.append((NAMED_GRIDS, committed_named))
        extra = sorted(set(os.listdir(committed)) - set(os.listdir(GOLDEN)))
        bad += [f"{f}: committed but not generated" for f in extra if f.endswith(".npz")]
        for new, old in pairs:
            name = os.path.basename(new)
            if not os.path.exists(old):
                bad.append(f"{name}: generated but not committed")
                continue
            with np.load(new, allow_pickle=False) as a, np.load(old, allow_pickle=False) as b:
                if sorted(a.files) != sorted(b.files):
                    bad.append(f"{name}: different arrays {sorted(set(a.files) ^ set(b.files))[:6]}")
                    continue
                for k in a.files:
                    if a[k].dtype != b[k].dtype or a[k].shape != b[k].shape or not np.array_equal(a[k], b[k]):
                        bad.append(f"{name}: array {k} differs")
    GOLDEN, NAMED_GRIDS = committed, committed_named
    for line in bad:
        print("MISMATCH", line)
    print("golden fixtures reproduce exactly" if not bad else f"{len(bad)} mismatches")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(check() if "--check" in sys.argv else main())
