"""GPU parity tests proper: the HIP engine, called through the C ABI, must be bit-exact against
  (1) the golden traces recorded from the unmodified reference (tests/golden/),
  (2) the reference's own published SHA-256 parity digests,
  (3) the CPU oracle on seeded inputs at BASELINE.json's sizes (c2, c3, c5 shapes).
Run with `pytest -m gpu` on an MI355X.
"""

from __future__ import annotations

import numpy as np
import pytest

from test_oracle_golden import REF_DIGEST, REF_SUMMARY, run_parity_digest
from trace_util import (
    BATCH_FIXTURES, MICRO_CASES, EngineStepper, OracleStepper, compare_steppers, load_golden, replay_batch_trace,
    replay_micro_case, synth_grids,
)

pytestmark = pytest.mark.gpu


class _EngineSingle:
    """Single-env view used by the parity-digest protocol."""

    def __init__(self, grid, cfg, rng_words, fixed_starts, fixed_goals):
        self.st = EngineStepper(grid[None], cfg, rng_words=rng_words[None], fixed_starts=fixed_starts[None],
                                fixed_goals=fixed_goals[None])

    positions = property(lambda self: self.st.positions()[0])
    goals = property(lambda self: self.st.goals()[0])

    def reset(self):
        return self.st.reset()[0].copy()

    def step(self, acts):
        o = self.st.step(np.asarray(acts, np.int8)[None], auto_reset=False)
        return (o["obs"][0], o["rewards"][0], bool(o["terminated"][0]), bool(o["truncated"][0]), o["info_all"][0],
                o["info_agent"][0])


@pytest.mark.parametrize("kind", ["stochastic", "deterministic"])
def test_engine_reproduces_reference_parity_digest(kind):
    digest, summary = run_parity_digest(_EngineSingle, kind)
    assert digest == REF_DIGEST[kind]
    assert summary == REF_SUMMARY[kind]


@pytest.mark.parametrize("name", BATCH_FIXTURES)
def test_engine_matches_golden_trace(name):
    fx = load_golden(name)
    stats = replay_batch_trace(EngineStepper, fx)
    assert stats["steps"] == fx["actions"].shape[0]


@pytest.mark.parametrize("name", MICRO_CASES)
def test_engine_micro_cases(name):
    replay_micro_case(EngineStepper, load_golden("g5_micro_cases"), name)


@pytest.mark.parametrize("lanes", [8, 16, 32, 64])
def test_engine_golden_trace_with_wider_groups(lanes):
    """Same env, more lanes per env than needed: results must not depend on the group width."""
    fx = load_golden("g3b_tight_6x7_n6")
    replay_batch_trace(lambda *a, **k: EngineStepper(*a, lanes_per_env=lanes, **k), fx)
    fx = load_golden("g4b_lifelong_5x9_n10")
    if lanes >= 16:
        replay_batch_trace(lambda *a, **k: EngineStepper(*a, lanes_per_env=lanes, **k), fx, steps=200, check_rng=False)


# ---- engine vs oracle at the BASELINE.json shapes ------------------------------------------------
def _vs_oracle(B, H, W, N, density, steps, cfg_extra=None, greedy=False):
    cfg = {"env_name": "synthetic", "num_agents": N, "sensor_range": 2, "steps_per_episode": 100,
           "include_action_mask_in_obs": True}
    cfg.update(cfg_extra or {})
    grids = synth_grids(B, H, W, density, N)
    seeds = list(range(B))
    rng = np.random.default_rng(999)
    if greedy:
        acts = rng.choice(5, size=(steps, B, N), p=[0.1, 0.1, 0.3, 0.4, 0.1]).astype(np.int8)
    else:
        acts = rng.integers(0, 5, size=(steps, B, N)).astype(np.int8)
    return compare_steppers(EngineStepper(grids, cfg, seeds=seeds), OracleStepper(grids, cfg, seeds=seeds), acts,
                            check_state_every=25)


def test_specialized_and_generic_kernels_are_both_exercised():
    """The BASELINE shapes pick a compile-time specialised step kernel; force_generic_kernel must disable it."""
    from dl_reference_models_amd import workloads as wl
    from dl_reference_models_amd.vec_env import VecReferenceModel

    for name, want in (("c3_8192x32x32_n8", 1), ("c2_1024x16x16_n4", 2), ("c5_1024x64x64_n64_lifelong", 3)):
        cfg = wl.workload_config(name, list(range(4)))
        assert VecReferenceModel(cfg).launch_info()["specialized_kernel"] == want
        assert VecReferenceModel(dict(cfg, force_generic_kernel=True)).launch_info()["specialized_kernel"] == 0


@pytest.mark.parametrize("shape", [(512, 16, 16, 4, 0.20, {}), (1024, 32, 32, 8, 0.40, {}),
                                   (64, 64, 64, 64, 0.20, {"lifelong_mapf": True, "steps_per_episode": 256}),
                                   (512, 32, 32, 8, 0.40, {"include_action_mask_in_obs": False})])
def test_generic_kernel_on_baseline_shapes(shape):
    """Same shapes as the specialised kernels, run through the runtime-config kernel."""
    B, H, W, N, density, extra = shape
    _vs_oracle(B, H, W, N, density, 260, dict(extra, force_generic_kernel=True))


def test_engine_vs_oracle_reference_default_obs_layout():
    """Reference-default observation (mask off, L = 28): specialisations 4 and 5."""
    _vs_oracle(1024, 32, 32, 8, 0.40, 220, {"include_action_mask_in_obs": False})
    _vs_oracle(512, 16, 16, 4, 0.20, 220, {"include_action_mask_in_obs": False})


def test_engine_vs_oracle_reference_training_setup():
    """The reference's own training configuration (main.py:55-67: 16 agents, sensor_range 3, mask off): specialisation 6,
    16-lane groups with the sampler workgroups in front of the grid; also through the runtime-config kernel."""
    extra = {"sensor_range": 3, "include_action_mask_in_obs": False, "steps_per_episode": 64}
    stats = _vs_oracle(520, 32, 32, 16, 0.20, 200, extra)
    assert stats["episodes"] >= 3 * 520
    _vs_oracle(130, 32, 32, 16, 0.20, 150, dict(extra, force_generic_kernel=True))


def test_engine_vs_oracle_c2_1024x16x16x4():
    """BASELINE config 2: 1024 vectorized 16x16 grids, 4 agents, bit-exact check vs CPU."""
    stats = _vs_oracle(1024, 16, 16, 4, 0.20, 320)
    assert stats["episodes"] >= 3 * 1024 and stats["livelock_events"] > 0


def test_engine_vs_oracle_c3_8192x32x32x8():
    """BASELINE config 3 shape (the bench workload): 8192 envs, 32x32, 8 agents, density 0.40."""
    stats = _vs_oracle(8192, 32, 32, 8, 0.40, 210)
    assert stats["episodes"] >= 2 * 8192 and stats["deadlock_events"] > 0 and stats["livelock_events"] > 0


def test_engine_vs_oracle_c3_biased_actions_ragged_batch():
    """Biased action stream (more deadlocks / blocking) on a batch that does not fill the last wave."""
    stats = _vs_oracle(1003, 32, 32, 8, 0.40, 230, greedy=True)
    assert stats["deadlock_events"] > 0


def test_engine_vs_oracle_c5_64x64x64_lifelong():
    """BASELINE config 5: 64x64 grid, 64 agents, lifelong goal respawn (one wavefront per env)."""
    stats = _vs_oracle(256, 64, 64, 64, 0.20, 300, {"lifelong_mapf": True, "steps_per_episode": 256})
    assert stats["goals"] > 0 and stats["episodes"] >= 256


@pytest.mark.parametrize("extra", [{"force_pair_walk": True}, {"lock_nearby_manhattan": 5, "lock_min_neighbors": 2},
                                   {"lock_nearby_manhattan": 9}, {"sensor_range": 5, "include_goal_distance": True}])
def test_wide_group_paths_cell_map_and_pair_walk(extra):
    """N > 16: the LDS cell-map path (default) and the all-pairs walk (forced, or when the lock radius exceeds the
    map border) must both match the oracle."""
    _vs_oracle(48, 40, 37, 33, 0.15, 200, dict({"lifelong_mapf": True, "steps_per_episode": 70}, **extra))
    _vs_oracle(24, 64, 64, 64, 0.20, 150, dict({"steps_per_episode": 60}, **extra))


def test_engine_vs_oracle_odd_shapes():
    for (B, H, W, N, sr, extra) in [
        (37, 7, 9, 12, 3, {"lifelong_mapf": True, "steps_per_episode": 50, "deadlock_window_steps": 3,
                           "livelock_window_steps": 5, "lock_nearby_manhattan": 3, "lock_min_neighbors": 2,
                           "lock_progress_epsilon": 0.5, "include_goal_distance": True}),
        (65, 3, 64, 20, 1, {"steps_per_episode": 40, "normalize_goal_delta": False}),
        (19, 64, 2, 33, 5, {"steps_per_episode": 30, "include_blocking_pressure_in_obs": False}),
        (130, 5, 5, 3, 0, {"steps_per_episode": 17, "enable_lock_metrics": False}),
        (9, 12, 12, 1, 4, {"steps_per_episode": 25, "lifelong_mapf": True}),
        (50, 9, 9, 17, 2, {"deadlock_window_steps": 64, "livelock_window_steps": 64, "steps_per_episode": 150}),
    ]:
        cfg = {"env_name": "synthetic", "num_agents": N, "sensor_range": sr, "include_action_mask_in_obs": True}
        cfg.update(extra)
        grids = synth_grids(B, H, W, 0.15, N, base_seed=70_000)
        acts = np.random.default_rng(5).integers(0, 5, size=(180, B, N)).astype(np.int8)
        seeds = list(range(300, 300 + B))
        compare_steppers(EngineStepper(grids, cfg, seeds=seeds), OracleStepper(grids, cfg, seeds=seeds), acts)


# ---- error paths --------------------------------------------------------------------------------
def test_engine_bad_action_raises_value_error_with_partial_mutation():
    fx = load_golden("g5_bad_action")
    st = EngineStepper(fx["grid"][None], fx["config"], rng_words=fx["rng_words"][None])
    st.reset()
    assert np.array_equal(st.positions()[0], fx["positions0"])
    st.step(fx["actions"][None], auto_reset=False)
    with pytest.raises(ValueError, match="Invalid action 7 for agent_1"):
        st.env.poll_error()
    assert np.array_equal(st.positions()[0], fx["positions_after"])
    assert int(st.env.get_state()["counters"][0, 0]) == int(fx["step_count_after"])
    st.env.poll_error()  # cleared


def test_engine_too_few_free_cells_raises():
    from dl_reference_models_amd.vec_env import VecReferenceModel

    grid = np.ones((3, 3), np.uint8)
    grid[0, :] = 0
    with pytest.raises(ValueError, match="only 3 free cells"):
        VecReferenceModel({"grid": grid, "num_agents": 2, "seed": 0})


# NOTE: the reference's RuntimeError "No valid cell available for lifelong goal reassignment" (MA-env:296-298)
# is unreachable: the ctor demands F >= 2N free cells while a respawn excludes at most 2N-1 of them.  The engine
# keeps the check (MAPF_ERR_NO_RESPAWN) but no input can trigger it, so there is nothing to test.


def test_goal_delta_division_is_correctly_rounded_for_every_delta():
    """fp32 goal_delta = int / (dim-1) must equal NumPy's float32 division for every possible operand."""
    import torch

    from dl_reference_models_amd.vec_env import VecReferenceModel

    for dim in (2, 3, 7, 20, 31, 33, 64):
        grid = np.zeros((dim, dim), np.uint8)
        env = VecReferenceModel({"grid": grid, "num_envs": dim, "num_agents": 2, "seed": 0, "sensor_range": 0,
                                 "include_blocking_pressure_in_obs": False})
        env.reset()
        pos = np.zeros((dim, 2, 2), np.int16)
        goals = np.zeros((dim, 2, 2), np.int16)
        for b in range(dim):  # agent 0 at (0, b) with goal (b, 0): deltas (+b, -b); agent 1 mirrored
            pos[b, 0] = (0, b)
            goals[b, 0] = (b, 0)
            pos[b, 1] = (dim - 1, dim - 1 - b)
            goals[b, 1] = (dim - 1 - b, dim - 1)
        pos[0, 1], goals[0, 1] = (dim - 1, dim - 1), (dim - 1, dim - 2) if dim > 1 else (0, 0)
        goals[0, 0] = (0, 1)
        env.set_state(positions=pos, goals=goals, starts=pos, clear_episode=True)
        out = env.step(torch.zeros((dim, 2), dtype=torch.int8, device=env.device), auto_reset=False)
        obs = out["obs"].cpu().numpy()
        den = np.float32(max(dim - 1, 1))
        for b in range(dim):
            for a in range(2):
                want = (goals[b, a] - pos[b, a]).astype(np.float32) / den
                assert np.array_equal(obs[b, a, 1:3], want.astype(np.float32)), (dim, b, a)


def test_state_roundtrip_and_rng_words():
    st = EngineStepper(synth_grids(5, 8, 8, 0.1, 3), {"num_agents": 3, "sensor_range": 1}, seeds=[1, 2, 3, 4, 5])
    st.reset()
    s0 = st.env.get_state()
    st.env.set_state(**{k: s0[k] for k in ("positions", "goals", "starts", "reached", "completed_once", "pressure_prev",
                                            "counters", "rng_words", "lock_history", "distance_ring")})
    s1 = st.env.get_state()
    for k in s0:
        assert np.array_equal(s0[k], s1[k]), k


@pytest.mark.parametrize("shape", [(16, 16, 16, 4), (24, 32, 32, 8), (4, 12, 12, 20)])
def test_sequential_and_parallel_reset_samplers_agree_with_the_oracle(shape):
    """The in-kernel reset has two restatements of rng.choice(F, 2N, replace=False): the lane-parallel one (PCG64
    jump-ahead, all draws at once) and the sequential one it falls back to after a Lemire rejection (probability
    ~1e-7 per draw, so never seen in a test run).  `force_sequential_reset` takes the fallback on every reset; both
    must reproduce the oracle across many episode boundaries, RNG words included."""
    b, h, w, n = shape
    cfg = {"num_agents": n, "sensor_range": 2, "steps_per_episode": 9, "include_action_mask_in_obs": True}
    grids = synth_grids(b, h, w, 0.2, n)
    seeds = list(range(900, 900 + b))
    acts = np.random.default_rng(12).integers(0, 5, size=(60, b, n)).astype(np.int8)
    for knob in (False, True):
        eng = EngineStepper(grids, cfg, seeds=seeds, force_sequential_reset=knob)
        stats = compare_steppers(eng, OracleStepper(grids, cfg, seeds=seeds), acts)
        assert stats["episodes"] >= 6 * b
    if (h, w, n) == (32, 32, 8):
        assert eng.env.launch_info()["specialized_kernel"] == 1  # the knob does not change the kernel choice


@pytest.mark.parametrize("n", [3, 8])
def test_reset_with_exactly_2n_free_cells_takes_the_sequential_sampler(n):
    """F == 2N: the first Floyd draw has bound 0 and consumes no random number (NumPy returns the offset without
    drawing), so the in-kernel reset leaves its parallel sampler for the sequential restatement.  Several resets in
    a row against the oracle pin the stream alignment."""
    h, w = 4, 2 * n  # 8N cells, 2N of them free
    grids = np.ones((6, h, w), np.uint8)
    rng = np.random.default_rng(8)
    for b in range(grids.shape[0]):
        free = rng.choice(h * w, size=2 * n, replace=False)
        grids[b].reshape(-1)[free] = 0
    cfg = {"num_agents": n, "sensor_range": 1, "steps_per_episode": 7, "include_action_mask_in_obs": True}
    seeds = list(range(700, 706))
    acts = np.random.default_rng(9).integers(0, 5, size=(40, grids.shape[0], n)).astype(np.int8)
    stats = compare_steppers(EngineStepper(grids, cfg, seeds=seeds), OracleStepper(grids, cfg, seeds=seeds), acts)
    assert stats["episodes"] >= 5 * grids.shape[0]


@pytest.mark.parametrize("shape", [(24, 12, 12, 8, {}), (5, 9, 14, 3, {"lifelong_mapf": True}), (3, 20, 20, 20, {})])
def test_step_with_every_output_null_advances_the_state_identically(shape):
    """The C ABI allows obs / rewards / flags / info to be NULL (mapf_step.h): the step kernel then runs without
    its observation wave.  The state after T such steps must equal the state after T ordinary steps (and so must the
    static-state observation computed from it afterwards)."""
    import ctypes as C
    import torch

    b, h, w, n, extra = shape
    cfg = {"num_agents": n, "sensor_range": 2, "steps_per_episode": 17, "include_action_mask_in_obs": True, **extra}
    grids = synth_grids(b, h, w, 0.15, n)
    full = EngineStepper(grids, cfg, seeds=list(range(50, 50 + b)))
    bare = EngineStepper(grids, cfg, seeds=list(range(50, 50 + b)))
    full.reset()
    bare.reset()
    acts = np.random.default_rng(4).integers(0, 5, size=(40, b, n)).astype(np.int8)
    env = bare.env
    for t in range(acts.shape[0]):
        out = full.env.step(torch.from_numpy(acts[t]).to(full.env.device), auto_reset=False)
        done = (out["terminated"] | out["truncated"]).bool()
        a = torch.from_numpy(acts[t]).to(env.device)
        rc = env._lib.mapf_step(env._h, C.c_void_p(a.data_ptr()), None, None, None, None, None, None, None, 0, env._stream())
        assert rc == 0
        if bool(done.any()):  # finished envs are reset explicitly in both engines (same RNG stream)
            full.env.reset(done.to(torch.uint8))
            env.reset(done.to(torch.uint8))
    env.poll_error()
    sa, sb = full.env.get_state(), env.get_state()
    for k in sa:
        assert np.array_equal(sa[k], sb[k]), k
    assert np.array_equal(env.observe().cpu().numpy(), full.env.observe().cpu().numpy())


@pytest.mark.parametrize("extra", [{}, {"livelock_window_steps": 40, "deadlock_window_steps": 20},
                                   {"lifelong_mapf": True, "livelock_window_steps": 5, "deadlock_window_steps": 3}])
def test_state_snapshot_resumes_bit_exactly_in_a_fresh_engine(extra):
    """get_state / set_state is a checkpoint: a fresh engine loaded with a mid-episode snapshot (positions, flags,
    counters, RNG words, lock history, distance ring) continues exactly like the original."""
    cfg = {"num_agents": 6, "sensor_range": 2, "steps_per_episode": 90, "include_action_mask_in_obs": True}
    cfg.update(extra)
    grids = synth_grids(40, 12, 12, 0.25, 6, base_seed=90_000)
    acts = np.random.default_rng(3).choice(5, size=(160, 40, 6), p=[0.1, 0.1, 0.5, 0.2, 0.1]).astype(np.int8)
    a = EngineStepper(grids, cfg, seeds=list(range(40)))
    a.reset()
    for t in range(57):
        a.step(acts[t])
    snap = a.env.get_state()
    b = EngineStepper(grids, cfg, seeds=list(range(1000, 1040)))  # different RNG streams until the snapshot lands
    b.reset()
    b.env.set_state(**snap)
    s2 = b.env.get_state()
    for k in snap:
        assert np.array_equal(snap[k], s2[k]), k
    for t in range(57, 160):
        ra, rb = a.step(acts[t]), b.step(acts[t])
        for k in ("obs", "rewards", "terminated", "truncated", "info_all", "info_agent"):
            assert np.array_equal(ra[k], rb[k]), (k, t)
    assert np.array_equal(a.rng_words(), b.rng_words())


@pytest.mark.parametrize("shape", [
    (512, 32, 32, 8, 0.40, {}),                                                     # specialised (c3 shape)
    (300, 16, 16, 4, 0.20, {"include_goal_distance": True}),                        # runtime-config kernel, ragged last wave
    (32, 64, 64, 64, 0.20, {"lifelong_mapf": True, "steps_per_episode": 60}),       # lifelong, respawns + resets inside the loop
    (40, 9, 9, 5, 0.15, {"livelock_window_steps": 30, "deadlock_window_steps": 20, "steps_per_episode": 45}),  # int16 ring path
    (200, 24, 24, 16, 0.20, {"sensor_range": 3, "include_action_mask_in_obs": False, "steps_per_episode": 40}),  # specialisation 6
])
def test_step_many_equals_repeated_single_steps(shape):
    """mapf_step_many(T) must produce, step for step, what T mapf_step launches produce (and the same final state)."""
    import torch

    B, H, W, N, density, extra = shape
    cfg = {"env_name": "synthetic", "num_agents": N, "sensor_range": 2, "steps_per_episode": 50,
           "include_action_mask_in_obs": True}
    cfg.update(extra)
    grids = synth_grids(B, H, W, density, N, base_seed=120_000)
    seeds = list(range(B))
    T = 130
    acts = np.random.default_rng(11).integers(0, 5, size=(T, B, N)).astype(np.int8)
    one = EngineStepper(grids, cfg, seeds=seeds)
    many = EngineStepper(grids, cfg, seeds=seeds)
    one.reset()
    many.reset()
    fused = many.env.step_many(torch.from_numpy(acts).to(many.env.device), obs_mode=2)
    fused = {k: v.cpu().numpy() for k, v in fused.items()}
    for t in range(T):
        o = one.step(acts[t], auto_reset=True)
        for k in ("obs", "rewards", "terminated", "truncated", "info_all", "info_agent"):
            assert np.array_equal(o[k], fused[k][t]), (k, t)
    sa, sb = one.env.get_state(), many.env.get_state()
    for k in sa:
        assert np.array_equal(sa[k], sb[k]), k
    # obs_mode 1 returns only the last observation; obs_mode 0 none
    third = EngineStepper(grids, cfg, seeds=seeds)
    third.reset()
    last = third.env.step_many(torch.from_numpy(acts).to(third.env.device), obs_mode=1, outputs=False)
    assert np.array_equal(last["obs"].cpu().numpy(), fused["obs"][-1]) and last["rewards"] is None


@pytest.mark.parametrize("lifelong", [False, True])
def test_episode_metrics_match_what_the_callbacks_would_log(lifelong):
    """Device-side episode accumulators vs the same sums taken from the ORACLE's outputs at every episode end,
    using the definitions of the reference's callbacks (src/trainers/callbacks.py:138-181, :236-335)."""
    from dl_reference_models_amd._lib import (ACC_BLOCKING_COUNT, ACC_COMPLETED_AGENTS, ACC_DEADLOCK_COUNT, ACC_DEADLOCK_STEPS,
                                              ACC_EPISODES, ACC_EPISODE_STEPS, ACC_GOALS_REACHED, ACC_LIVELOCK_COUNT,
                                              ACC_LIVELOCK_STEPS, ACC_SUCCESSES)

    B, N = 96, 4
    cfg = {"env_name": "synthetic", "num_agents": N, "sensor_range": 1, "steps_per_episode": 35, "lifelong_mapf": lifelong}
    grids = synth_grids(B, 6, 6, 0.10, N, base_seed=130_000)
    seeds = list(range(B))
    eng, orc_ = EngineStepper(grids, cfg, seeds=seeds), OracleStepper(grids, cfg, seeds=seeds)
    eng.reset()
    orc_.reset()
    rng = np.random.default_rng(21)
    want = np.zeros(12, dtype=np.int64)
    steps_in_ep = np.zeros(B, dtype=np.int64)
    for t in range(200):
        # mostly greedy actions so that finite episodes also END IN SUCCESS sometimes
        pos, goals = orc_.positions().astype(int), orc_.goals().astype(int)
        d = goals - pos
        greedy = np.where(np.abs(d[..., 0]) >= np.abs(d[..., 1]), np.where(d[..., 0] > 0, 3, np.where(d[..., 0] < 0, 1, 0)),
                          np.where(d[..., 1] > 0, 2, 4))
        acts = np.where(rng.random((B, N)) < 0.8, greedy, rng.integers(0, 5, size=(B, N))).astype(np.int8)
        ro, re = orc_.step(acts), eng.step(acts)
        assert np.array_equal(ro["info_all"], re["info_all"]) and np.array_equal(ro["terminated"], re["terminated"])
        steps_in_ep += 1
        done = (ro["terminated"] | ro["truncated"]).astype(bool)
        ia = ro["info_all"][done]
        want[ACC_EPISODES] += done.sum()
        want[ACC_SUCCESSES] += (ro["terminated"].astype(bool) & ~ro["truncated"].astype(bool)).sum()
        # at episode end info_all holds the episode totals the callbacks read from the env attributes;
        # _episode_goals_reached_total equals goals_reached_total in both modes (first arrivals are counted once)
        want[ACC_GOALS_REACHED] += int(ia[:, 1].sum())
        want[ACC_BLOCKING_COUNT] += int(ia[:, 3].sum())
        want[ACC_DEADLOCK_COUNT] += int(ia[:, 8].sum())
        want[ACC_LIVELOCK_COUNT] += int(ia[:, 9].sum())
        want[ACC_DEADLOCK_STEPS] += int(ia[:, 10].sum())
        want[ACC_LIVELOCK_STEPS] += int(ia[:, 11].sum())
        want[ACC_COMPLETED_AGENTS] += int(np.rint(ia[:, 12] * N).sum())
        want[ACC_EPISODE_STEPS] += int(steps_in_ep[done].sum())
        steps_in_ep[done] = 0
    got = eng.env.episode_sums()
    assert np.array_equal(got, want), (got, want)
    assert want[ACC_EPISODES] >= 5 * B
    m = eng.env.episode_metrics()
    assert m["episodes"] == want[ACC_EPISODES] and m["goals_reached"] == want[ACC_GOALS_REACHED] / want[ACC_EPISODES]
    if lifelong:
        assert want[ACC_SUCCESSES] == 0 and m["episode_len_mean"] == 35.0
        assert m["success_rate"] == m["completion_ratio"] == want[ACC_COMPLETED_AGENTS] / (want[ACC_EPISODES] * N)
        assert m["throughput"] == pytest.approx(want[ACC_GOALS_REACHED] / want[ACC_EPISODE_STEPS])
    else:
        assert want[ACC_SUCCESSES] > 0 and m["success_rate"] == want[ACC_SUCCESSES] / want[ACC_EPISODES]
        assert "throughput" not in m
    assert np.array_equal(eng.env.episode_sums(reset=True), want) and eng.env.episode_sums().sum() == 0


def test_randomized_configuration_fuzz_engine_vs_oracle():
    """Random shapes / flags / windows / episode lengths / group widths: every combination must stay bit-exact.
    (Seeded: the same 40 configurations every run.)"""
    rng = np.random.default_rng(20260101)
    for case in range(40):
        H, W = int(rng.integers(1, 65)), int(rng.integers(2, 65))
        N = int(rng.integers(1, min(64, max(1, (H * W) // 3)) + 1))
        sr = int(rng.integers(0, 6))
        cfg = {
            "env_name": "synthetic", "num_agents": N, "sensor_range": sr,
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
        p = rng.dirichlet(np.ones(5))
        acts = rng.choice(5, size=(90, B, N), p=p).astype(np.int8)
        try:
            compare_steppers(EngineStepper(grids, cfg, seeds=seeds, **extra), OracleStepper(grids, cfg, seeds=seeds), acts)
        except AssertionError as exc:
            raise AssertionError(f"fuzz case {case}: cfg={cfg} B={B} HxW={H}x{W} density={density} extra={extra}: {exc}") from exc


# ---- episode boundaries: pre-drawn placements, fast / slow reset paths ---------------------------------------
@pytest.mark.parametrize("want_final", [False, True])
@pytest.mark.parametrize("shape", [
    (256, 32, 32, 8, 0.40, {}),                                                  # c3 shape: full waves, specialised kernel
    (256, 32, 32, 8, 0.40, {"force_generic_kernel": True}),
    (100, 16, 16, 4, 0.20, {"steps_per_episode": 37}),                           # c2 shape, ragged last wave
    (50, 9, 9, 3, 0.15, {"steps_per_episode": 23, "include_goal_distance": True}),  # N < lanes per env
    (40, 10, 10, 6, 0.10, {"steps_per_episode": 3}),                             # episodes shorter than the sampler's lead
    (40, 10, 10, 6, 0.10, {"steps_per_episode": 1}),
    (30, 12, 12, 20, 0.20, {"steps_per_episode": 19}),                           # wide groups: LDS cell map, slow reset + slot
    (64, 8, 8, 5, 0.10, {"steps_per_episode": 11, "force_sequential_reset": True}),  # sampler never succeeds
    (48, 8, 8, 5, 0.10, {"steps_per_episode": 11, "lanes_per_env": 16}),
])
def test_staggered_episode_boundaries_match_the_oracle(shape, want_final):
    """Envs finish in different steps (staggered phases + short episodes), so every launch mixes envs that are
    re-placed from a pre-drawn slot, envs that draw inline and envs that do not reset; with and without the
    terminal observation (final_obs) being asked for.  RNG words are compared at the end."""
    B, H, W, N, density, extra = shape
    cfg = {"env_name": "synthetic", "num_agents": N, "sensor_range": 2, "steps_per_episode": 29,
           "include_action_mask_in_obs": True}
    cfg.update(extra)
    grids = synth_grids(B, H, W, density, N, base_seed=310_000)
    seeds = list(range(500, 500 + B))
    acts = np.random.default_rng(77).integers(0, 5, size=(150, B, N)).astype(np.int8)
    spe = cfg["steps_per_episode"]
    stats = compare_steppers(EngineStepper(grids, cfg, seeds=seeds, want_final_obs=want_final),
                             OracleStepper(grids, cfg, seeds=seeds), acts, check_state_every=7,
                             step_counts=np.arange(B) % spe)
    assert stats["episodes"] >= B * (150 // spe - 1)


def test_goal_seeking_policy_ends_episodes_by_success_at_any_step():
    """Success terminations (every agent on its goal) arrive at arbitrary steps, including the first steps of an
    episode when the background sampler has not yet provided a placement: greedy actions on open grids."""
    B, H, W, N = 96, 7, 7, 2
    cfg = {"env_name": "synthetic", "num_agents": N, "sensor_range": 1, "steps_per_episode": 40}
    grids = synth_grids(B, H, W, 0.0, N, base_seed=77_000)
    seeds = list(range(B))
    eng, orc = EngineStepper(grids, cfg, seeds=seeds, want_final_obs=False), OracleStepper(grids, cfg, seeds=seeds)
    _ = eng.reset(), orc.reset()
    rng = np.random.default_rng(5)
    successes = 0
    for t in range(220):
        pos, goal = orc.positions().astype(int), orc.goals().astype(int)
        d = goal - pos
        # step along the larger-magnitude axis towards the goal (UP 1, RIGHT 2, DOWN 3, LEFT 4), a little noise
        vert = np.where(d[..., 0] < 0, 1, 3)
        horz = np.where(d[..., 1] > 0, 2, 4)
        a = np.where(np.abs(d[..., 0]) >= np.abs(d[..., 1]), vert, horz)
        a = np.where((d == 0).all(-1), 0, a)
        a = np.where(rng.random(a.shape) < 0.1, rng.integers(0, 5, a.shape), a).astype(np.int8)
        ra, rb = eng.step(a), orc.step(a)
        for k in ("obs", "rewards", "terminated", "truncated", "info_all", "info_agent"):
            assert np.array_equal(ra[k], rb[k]), (k, t)
        successes += int((rb["terminated"].astype(bool) & ~rb["truncated"].astype(bool)).sum())
        assert np.array_equal(eng.positions(), orc.positions()) and np.array_equal(eng.goals(), orc.goals()), t
    assert successes > 5 * B
    assert np.array_equal(eng.rng_words(), orc.rng_words())


# ---- device-side action source of the fused kernel --------------------------------------------------------------
@pytest.mark.parametrize("shape", [
    (512, 32, 32, 8, 0.40, {"steps_per_episode": 40}),                                # specialised, two-wave fused kernel
    (300, 16, 16, 4, 0.20, {"include_goal_distance": True, "steps_per_episode": 25}),  # runtime-config kernel, ragged wave
    (50, 9, 9, 5, 0.15, {"steps_per_episode": 13, "sensor_range": 1}),                # N < lanes per env
    (24, 24, 24, 40, 0.20, {"lifelong_mapf": True, "steps_per_episode": 30}),         # wide groups: single-wave fused kernel
])
def test_step_many_sampled_policy_is_masked_uniform_and_replayable(shape):
    """mapf_step_many_sampled: the actions come from the in-kernel masked-random policy.  (1) every action taken was
    allowed by the mask of the observation it was picked from (the caller's observation for the first step, the
    previous step's -- reset observation included -- afterwards); (2) replaying the returned actions through single
    mapf_step launches on a twin engine gives the same observations, rewards, flags, info and final state; (3) the
    choice is spread over the valid actions; (4) same seed, same actions."""
    import torch

    B, H, W, N, density, extra = shape
    cfg = {"env_name": "synthetic", "num_agents": N, "sensor_range": 2, "steps_per_episode": 50,
           "include_action_mask_in_obs": True}
    cfg.update(extra)
    grids = synth_grids(B, H, W, density, N, base_seed=440_000)
    seeds = list(range(B))
    a, b, c = (EngineStepper(grids, cfg, seeds=seeds) for _ in range(3))
    obs0 = a.env.reset().clone()
    b.env.reset()
    c.env.reset()
    T, L = 70, a.L
    out = a.env.step_many_sampled(T, seed=1234)
    acts = out["actions"].cpu().numpy()
    obs = out["obs"].cpu().numpy()
    assert acts.min() >= 0 and acts.max() <= 4
    # (1) validity against the mask the policy saw
    prev = np.concatenate([obs0.cpu().numpy()[None], obs[:-1]])
    mask = prev[..., L - 5:] > 0.5  # [T, B, N, 5]
    assert np.take_along_axis(mask, acts[..., None].astype(np.int64), axis=-1).all()
    # (3) every action is used, and where several are valid NO_OP is not the only pick
    assert set(np.unique(acts)) == {0, 1, 2, 3, 4}
    several = mask.sum(-1) >= 3
    assert 0.15 < (acts[several] == 0).mean() < 0.5
    # (2) replay through single steps
    for t in range(T):
        r = b.env.step(torch.from_numpy(acts[t]).to(b.env.device), auto_reset=True)
        for k in ("obs", "rewards", "terminated", "truncated", "info_all", "info_agent"):
            assert np.array_equal(r[k].cpu().numpy(), out[k][t].cpu().numpy()), (k, t)
    sa, sb = a.env.get_state(), b.env.get_state()
    for k in sa:
        assert np.array_equal(sa[k], sb[k]), k
    assert int(sa["counters"][:, 9].sum()) >= B  # episodes ended (and envs were re-placed) inside the launch
    # (4) reproducible; another seed differs
    again = c.env.step_many_sampled(T, seed=1234)["actions"].cpu().numpy()
    assert np.array_equal(again, acts)
    a2 = EngineStepper(grids, cfg, seeds=seeds)
    a2.env.reset()
    assert not np.array_equal(a2.env.step_many_sampled(T, seed=99)["actions"].cpu().numpy(), acts)


def test_step_many_sampled_needs_the_mask_in_the_observation():
    st = EngineStepper(synth_grids(4, 8, 8, 0.1, 2), {"num_agents": 2, "sensor_range": 1}, seeds=[1, 2, 3, 4])
    st.env.reset()
    with pytest.raises(ValueError, match="action mask"):
        st.env.step_many_sampled(5, seed=1)
