"""Round-4 additions, each against the oracle or a reference-recorded fixture:
  * the hand-checked known answer of the reference's tests/get_obs.py:141-164 through the HIP path (mapf_observe and the
    facade's get_obs / get_action_mask), not only through the oracle;
  * `_assign_new_goal(agent_idx)` called by itself (MA-env:284-304; the reference's lifelong tests do,
    tests/test_reference_model_lifelong.py:132-173): mapf_assign_new_goal vs the oracle's mo_assign_new_goal, goals and
    generator words, in lifelong mode, in finite mode with a pre-drawn placement pending, and with k = 1 (no draw);
  * mapf_set_grids fixes the visible streams and the pass bits on the device (no host round trip);
  * the c5 shape against the oracle at its full batch of 1 024 envs."""
from __future__ import annotations

import ctypes as C

import numpy as np
import pytest
import torch

from trace_util import EngineStepper, OracleStepper, _eq, load_golden, synth_grids

pytestmark = pytest.mark.gpu


# ---- tests/get_obs.py:141-164 ------------------------------------------------------------------------------------------
def test_get_obs_known_answer_through_the_hip_path():
    ka = load_golden("g5_get_obs_known_answer")
    sr = int(ka["sensor_range"])
    v = 2 * sr + 1
    cfg = {"env_name": "synthetic", "num_agents": 2, "sensor_range": sr, "include_action_mask_in_obs": True,
           "include_blocking_pressure_in_obs": False, "seed": 0}
    eng = EngineStepper(ka["grid"][None], cfg, seeds=[0])
    eng.env.set_state(positions=ka["positions"][None].astype(np.int16), goals=ka["goals"][None].astype(np.int16),
                      starts=ka["positions"][None].astype(np.int16), clear_episode=True)
    obs = eng.env.observe().cpu().numpy()[0]  # mapf_observe: [N][L] = local window, goal delta (2), mask (5)
    for a in (0, 1):
        _eq(f"local obs of agent {a}", obs[a, : v * v].astype(np.uint8).reshape(v, v), ka[f"expected_obs_agent_{a}"])
        _eq(f"mask of agent {a}", obs[a, v * v + 2:].astype(np.int8), ka[f"expected_mask_agent_{a}"])


def test_get_obs_known_answer_through_the_facade():
    from dl_reference_models_amd.reference_model_multi_agent import ReferenceModel

    ka = load_golden("g5_get_obs_known_answer")
    env = ReferenceModel({"env_name": "synthetic", "grid": ka["grid"], "num_agents": 2, "sensor_range": int(ka["sensor_range"]),
                          "seed": 0})
    env._positions_arr[:] = ka["positions"]
    env._starts_arr[:] = ka["positions"]
    env._goals_arr[:] = ka["goals"]
    env._rebuild_occupancy_owner()
    env._rebuild_goal_owner()
    for a in (0, 1):
        local = env.get_obs(f"agent_{a}")
        _eq(f"get_obs(agent_{a})", local, ka[f"expected_obs_agent_{a}"])
        _eq(f"get_action_mask(agent_{a})", env.get_action_mask(local), ka[f"expected_mask_agent_{a}"])


# ---- _assign_new_goal by itself -------------------------------------------------------------------------------------------
def _assign_both(eng, orc, env, agent):
    got = eng.env.assign_new_goal(env, agent)
    rc = orc.batch.envs[env].assign_new_goal(agent)
    assert rc == 0, rc
    _eq(f"new goal of env {env} agent {agent}", got, orc.batch.envs[env].goals[agent])


@pytest.mark.parametrize("n,H,W,lifelong", [(8, 9, 11, True), (64, 20, 21, True), (5, 7, 7, False)])
def test_assign_new_goal_matches_the_oracle(n, H, W, lifelong):
    B = 6
    cfg = {"env_name": "synthetic", "num_agents": n, "sensor_range": 2, "steps_per_episode": 40, "lifelong_mapf": lifelong,
           "include_action_mask_in_obs": True}
    grids = synth_grids(B, H, W, 0.15, n, base_seed=77_000)
    seeds = list(range(50, 50 + B))
    eng, orc = EngineStepper(grids, cfg, seeds=seeds), OracleStepper(grids, cfg, seeds=seeds)
    _eq("reset", eng.reset(), orc.reset())
    rng = np.random.default_rng(4)
    for rnd in range(12):
        for _ in range(3):  # a few steps in between: positions move, lifelong respawns consume the stream inside step()
            a = rng.integers(0, 5, size=(B, n)).astype(np.int8)
            ra, rb = eng.step(a), orc.step(a)
            for k in ("obs", "rewards", "terminated", "truncated", "info_all"):
                _eq(k, ra[k], rb[k], rnd)
        for _ in range(4):
            _assign_both(eng, orc, int(rng.integers(0, B)), int(rng.integers(0, n)))
        _eq("goals", eng.goals(), orc.goals(), rnd)
        _eq("rng words", eng.rng_words(), orc.rng_words(), rnd)
        obs = eng.env.observe().cpu().numpy()  # mapf_observe on the state the respawns left
        for b in range(B):
            win, mask = _oracle_windows(orc, b)
            _eq(f"windows of env {b}", obs[b, :, :25], win, rnd)
            _eq(f"masks of env {b}", obs[b, :, -5:], mask, rnd)


def _oracle_windows(orc, b):
    """Local windows and masks of env b's agents from its current state through the oracle's own helpers
    (mo_get_obs / mo_get_action_mask), flattened like the head and tail of an observation row."""
    e = orc.batch.envs[b]
    wins = [e.get_obs(a) for a in range(orc.N)]
    return (np.stack([w.ravel() for w in wins]).astype(np.float32),
            np.stack([e.get_action_mask(w) for w in wins]).astype(np.float32))


def test_assign_new_goal_voids_a_pending_placement_and_the_next_reset_still_matches():
    """Finite mode: the background draw has already advanced the env's stream for the NEXT reset (slot pending, visible state
    in vis_rng).  A respawn by itself draws from the VISIBLE state and voids that placement; the following resets must be
    the oracle's."""
    B, n = 64, 8
    cfg = {"env_name": "synthetic", "num_agents": n, "sensor_range": 2, "steps_per_episode": 30, "include_action_mask_in_obs": True}
    grids = synth_grids(B, 12, 12, 0.1, n, base_seed=78_000)
    seeds = list(range(B))
    eng, orc = EngineStepper(grids, cfg, seeds=seeds), OracleStepper(grids, cfg, seeds=seeds)
    _eq("reset", eng.reset(), orc.reset())
    rng = np.random.default_rng(5)

    def steps(k, tag):
        for t in range(k):
            a = rng.integers(0, 5, size=(B, n)).astype(np.int8)
            ra, rb = eng.step(a), orc.step(a)
            for key in ("obs", "rewards", "terminated", "truncated", "info_all"):
                _eq(key, ra[key], rb[key], f"{tag}{t}")

    steps(14, "a")  # (seven slices: every env has a placement pending by now)
    slots = np.zeros((B, n), np.uint32)
    eng.env._lib.mapf_debug_slots(eng.env._h, slots.ctypes.data_as(C.c_void_p), None, None)
    pending = np.flatnonzero(slots[:, 0] != 0xFFFFFFFF)
    assert len(pending) > B // 2, "expected pre-drawn placements by step 14"
    idle = np.flatnonzero(slots[:, 0] == 0xFFFFFFFF)
    for env in pending[:20].tolist() + idle[:2].tolist():
        _assign_both(eng, orc, int(env), int(rng.integers(0, n)))
    _eq("rng words", eng.rng_words(), orc.rng_words())
    _eq("goals", eng.goals(), orc.goals())
    steps(40, "b")  # across the episode boundary: every env resets from the stream the respawn left
    _eq("rng words after the resets", eng.rng_words(), orc.rng_words())


def test_assign_new_goal_with_one_candidate_draws_nothing():
    """1 x 4 corridor, 2 agents, F = 2N: with the other agent's goal on a free cell the asking agent has exactly one
    candidate -- its own old goal -- and rng.integers(1) consumes nothing (NumPy returns 0 without a draw).  k = 0, the
    reference's RuntimeError (MA-env:296-298), cannot be reached from a valid state: k >= F - N - (N - 1) >= 1."""
    grid = np.zeros((1, 1, 4), np.uint8)
    cfg = {"env_name": "synthetic", "num_agents": 2, "sensor_range": 1, "lifelong_mapf": True}
    eng, orc = EngineStepper(grid, cfg, seeds=[3]), OracleStepper(grid, cfg, seeds=[3])
    pos = np.array([[[0, 0], [0, 1]]], np.int16)
    goals = np.array([[[0, 2], [0, 3]]], np.int16)
    eng.set_state(pos, goals)
    orc.set_state(pos, goals)
    before = eng.rng_words().copy()
    _assign_both(eng, orc, 0, 0)  # candidates: cell 2 only (0, 1 occupied, 3 is agent 1's goal) -> k = 1
    assert eng.goals()[0, 0].tolist() == [0, 2]
    _eq("rng words untouched by k = 1", eng.rng_words(), before)
    _eq("rng words", eng.rng_words(), orc.rng_words())
    # F = 6 = 2N with three agents: agent 0 stands on its own goal, the others' goals are free cells: one cell is left
    grid3 = np.zeros((1, 1, 6), np.uint8)
    cfg3 = dict(cfg, num_agents=3)
    eng3, orc3 = EngineStepper(grid3, cfg3, seeds=[4]), OracleStepper(grid3, cfg3, seeds=[4])
    pos3 = np.array([[[0, 0], [0, 1], [0, 2]]], np.int16)
    goals3 = np.array([[[0, 0], [0, 3], [0, 4]]], np.int16)
    eng3.set_state(pos3, goals3)
    orc3.set_state(pos3, goals3)
    _assign_both(eng3, orc3, 0, 0)
    assert eng3.goals()[0, 0].tolist() == [0, 5]
    _eq("rng words", eng3.rng_words(), orc3.rng_words())
    with pytest.raises(RuntimeError, match="out of range"):
        eng3.env.assign_new_goal(0, 3)


def test_facade_assign_new_goal_matches_the_oracle():
    """The drop-in's `_assign_new_goal(agent_idx)` (the reference's lifelong tests call it, :132-173)."""
    from dl_reference_models_amd.reference_model_multi_agent import ReferenceModel
    import oracle as orc_mod

    grid = synth_grids(1, 10, 12, 0.2, 4, base_seed=91_000)[0]
    cfg = {"env_name": "synthetic", "grid": grid, "num_agents": 4, "sensor_range": 2, "lifelong_mapf": True, "seed": 21}
    env = ReferenceModel(cfg)
    ref = orc_mod.OracleEnv(grid, cfg)
    obs, _ = env.reset()
    ref.reset()
    _eq("positions after reset", env._positions_arr, ref.positions)
    for k in range(10):
        a = k % 4
        new_goal = env._assign_new_goal(a)
        assert ref.assign_new_goal(a) == 0
        _eq(f"returned goal {k}", np.asarray(new_goal), ref.goals[a])
        _eq(f"_goals_arr {k}", env._goals_arr, ref.goals)
        assert new_goal.dtype == env._coord_dtype
    _eq("rng words", env._engine.get_state()["rng_words"][0], ref.rng_words())
    # and the next step still matches (the goal owner map of the kernel is derived from the goals it reads)
    acts = {f"agent_{i}": int(i % 5) for i in range(4)}
    o, r, *_ = env.step(acts)
    rc, o2, r2, *_ = ref.step(np.array([acts[f"agent_{i}"] for i in range(4)], np.int32))
    assert rc == 0
    for i in range(4):
        _eq(f"obs agent {i}", o[f"agent_{i}"], o2[i])


# ---- mapf_set_grids: device-side fix-ups ------------------------------------------------------------------------------
def test_set_grids_recomputes_pass_bits_on_the_device():
    """After new grids the agents' pass bits (which neighbours the grid lets them step on) come from the new rows: a move
    into a cell that has become free must succeed, a move into a new obstacle must fail -- engine vs oracle built on the new
    grids with the same state."""
    B, n, H, W = 9, 8, 10, 10
    cfg = {"env_name": "synthetic", "num_agents": n, "sensor_range": 2, "steps_per_episode": 50, "include_action_mask_in_obs": True}
    g1 = synth_grids(B, H, W, 0.3, n, base_seed=81_000)
    eng = EngineStepper(g1, cfg, seeds=list(range(B)))
    eng.reset()
    st = eng.env.get_state()
    # new grids: obstacles reshuffled except under agents and goals (so the state stays valid)
    g2 = synth_grids(B, H, W, 0.3, n, base_seed=82_000)
    for b in range(B):
        for cell in np.concatenate([st["positions"][b], st["goals"][b]]):
            g2[b, cell[0], cell[1]] = 0
    eng.env.set_grids(g2)
    orc = OracleStepper(g2, cfg, rng_words=eng.rng_words())
    orc.set_state(st["positions"], st["goals"], rng_words=eng.rng_words())
    eng.set_state(st["positions"], st["goals"], rng_words=eng.rng_words())  # (clears episode bookkeeping on both sides)
    rng = np.random.default_rng(8)
    for t in range(30):
        a = rng.integers(0, 5, size=(B, n)).astype(np.int8)
        ra, rb = eng.step(a), orc.step(a)
        for k in ("obs", "rewards", "terminated", "truncated", "info_all"):
            _eq(k, ra[k], rb[k], t)
    _eq("positions", eng.positions(), orc.positions())


# ---- c5 at its full batch -----------------------------------------------------------------------------------------------
def test_engine_vs_oracle_c5_full_batch():
    """BASELINE config 5 at B = 1 024 (it was compared at B = 256 only): 64 x 64, 64 agents, lifelong."""
    from dl_reference_models_amd import workloads as wl

    name = "c5_1024x64x64_n64_lifelong"
    B = wl.WORKLOADS[name][0]
    ids = list(range(B))
    cfg = wl.workload_config(name, ids)
    grids = cfg.pop("grid")
    seeds = cfg.pop("seeds")
    cfg.pop("num_envs")
    eng, orc = EngineStepper(grids, cfg, seeds=seeds), OracleStepper(grids, cfg, seeds=seeds)
    assert eng.env.launch_info()["specialized_kernel"] == 3
    _eq("reset", eng.reset(), orc.reset())
    rng = np.random.default_rng(999)
    for t in range(12):
        a = rng.integers(0, 5, size=(B, 64)).astype(np.int8)
        ra, rb = eng.step(a), orc.step(a)
        for k in ("obs", "rewards", "terminated", "truncated", "info_all", "info_agent"):
            _eq(k, ra[k], rb[k], t)
    _eq("goals", eng.goals(), orc.goals())
    _eq("rng words", eng.rng_words(), orc.rng_words())


# ---- k_stepw: the three-wave kernel of 64-lane groups ------------------------------------------------------------------
def _wide_cfg(n, **over):
    cfg = {"env_name": "synthetic", "num_agents": n, "sensor_range": 2, "include_action_mask_in_obs": True, "steps_per_episode": 13}
    cfg.update(over)
    return cfg


@pytest.mark.parametrize("two_wave", [False, True])
@pytest.mark.parametrize("n,H,W,over", [
    (64, 64, 64, {"lifelong_mapf": True, "steps_per_episode": 9}),                       # the c5 shape, episodes ending often
    (40, 20, 21, {"steps_per_episode": 7}),                                               # finite: slots from the sampler workgroups, N < 64
    (33, 16, 17, {"deterministic": True, "steps_per_episode": 5}),                        # fixed starts: fast reset without a slot
    (48, 30, 30, {"livelock_window_steps": 40, "deadlock_window_steps": 20, "lifelong_mapf": True}),  # int16 distance ring
    (36, 12, 12, {"sensor_range": 4, "lock_nearby_manhattan": 4, "lock_min_neighbors": 2, "include_goal_distance": True}),
    (6, 9, 9, {"lanes_per_env": 64, "steps_per_episode": 4}),                             # a small env forced onto one wavefront
])
def test_wide_kernel_and_the_two_wave_kernel_match_the_oracle(n, H, W, over, two_wave):
    """64-lane groups step on k_stepw (three waves per env, bit rows in LDS) unless the `wide_kernel` knob keeps them on the
    two-wave kernel with the word-per-cell map: both against the oracle, with and without the terminal observation."""
    over = dict(over)
    kw = {"lanes_per_env": over.pop("lanes_per_env")} if "lanes_per_env" in over else {}
    if two_wave:
        kw["wide_kernel"] = "two_wave"
    cfg = _wide_cfg(n, **over)
    B = 21
    grids = synth_grids(B, H, W, 0.12, n, base_seed=93_000)
    seeds = list(range(40, 40 + B))
    fs = fg = None
    if cfg.get("deterministic"):
        rng = np.random.default_rng(3)
        cells = [np.argwhere(grids[b] == 0) for b in range(B)]
        pick = [c[rng.permutation(len(c))[: 2 * n]] for c in cells]
        fs = np.stack([q[:n] for q in pick]).astype(np.int16)
        fg = np.stack([q[n:] for q in pick]).astype(np.int16)
    for want_final in (False, True):
        eng = EngineStepper(grids, cfg, seeds=seeds, fixed_starts=fs, fixed_goals=fg, want_final_obs=want_final, **kw)
        orc = OracleStepper(grids, cfg, seeds=seeds, fixed_starts=fs, fixed_goals=fg)
        assert eng.env.launch_info()["threads"] == (128 if two_wave else 192)
        _eq("reset", eng.reset(), orc.reset())
        counts = np.arange(B) % int(cfg["steps_per_episode"])
        eng.set_step_counts(counts)
        orc.set_step_counts(counts)
        rng = np.random.default_rng(11)
        for t in range(45):
            a = rng.integers(0, 5, size=(B, n)).astype(np.int8)
            ra, rb = eng.step(a), orc.step(a)
            for k in ("obs", "rewards", "terminated", "truncated", "info_all", "info_agent"):
                _eq(k, ra[k], rb[k], t)
            done = (rb["terminated"] | rb["truncated"]).astype(bool)
            if want_final and done.any():
                _eq("final_obs", ra["final_obs"][done], rb["final_obs"][done], t)
        _eq("positions", eng.positions(), orc.positions())
        _eq("goals", eng.goals(), orc.goals())
        _eq("rng words", eng.rng_words(), orc.rng_words())
        eng.env.poll_error()


def test_wide_kernel_invalid_action_partial_mutation():
    """MA-env:502-506 at 64 lanes per env: the agents before the bad one are processed (moves, goal logic, lifelong
    respawns), nothing after the loop runs; the engine latches the error.  State compared with the oracle stopped the
    same way."""
    n, B = 40, 5
    cfg = _wide_cfg(n, lifelong_mapf=True, steps_per_episode=50)
    grids = synth_grids(B, 14, 15, 0.1, n, base_seed=94_000)
    seeds = list(range(B))
    eng, orc = EngineStepper(grids, cfg, seeds=seeds), OracleStepper(grids, cfg, seeds=seeds)
    _eq("reset", eng.reset(), orc.reset())
    rng = np.random.default_rng(2)
    for t in range(6):
        a = rng.integers(0, 5, size=(B, n)).astype(np.int8)
        ra, rb = eng.step(a), orc.step(a)
        _eq("obs", ra["obs"], rb["obs"], t)
    a = rng.integers(0, 5, size=(B, n)).astype(np.int8)
    a[2, 17] = 7  # env 2: agents 0 .. 16 are processed, then the reference raises
    eng.step(a)
    with pytest.raises(ValueError, match="Invalid action 7 for agent_17"):
        eng.env.poll_error()
    for b in range(B):
        rc, *_ = orc.batch.envs[b].step(a[b].astype(np.int32))
        assert rc == (orc.orc.ERR_BAD_ACTION if b == 2 else 0)
    _eq("positions", eng.positions(), orc.positions())
    _eq("goals", eng.goals(), orc.goals())
    _eq("rng words", eng.rng_words(), orc.rng_words())
    st = eng.env.get_state()
    _eq("step counts", st["counters"][:, 0], np.array([e.step_count for e in orc.batch.envs], np.int32))
    # and the engine goes on like the reference's env object does after the exception
    for t in range(5):
        a = rng.integers(0, 5, size=(B, n)).astype(np.int8)
        ra, rb = eng.step(a), orc.step(a)
        for k in ("obs", "rewards", "info_all"):
            _eq(k, ra[k], rb[k], t)


def test_wide_kernel_masked_step_leaves_masked_envs_untouched():
    n, B = 64, 7
    cfg = _wide_cfg(n, lifelong_mapf=True, steps_per_episode=30)
    grids = synth_grids(B, 20, 20, 0.1, n, base_seed=95_000)
    eng = EngineStepper(grids, cfg, seeds=list(range(B)))
    eng.reset()
    before = eng.env.get_state()
    mask = torch.tensor([1, 0, 1, 1, 0, 0, 1], dtype=torch.uint8, device=eng.env.device)
    a = torch.from_numpy(np.random.default_rng(1).integers(0, 5, size=(B, n)).astype(np.int8)).to(eng.env.device)
    a[1, 3] = 9  # garbage in a masked-out row must not latch anything
    eng.env.step(a, env_mask=mask)
    eng.env.poll_error()
    after = eng.env.get_state()
    for b in (1, 4, 5):
        for k in before:
            assert np.array_equal(before[k][b], after[k][b]), (k, b)
    assert (after["counters"][[0, 2, 3, 6], 0] == 1).all()


# ---- k_step3 with bit rows: 16-lane groups (the reference's training setup and its runtime-config siblings) -----------
_KNOBS16 = {None: {}, "table_walk": {"small_group_observation": "table_walk"}, "rows_off": {"small_group_rows": "off"}}


@pytest.mark.parametrize("knob", list(_KNOBS16))
@pytest.mark.parametrize("H,W,over", [
    (32, 32, {"sensor_range": 3, "include_action_mask_in_obs": False}),                      # specialisation 6 (main.py:55-68)
    (20, 21, {"sensor_range": 2, "steps_per_episode": 6}),                                    # runtime-config kernel, 5 x 5 windows, action mask
    (12, 54, {"sensor_range": 3, "include_goal_distance": True, "include_action_mask_in_obs": False}),  # widest grid with sentinel columns
    (12, 56, {"sensor_range": 3, "include_action_mask_in_obs": False}),                      # no sentinel columns: no rows whatever the knob
    (16, 16, {"sensor_range": 4, "steps_per_episode": 5}),                                    # 9 x 9 windows: no rows
    (4, 40, {"sensor_range": 1, "steps_per_episode": 4, "include_action_mask_in_obs": False}),  # a corridor: contended moves, intents onto goals
])
def test_sixteen_lane_groups_with_and_without_bit_rows_match_the_oracle(H, W, over, knob):
    """Groups of 16 agents on the three-wave kernel: observation wave from the bit rows (goals / old cells, toggles of the
    lower-index movers nearby), intent blocking from the intent rows, per-agent outputs by the observation wave (round 4) --
    and, by the engine knobs, round 3's table walk / round 3's kernel.  All against the oracle: staggered episode ends
    (foreseen reset observations), with and without the terminal observation, a ragged last workgroup."""
    n, B = 16, 22
    cfg = {"env_name": "synthetic", "num_agents": n, "include_action_mask_in_obs": True, "steps_per_episode": 9}
    cfg.update(over)
    grids = synth_grids(B, H, W, 0.1, n, base_seed=96_000)
    seeds = list(range(70, 70 + B))
    for want_final in (False, True):
        eng = EngineStepper(grids, cfg, seeds=seeds, want_final_obs=want_final, **_KNOBS16[knob])
        orc = OracleStepper(grids, cfg, seeds=seeds)
        assert eng.env.launch_info()["threads"] == 192
        _eq("reset", eng.reset(), orc.reset())
        counts = np.arange(B) % int(cfg["steps_per_episode"])
        eng.set_step_counts(counts)
        orc.set_step_counts(counts)
        rng = np.random.default_rng(12)
        for t in range(60):
            a = rng.integers(0, 5, size=(B, n)).astype(np.int8)
            if t % 3 == 2:  # more agents pushing the same way: blocked moves, intents into occupied cells
                a[:, ::2] = 2 + 2 * (t % 2)
            ra, rb = eng.step(a), orc.step(a)
            for k in ("obs", "rewards", "terminated", "truncated", "info_all", "info_agent"):
                _eq(k, ra[k], rb[k], t)
            done = (rb["terminated"] | rb["truncated"]).astype(bool)
            if want_final and done.any():
                _eq("final_obs", ra["final_obs"][done], rb["final_obs"][done], t)
        _eq("positions", eng.positions(), orc.positions())
        _eq("goals", eng.goals(), orc.goals())
        _eq("rng words", eng.rng_words(), orc.rng_words())
        eng.env.poll_error()


def test_sixteen_lane_groups_kernel_compiled_at_creation_uses_the_bit_rows_too():
    """`jit_specialize=True` on a 16-agent configuration that is not prebuilt: the kernel hiprtc compiles at creation is the
    three-wave kernel with the bit-row paths (same launch plan, same LDS) -- against the oracle, staggered episode ends."""
    n, B, H, W = 16, 26, 18, 23
    cfg = {"env_name": "synthetic", "num_agents": n, "sensor_range": 2, "include_action_mask_in_obs": True, "include_goal_distance": True,
           "steps_per_episode": 8}
    grids = synth_grids(B, H, W, 0.1, n, base_seed=96_500)
    seeds = list(range(B))
    eng = EngineStepper(grids, cfg, seeds=seeds, jit_specialize=True)
    info = eng.env.launch_info()
    if not info["jit"]:
        pytest.skip(f"no run-time compilation here: {info['jit_note']}")
    assert info["threads"] == 192
    orc = OracleStepper(grids, cfg, seeds=seeds)
    _eq("reset", eng.reset(), orc.reset())
    counts = np.arange(B) % 8
    eng.set_step_counts(counts)
    orc.set_step_counts(counts)
    rng = np.random.default_rng(21)
    for t in range(50):
        a = rng.integers(0, 5, size=(B, n)).astype(np.int8)
        ra, rb = eng.step(a), orc.step(a)
        for k in ("obs", "rewards", "terminated", "truncated", "info_all", "info_agent"):
            _eq(k, ra[k], rb[k], t)
    _eq("rng words", eng.rng_words(), orc.rng_words())
    eng.env.poll_error()


def test_sixteen_lane_groups_unreached_agent_standing_on_its_goal():
    """The case the intent rows cannot decide before the moves: an agent that has not "reached" its goal but stands on it
    (an injected state), or whose target is its goal -- it publishes its intent iff it does not end the step there."""
    n, B = 16, 8
    cfg = {"env_name": "synthetic", "num_agents": n, "sensor_range": 3, "include_action_mask_in_obs": False, "steps_per_episode": 50}
    grids = synth_grids(B, 10, 10, 0.05, n, base_seed=97_000)
    seeds = list(range(B))
    eng, orc = EngineStepper(grids, cfg, seeds=seeds), OracleStepper(grids, cfg, seeds=seeds)
    _eq("reset", eng.reset(), orc.reset())
    rng = np.random.default_rng(5)
    for t in range(40):
        if t % 8 == 3:  # move every agent's goal under its feet (flags untouched: reached stays as it is)
            pos = eng.positions()
            eng.env.set_state(goals=pos.copy())
            for b in range(B):
                orc.batch.envs[b].goals[:] = pos[b]
                orc.batch.envs[b].rebuild_owner_maps()
        a = rng.integers(0, 5, size=(B, n)).astype(np.int8)
        if t == 19:  # everybody stays on its goal: the episode ends by SUCCESS, long before the step limit -- a reset
            a[::2] = 0  # observation the preparing waves did not see coming (the table-walk fallback), in every other env
        ra, rb = eng.step(a), orc.step(a)
        if t == 19:
            assert rb["terminated"][::2].all() and not rb["truncated"][::2].any()
        for k in ("obs", "rewards", "terminated", "truncated", "info_all", "info_agent"):
            _eq(k, ra[k], rb[k], t)
    _eq("rng words", eng.rng_words(), orc.rng_words())
    eng.env.poll_error()


def test_sixteen_lane_groups_invalid_action_and_masked_step():
    """A workgroup with an invalid action or a step mask runs the two-wave code of the kernel (no rows): same results."""
    n, B = 16, 9
    cfg = {"env_name": "synthetic", "num_agents": n, "sensor_range": 3, "include_action_mask_in_obs": False, "steps_per_episode": 30}
    grids = synth_grids(B, 14, 15, 0.1, n, base_seed=98_000)
    seeds = list(range(B))
    eng, orc = EngineStepper(grids, cfg, seeds=seeds), OracleStepper(grids, cfg, seeds=seeds)
    _eq("reset", eng.reset(), orc.reset())
    rng = np.random.default_rng(2)
    for t in range(5):
        a = rng.integers(0, 5, size=(B, n)).astype(np.int8)
        ra, rb = eng.step(a), orc.step(a)
        _eq("obs", ra["obs"], rb["obs"], t)
    a = rng.integers(0, 5, size=(B, n)).astype(np.int8)
    a[2, 9] = 7
    eng.step(a)
    with pytest.raises(ValueError, match="Invalid action 7 for agent_9"):
        eng.env.poll_error()
    for b in range(B):
        rc, *_ = orc.batch.envs[b].step(a[b].astype(np.int32))
        assert rc == (orc.orc.ERR_BAD_ACTION if b == 2 else 0)
    _eq("positions", eng.positions(), orc.positions())
    for t in range(5):
        a = rng.integers(0, 5, size=(B, n)).astype(np.int8)
        ra, rb = eng.step(a), orc.step(a)
        for k in ("obs", "rewards", "info_all", "info_agent"):
            _eq(k, ra[k], rb[k], t)
    before = eng.env.get_state()
    mask = torch.tensor([1, 0, 1, 1, 0, 0, 1, 1, 0], dtype=torch.uint8, device=eng.env.device)
    at = torch.from_numpy(rng.integers(0, 5, size=(B, n)).astype(np.int8)).to(eng.env.device)
    eng.env.step(at, env_mask=mask)
    eng.env.poll_error()
    after = eng.env.get_state()
    for b in (1, 4, 5, 8):
        for k in before:
            assert np.array_equal(before[k][b], after[k][b]), (k, b)


# ---- single-agent env: episode ends at different steps, with and without the terminal observation ----------------------
@pytest.mark.parametrize("lanes,B,H,W,N,spe", [(0, 130, 16, 16, 4, 7), (0, 40, 32, 32, 8, 5), (8, 65, 9, 7, 5, 3), (64, 24, 12, 12, 3, 1),
                                                (16, 33, 10, 10, 12, 4)])
@pytest.mark.parametrize("want_final", [False, True])
def test_single_agent_env_staggered_resets_match_the_oracle(lanes, B, H, W, N, spe, want_final):
    """Round 4 rebuilt this env's episode boundary: the placement draw is lane-parallel (first occurrences among 2N + 8
    attempts made at once, SA-env:158-191), the reset observation is the observation wave's -- in ONE pass when nobody asks
    for the terminal observation, in two when somebody does.  Episodes end in different steps (staggered phases, short
    episodes, goal-seeking actions so that some end by success); generator words compared."""
    from trace_util import CteEngineStepper, CteOracleStepper

    cfg = {"env_name": "synthetic", "num_agents": N, "steps_per_episode": spe}
    grids = synth_grids(B, H, W, 0.2, N, base_seed=150_000 + lanes)
    seeds = list(range(B))
    kw = {"lanes_per_env": lanes} if lanes else {}
    eng = CteEngineStepper(grids, cfg, seeds=seeds, **kw)
    orc = CteOracleStepper(grids, cfg, seeds=seeds)
    _eq("reset obs", eng.reset(), orc.reset())
    counts = np.arange(B) % spe
    eng.env.set_step_counts(counts)
    for e, c in zip(orc.envs, counts):
        e._step[0] = int(c)
    rng = np.random.default_rng(9)
    for t in range(50):
        pos, gl = orc.positions().astype(int), orc.goals().astype(int)
        d = gl - pos
        greedy = np.where(np.abs(d[..., 0]) >= np.abs(d[..., 1]), np.where(d[..., 0] > 0, 3, np.where(d[..., 0] < 0, 1, 0)),
                          np.where(d[..., 1] > 0, 2, 4))
        acts = np.where(rng.random((B, N)) < 0.7, greedy, rng.integers(0, 5, size=(B, N))).astype(np.int8)
        a = torch.as_tensor(acts, device=eng.env.device)
        out = eng.env.step(a, auto_reset=True, want_final_obs=want_final)
        ref = orc.step(acts)
        for k in ("obs", "reward", "terminated", "truncated", "info"):
            _eq(k, out[k].cpu().numpy(), ref[k], t)
        done = (ref["terminated"] | ref["truncated"]).astype(bool)
        if want_final and done.any():
            _eq("final_obs", out["final_obs"].cpu().numpy()[done], ref["final_obs"][done], t)
        if t % 7 == 0:
            _eq("rng words", eng.rng_words(), orc.rng_words(), t)
    _eq("positions", eng.positions(), orc.positions())
    _eq("goals", eng.goals(), orc.goals())
    eng.env.poll_error()


# ---- single-agent env: next-episode placements pre-drawn by sampler workgroups -------------------------------------------
def _cte_slots(eng):
    import ctypes as C
    B, N = eng.B, eng.N
    sl = np.zeros(B * N, np.uint32)
    eng.env._lib.mapf_debug_slots(eng.env._h, sl.ctypes.data_as(C.c_void_p), None, None)
    return sl.reshape(B, N)


@pytest.mark.parametrize("B,H,W,N,spe,lanes", [(200, 16, 16, 4, 7, 0), (70, 32, 32, 8, 5, 0), (33, 10, 10, 12, 2, 16), (130, 9, 7, 5, 11, 8)])
def test_single_agent_env_predrawn_placements(B, H, W, N, spe, lanes):
    """Single-step launches of the single-agent env carry sampler workgroups that draw the NEXT episode's placement of an
    env that has none and cannot end its episode in that launch; a reset then takes it (no inline draw).  While a placement
    is pending the stream array is one draw ahead and `get_state` reports the VISIBLE stream -- compared with the oracle's
    generator after EVERY step, with staggered phases; slots do get filled and consumed; an explicit `reset(env_mask)`
    consumes them too; fused launches in between take and leave them as they are."""
    import torch

    from trace_util import CteEngineStepper, CteOracleStepper

    cfg = {"env_name": "synthetic", "num_agents": N, "steps_per_episode": spe}
    grids = synth_grids(B, H, W, 0.15, N, base_seed=99_000)
    seeds = list(range(300, 300 + B))
    a = CteEngineStepper(grids, cfg, seeds=seeds, lanes_per_env=lanes)
    b = CteOracleStepper(grids, cfg, seeds=seeds)
    _eq("reset obs", a.reset(), b.reset())
    counts = np.arange(B) % spe
    a.env.set_step_counts(counts)
    for e, c in zip(b.envs, counts):
        e._step[0] = int(c)
    rng = np.random.default_rng(8)
    seen_valid = 0
    for t in range(6 * spe + 9):
        acts = rng.integers(0, 5, size=(B, N)).astype(np.int8)
        ra, rb = a.step(acts), b.step(acts)
        for k in ("obs", "reward", "terminated", "truncated", "info"):
            _eq(k, ra[k], rb[k], t)
        done = (rb["terminated"] | rb["truncated"]).astype(bool)
        if done.any():
            _eq("final_obs", ra["final_obs"][done], rb["final_obs"][done], t)
        _eq("visible stream", a.rng_words(), b.rng_words(), t)
        seen_valid = max(seen_valid, int((_cte_slots(a)[:, 0] < 0xFFFFFFF0).sum()))
        if t == 2 * spe + 1:  # an explicit reset of some envs: pending placements are what rng.choice returns
            mask = (np.arange(B) % 3 == 0)
            oa = a.env.reset(env_mask=torch.from_numpy(mask.astype(np.uint8))).cpu().numpy()
            for i in np.nonzero(mask)[0]:
                _eq(f"explicit reset obs of env {i}", oa[i], b.envs[i].reset(), t)
            _eq("visible stream after reset", a.rng_words(), b.rng_words(), t)
        if t == 4 * spe:  # a fused launch in between
            T = spe + 3
            acts_t = rng.integers(0, 5, size=(T, B, N)).astype(np.int8)
            out = a.env.step_many(torch.from_numpy(acts_t).to(a.env.device), obs_mode=2)
            refs = [b.step(acts_t[k]) for k in range(T)]
            _eq("fused obs", out["obs"].cpu().numpy(), np.stack([r["obs"] for r in refs]), t)
            _eq("visible stream after the fused launch", a.rng_words(), b.rng_words(), t)
    _eq("positions", a.positions(), b.positions())
    _eq("goals", a.goals(), b.goals())
    # (a sampler wave draws for one env per lane group and launch: short episodes on wide groups outrun it and most resets
    #  draw inline -- both paths are in the comparison above; episodes of two steps are never pre-drawn: the step limit is
    #  always one step away)
    if spe > 2:
        assert seen_valid >= 4, seen_valid
    a.env.poll_error()
