"""Pins the CPU oracle (oracle/mapf_oracle.c) to the reference: golden traces recorded from the
unmodified reference env (oracle/gen_golden.py -> tests/golden/), the reference's own published
SHA-256 parity digests (reference tests/test_reference_model_multi_agent_parity.py:12-24), its
micro-case tests, and NumPy PCG64 known answers.  CPU only."""

from __future__ import annotations

import json

import numpy as np
import pytest

import oracle as orc
from digest_util import TraceHasher, obs_slices, ref_reset_payload, ref_step_payload
from trace_util import (
    BATCH_FIXTURES, MICRO_CASES, OracleStepper, load_golden, replay_batch_trace, replay_micro_case,
)

# published constants of the reference's golden-trace test (…_parity.py:12,19 and :13-17,:20-24)
REF_DIGEST = {
    "stochastic": "d58a9e9e0e383f29c5d7f96a1338dfd73c9035dc5335f66b2ce11a0d8e0452de",
    "deterministic": "2612dc3eeab5b4fd69d8cbe7fb4f01e35cf52a6f05b07bc955a306ae765c2595",
}
REF_SUMMARY = {
    "stochastic": [(0, 100, -3.5), (1, 100, -4.0), (2, 100, -2.5)],
    "deterministic": [(0, 100, -3.5), (1, 100, -3.5), (2, 100, -4.0)],
}


def run_parity_digest(make_env, kind: str):
    """Re-run the reference parity-test protocol through `make_env` and return (digest, summary).

    make_env(grid, config, rng_words, fixed_starts, fixed_goals) -> single-env object with
    reset() -> obs[N,L]; step(actions[N]) -> (obs, rewards, term, trunc, info_all, info_agent);
    positions / goals properties.
    """
    fx = load_golden("g1_parity_" + kind)
    cfg = fx["config"]
    env = make_env(fx["grid"], cfg, fx["rng_words"], fx["ctor_starts"], fx["fixed_goals"])
    sl = obs_slices(cfg["sensor_range"], False, False, True)
    arng = np.random.default_rng(999)
    th = TraceHasher()
    summary = []
    for ep in range(3):
        obs = env.reset()
        th.reset_record(ep, *ref_reset_payload(obs, env.positions, env.goals, sl, cfg["sensor_range"]))
        rsum = 0.0
        for st in range(140):
            acts = np.array([int(arng.integers(0, 5)) for _ in range(4)])
            obs, rew, term, trunc, info_all, info_agent = env.step(acts)
            rsum += float(sum(float(x) for x in rew))
            th.step_record(ep, st, *ref_step_payload(acts, obs, rew, term, trunc, info_all, info_agent, env.positions,
                                                     env.goals, sl, cfg["sensor_range"]))
            if term or trunc:
                summary.append((ep, st + 1, round(rsum, 6)))
                break
    return th.hexdigest(), summary


class _OracleSingle:
    def __init__(self, grid, cfg, rng_words, fixed_starts, fixed_goals):
        self.e = orc.OracleEnv(grid, cfg, rng_words=rng_words, fixed_starts=fixed_starts, fixed_goals=fixed_goals)

    positions = property(lambda self: self.e.positions.copy())
    goals = property(lambda self: self.e.goals.copy())

    def reset(self):
        rc, obs = self.e.reset()
        assert rc == 0
        return obs

    def step(self, acts):
        rc, obs, rew, term, trunc, info_all, info_agent = self.e.step(acts)
        assert rc == 0
        return obs, rew, term, trunc, info_all, info_agent


@pytest.mark.parametrize("kind", ["stochastic", "deterministic"])
def test_oracle_reproduces_reference_parity_digest(kind):
    digest, summary = run_parity_digest(_OracleSingle, kind)
    assert digest == REF_DIGEST[kind]
    assert summary == REF_SUMMARY[kind]
    assert str(load_golden("g1_parity_" + kind)["digest"]) == REF_DIGEST[kind]


@pytest.mark.parametrize("name", BATCH_FIXTURES)
def test_oracle_matches_golden_trace(name):
    fx = load_golden(name)
    stats = replay_batch_trace(OracleStepper, fx)
    assert stats["steps"] == fx["actions"].shape[0]


def test_golden_traces_cover_the_interesting_events():
    """The fixtures must actually contain deadlocks, livelocks, blocking, respawns, successes."""
    g3, g3b = load_golden("g3_c3_32x32_n8"), load_golden("g3b_tight_6x7_n6")
    assert g3["info_all"][:, :, 6].sum() >= 1 and g3["info_all"][:, :, 7].sum() >= 1
    assert g3b["info_all"][:, :, 6].sum() >= 10 and g3b["info_all"][:, :, 2].sum() >= 10
    g4, g4b = load_golden("g4_c5_64x64_n64_lifelong"), load_golden("g4b_lifelong_5x9_n10")
    assert g4["info_all"][:, :, 0].sum() >= 20 and g4b["info_all"][:, :, 0].max() >= 2
    g2 = load_golden("g2_c2_16x16_n4_greedy")
    assert (g2["terminated"].astype(bool) & ~g2["truncated"].astype(bool)).sum() >= 1
    ww = load_golden("g8_widewin_5x5_n5")
    assert ww["info_all"][:, :, 6].sum() >= 1 and ww["info_all"][:, :, 7].sum() >= 1


@pytest.mark.parametrize("name", MICRO_CASES)
def test_oracle_micro_cases(name):
    replay_micro_case(OracleStepper, load_golden("g5_micro_cases"), name)


def test_micro_case_semantics_are_what_the_reference_tests_assert():
    """Spot-check the recorded reference outputs against the assertions of the reference's own tests."""
    c = load_golden("g5_micro_cases")
    # tests/test_reference_model_lock_metrics.py:44-63
    ia = c["deadlock_on_goal_blocker.info_all"]
    assert ia[0, 6] == 0.0 and ia[1, 4] == 1.0 and ia[1, 6] == 1.0 and ia[1, 5] == 0.0 and ia[1, 8] == 1.0 and ia[2, 6] == 0.0
    # :66-86
    ia = c["deadlock_not_sticky.info_all"]
    assert ia[3, 4] == 1.0 and ia[3, 6] == 1.0
    # tests/test_reference_model_multi_agent_invariants.py:119-147: pressure of agent_1 = 0,1,1,0
    cfg = json.loads(str(c["blocking_pressure.config"]))
    sl = obs_slices(cfg["sensor_range"], False, True, False)["blocking_pressure_prev"]
    assert [float(c["blocking_pressure.obs"][t, 1, sl][0]) for t in range(4)] == [0.0, 1.0, 1.0, 0.0]
    # A.2 move rule
    assert c["follow_leader_low.positions"][0].tolist() == [[2, 3], [2, 2]]
    assert c["follow_leader_high.positions"][0].tolist() == [[2, 1], [2, 3]]
    assert c["swap.positions"][0].tolist() == [[2, 1], [2, 2]]
    assert c["cycle4.positions"][0].tolist() == [[1, 1], [1, 2], [2, 2], [2, 1]]
    assert c["contention.positions"][0].tolist() == [[2, 2], [2, 3], [1, 2]]
    # A.4 rewards
    assert c["both_reach.rewards"][0].tolist() == [1.5, 1.5] and c["both_reach.terminated"][0] == 1 and c["both_reach.truncated"][0] == 0
    assert c["truncation.terminated"][2] == 1 and c["truncation.truncated"][2] == 1
    assert c["truncation.rewards"][2].tolist() == [-1.0, -1.0]
    # tests/test_reference_model_lifelong.py:176-194
    assert c["lifelong_respawn.info_all"][0, 12] == pytest.approx(0.5)


def test_oracle_get_obs_known_answer():
    """Hand-checked 3x3 local observations + masks of the reference's tests/get_obs.py:141-164."""
    ka = load_golden("g5_get_obs_known_answer")
    env = orc.OracleEnv(ka["grid"], {"num_agents": 2, "sensor_range": int(ka["sensor_range"]), "seed": 0})
    env.positions[:] = ka["positions"]
    env.goals[:] = ka["goals"]
    env.rebuild_owner_maps()
    o0, o1 = env.get_obs(0), env.get_obs(1)
    assert np.array_equal(o0, ka["expected_obs_agent_0"])
    assert np.array_equal(o1, ka["expected_obs_agent_1"])
    assert np.array_equal(env.get_action_mask(o0), ka["expected_mask_agent_0"])
    assert np.array_equal(env.get_action_mask(o1), ka["expected_mask_agent_1"])


def test_oracle_bad_action_partial_mutation():
    fx = load_golden("g5_bad_action")
    cfg = fx["config"]
    env = orc.OracleEnv(fx["grid"], cfg, rng_words=fx["rng_words"])
    env.reset()
    assert np.array_equal(env.positions, fx["positions0"]) and np.array_equal(env.goals, fx["goals0"])
    rc, *_ = env.step(fx["actions"].astype(np.int32))
    assert rc == orc.ERR_BAD_ACTION
    assert np.array_equal(env.positions, fx["positions_after"])
    assert env.step_count == int(fx["step_count_after"])


def test_oracle_too_few_free_cells():
    grid = np.ones((3, 3), np.uint8)
    grid[0, :] = 0
    with pytest.raises(ValueError):
        orc.OracleEnv(grid, {"num_agents": 2, "seed": 0})


def test_oracle_pcg64_known_answers():
    fx = load_golden("g6_rng")
    grid = np.zeros((2, 2), np.uint8)
    for i in range(len(fx["seeds"])):
        env = orc.OracleEnv(grid, {"num_agents": 1, "seed": 0})
        env.set_rng_words(fx["words"][i])
        F, S = int(fx["F"][i]), int(fx["S"][i])
        assert np.array_equal(env.rng_choice(F, S), fx["choice1"][i][:S])
        ints = [env.rng_bounded(int(k) - 1) for k in fx["ks"]]
        assert ints == fx["ints"][i].tolist()
        assert np.array_equal(env.rng_choice(F, S), fx["choice2"][i][:S])
        assert np.array_equal(env.rng_words(), fx["final_words"][i])


def test_oracle_pcg64_against_live_numpy():
    """Interleaved choice / integers draws against the NumPy installed on this box."""
    grid = np.zeros((2, 2), np.uint8)
    for seed in range(40):
        rng = np.random.default_rng(seed)
        env = orc.OracleEnv(grid, {"num_agents": 1, "seed": 0})
        env.set_rng_words(orc.pcg64_words_from_state(rng.bit_generator.state))
        meta = np.random.default_rng(1000 + seed)
        for _ in range(12):
            if meta.random() < 0.5:
                F = int(meta.integers(2, 4097))
                S = int(meta.integers(1, min(F, 128) + 1))
                assert np.array_equal(env.rng_choice(F, S), rng.choice(F, size=S, replace=False))
            else:
                k = int(meta.integers(1, 5000))
                assert env.rng_bounded(k - 1) == int(rng.integers(k))
        assert np.array_equal(env.rng_words(), orc.pcg64_words_from_state(rng.bit_generator.state))
