"""The harness counterpart (scripts/benchmark_multi_agent_env.py) against what the reference's own script
produces on the same invocation (fixture g9, recorded from the reference): same number of finished episodes
and the same final state after 23 000 steps of its action stream."""

from __future__ import annotations

import importlib.util
import json
import os

import numpy as np
import pytest

from trace_util import ROOT, load_golden

pytestmark = pytest.mark.gpu


def _cli():
    spec = importlib.util.spec_from_file_location("bench_cli", os.path.join(ROOT, "scripts", "benchmark_multi_agent_env.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_reference_invocation_completes_the_same_number_of_episodes(tmp_path):
    """--env-name ReferenceModel-2-1 --num-agents 2 --deterministic --modes random --steps 20000 --warmup-steps 3000"""
    fx = load_golden("g9_benchmark_pin")
    mod = _cli()
    res = mod.main(["--env-name", "ReferenceModel-2-1", "--num-agents", "2", "--deterministic", "--modes", "random",
                    "--steps", "20000", "--warmup-steps", "3000", "--output-dir", str(tmp_path)])
    assert len(res) == 1 and res[0]["mode"] == "random"
    assert res[0]["episodes_completed"] == int(fx["episodes_completed"]) == 200
    assert res[0]["final_positions"] == fx["final_positions"].tolist()
    written = sorted(os.listdir(tmp_path))
    assert len(written) == 2 and written[0].endswith(".csv") and written[1].endswith(".json")
    payload = json.load(open(os.path.join(tmp_path, written[1])))
    assert payload[0]["env_config"]["env_name"] == "ReferenceModel-2-1" and payload[0]["steps"] == 20000
    header = open(os.path.join(tmp_path, written[0])).readline().strip().split(",")
    assert header == ["mode", "steps", "warmup_steps", "episodes_completed", "elapsed_s", "steps_per_s", "episodes_per_s",
                      "mean_step_ms"]


def test_vectorized_modes_and_throughput_assertion(tmp_path):
    mod = _cli()
    res = mod.main(["--num-envs", "256", "--steps", "300", "--warmup-steps", "50", "--modes", "random,masked",
                    "--output-dir", str(tmp_path), "--assert-min-steps-per-s", "5300"])
    # default config (2-1, 4 agents, 100-step episodes): every env truncates at steps 100, 200, 300 of 350
    assert res[0]["episodes_completed"] >= 2 * 256 and res[1]["episodes_completed"] >= 2 * 256
    assert res[0]["num_envs"] == 256 and res[0]["agent_steps_per_s"] == pytest.approx(res[0]["steps_per_s"] * 4)
    with pytest.raises(AssertionError, match="below required"):
        mod.main(["--num-envs", "8", "--steps", "50", "--warmup-steps", "5", "--modes", "random", "--output-dir",
                  str(tmp_path), "--assert-min-steps-per-s", "1e15"])


def test_fused_masked_mode_runs_the_policy_in_the_kernel(tmp_path):
    """--num-envs 256 --modes masked --fused 25: every launch advances 25 steps with the in-kernel masked-random policy."""
    mod = _cli()
    res = mod.main(["--env-name", "ReferenceModel-2-1", "--num-agents", "4", "--modes", "masked", "--steps", "500",
                    "--warmup-steps", "50", "--num-envs", "256", "--fused", "25", "--output-dir", str(tmp_path)])
    assert res[0]["mode"] == "masked" and res[0]["fused_steps_per_launch"] == 25 and res[0]["steps"] == 500
    assert res[0]["episodes_completed"] >= 256 * 4  # 500 steps of 100-step episodes: every env finished at least 4
