"""A short randomized soak of the specialised kernels' episode boundaries (pre-drawn placement slots, the sliced
background draw in the observation wave, fast and slow resets) against the oracle: random grids, batch sizes with
ragged waves, episode lengths down to 1, staggered phases, goal-seeking actions so that episodes also end by success.
The cases run in ONE sequence on purpose: round 2 found a store of idle lane groups into a neighbouring group's draw
scratch (draw_shuffle16 without its `on` guard) that only showed with what earlier launches had left in LDS; master
seed 2026 hit it at case 18, seed 77 with N = 4 at case 210.  tools/soak_specialized.py runs the long version."""
import os
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("master,cases,only_n,dense,check", [
    (2026, 40, None, None, False), (77, 30, 4, None, False), (4711, 30, None, "1", False), (1616, 25, 16, None, False),
    (1617, 15, 16, "1", False), (2027, 30, None, None, True), (2028, 15, None, "1", True), (2029, 20, 16, None, True)])
def test_short_soak_matches_oracle(master, cases, only_n, dense, check, monkeypatch):
    """dense = "1": the 128-register two-wave build of the specialised step kernels (k_step's WPS = 4: no speculative
    slice loads, slot word fetched lazily), which mapf_create otherwise only picks for grids of more than three waves per
    SIMD (`register_budget` engine knob = MAPF_FLAG_FORCE_DENSE); without it the small-group shapes run the three-wave kernel.
    check: the same soak on the CHECKING build (-DMAPF_CHECK: every data-dependent LDS scatter / gather index of the
    draw, the inline reset and the staging rows is range-checked against its lane group's region and latches
    MAPF_ERR_INTERNAL): the build that would have reported round 2's draw_shuffle16 bug at the store instead of 200 cases
    later as a wrong goal cell."""
    from soak_specialized import run_soak
    if check:
        monkeypatch.setenv("MAPF_CHECK_BUILD", "1")
    lines = []
    err = run_soak(master, cases, only_n=only_n, log=lines.append, poll_errors=check,
                   knobs={"register_budget": "dense"} if dense else None)
    assert err is None, err + "\n" + "\n".join(lines[-8:])


def test_checking_build_latches_an_index_outside_its_region(monkeypatch):
    """The checking build is not a no-op: a scratch region deliberately declared too small (MAPF_CHECK_SHRINK, read by
    mapf_create of the checking build only) trips site 10 at the first inline reset."""
    import numpy as np
    import torch

    from dl_reference_models_amd import _lib as L
    from dl_reference_models_amd.vec_env import VecReferenceModel
    from trace_util import synth_grids

    monkeypatch.setenv("MAPF_CHECK_BUILD", "1")
    monkeypatch.setenv("MAPF_CHECK_SHRINK", "1")
    B, n = 16, 8
    env = VecReferenceModel({"env_name": "synthetic", "num_agents": n, "sensor_range": 2, "steps_per_episode": 2, "num_envs": B,
                             "include_action_mask_in_obs": True, "grid": synth_grids(B, 10, 10, 0.1, n), "seeds": list(range(B))})
    env.reset()
    a = torch.zeros((B, n), dtype=torch.int8, device=env.device)
    with pytest.raises(RuntimeError, match="checking build"):
        for t in range(6):
            env.step(a)
            env.poll_error()


@pytest.mark.parametrize("master,cases,dense,sampler,only_n", [(2030, 30, None, False, None), (2031, 15, "1", False, None),
                                                              (2032, 15, None, True, None), (2033, 15, None, False, 16)])
def test_short_soak_of_the_runtime_config_kernels(master, cases, dense, sampler, only_n, monkeypatch):
    """The same soak with the runtime-config kernels forced onto the prebuilt shapes (SOAK_GENERIC): full groups of 4 / 8
    agents draw in slices and run the three-wave kernel (KRuntimeSliced), its two-wave 128-register build (dense), or --
    `background_draw: "sampler_workgroups"` -- the sampler workgroups they used before round 3."""
    from soak_specialized import run_soak
    monkeypatch.setenv("SOAK_GENERIC", "1")
    knobs = {}
    if dense is not None:
        knobs["register_budget"] = "dense"
    if sampler:
        knobs["background_draw"] = "sampler_workgroups"
    lines = []
    err = run_soak(master, cases, only_n=only_n, log=lines.append, knobs=knobs)
    assert err is None, err + "\n" + "\n".join(lines[-8:])
