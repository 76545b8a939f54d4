"""Pins the CPU restatement of the single-agent (CTE) sibling env (reference
src/environments/reference_model_single_agent.py) to traces recorded from the unmodified reference.
The reference's own tests only check dtype/bounds of this env (tests/test_reference_model_observation_dtypes.py:44-56),
so the recorded traces (tests/golden/gs_cte_*.npz) are what pins parity here."""

import numpy as np
import pytest

from trace_util import CTE_FIXTURES, CteOracleStepper, load_golden, replay_cte_trace


@pytest.mark.parametrize("name", CTE_FIXTURES)
def test_cte_oracle_matches_golden_trace(name):
    fx = load_golden(name)
    stats = replay_cte_trace(CteOracleStepper, fx)
    assert stats["steps"] == fx["actions"].shape[0] and stats["resets"] >= 2


def test_cte_fixtures_cover_penalties_and_the_mask_quirk():
    fx = load_golden("gs_cte_6x7_n5_penalties")
    assert fx["info"][:, :, 0].sum() >= 10  # blocking penalties fired
    r = fx["reward"]
    assert np.any(np.abs(r * 20 - np.round(r * 20)) > 1e-9)  # rewards that are not multiples of 0.05: -0.3 / -0.07 in play
    assert (fx["terminated"].astype(bool) & ~fx["truncated"].astype(bool)).sum() >= 1
    # the reference's mask treats an obstacle (code 1, odd) as enterable (reference_model_single_agent.py:483-493)
    fx = load_golden("gs_cte_16x16_n8")
    g, N = fx["grids"][0], 8
    obs = fx["reset0_obs"][0]
    grid_codes = obs[: g.size].reshape(g.shape)
    mask = obs[g.size:].reshape(N, 5)
    found = False
    for i in range(N):
        (x,), (y,) = np.where(grid_codes == 2 * i + 2)[0][:1], np.where(grid_codes == 2 * i + 2)[1][:1]
        for k, (dx, dy) in enumerate([(-1, 0), (0, 1), (1, 0), (0, -1)], start=1):
            nx, ny = x + dx, y + dy
            if 0 <= nx < g.shape[0] and 0 <= ny < g.shape[1] and g[nx, ny] == 1:
                assert mask[i, k] == 1.0
                found = True
    assert found
