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


@pytest.mark.parametrize("master,cases,only_n,dense", [(2026, 40, None, None), (77, 30, 4, None), (4711, 30, None, "1"),
                                                       (1616, 25, 16, None), (1617, 15, 16, "1")])
def test_short_soak_matches_oracle(master, cases, only_n, dense, monkeypatch):
    """dense = "1": the 128-register build of the specialised step kernels (k_step's WPS = 4: no speculative slice
    loads, slot word fetched lazily), which mapf_create otherwise only picks for grids of more than three waves per
    SIMD (MAPF_FORCE_DENSE is read by mapf_create)."""
    from soak_specialized import run_soak
    if dense is not None:
        monkeypatch.setenv("MAPF_FORCE_DENSE", dense)
    lines = []
    err = run_soak(master, cases, only_n=only_n, log=lines.append)
    assert err is None, err + "\n" + "\n".join(lines[-8:])
