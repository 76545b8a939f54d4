"""The CPU restatement under AddressSanitizer + UBSan (oracle/Makefile: `make asan`): the sanitizer-equivalent of SURVEY
section 5 on the side that has one (GPU AddressSanitizer is not available on this pool; the device side has the
checking build, tests/test_soak_gpu.py).  A child process -- the sanitizer runtime has to be preloaded into the
interpreter -- replays recorded reference traces (finite, lifelong, 40 agents, tight grids, exactly 2N free cells) and the
micro-cases through the instrumented library; any heap / stack / global overflow or undefined shift aborts it."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)

CHILD = r"""
import sys
sys.path.insert(0, %(tests)r); sys.path.insert(0, %(oracle)r); sys.path.insert(0, %(root)r)
from trace_util import BATCH_FIXTURES, MICRO_CASES, OracleStepper, load_golden, replay_batch_trace, replay_micro_case
n = 0
for name in ("g2_c2_16x16_n4", "g3b_tight_6x7_n6", "g4b_lifelong_5x9_n10", "g10_n40_lifelong_11x13", "g11_f_equals_2n_4x6_n3",
             "g12_lifelong_f_equals_2n", "g8_widewin_5x5_n5"):
    replay_batch_trace(lambda g, c, **kw: OracleStepper(g, c, **kw), load_golden(name))
    n += 1
fx = load_golden("g5_micro_cases")
for case in MICRO_CASES:
    replay_micro_case(lambda g, c, **kw: OracleStepper(g, c, **kw), fx, case)
    n += 1
print("asan replay ok", n)
"""


def test_oracle_replays_reference_traces_under_address_and_ub_sanitizers():
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "asan"], check=True)
    so = os.path.join(ROOT, "oracle", "_build", "libmapf_oracle_asan.so")
    libasan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True, check=True).stdout.strip()
    assert os.path.exists(so) and os.path.exists(libasan)
    env = dict(os.environ, LD_PRELOAD=libasan, MAPF_ORACLE_SO=so, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    code = CHILD % {"tests": HERE, "oracle": os.path.join(ROOT, "oracle"), "root": ROOT}
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "asan replay ok" in r.stdout, (r.returncode, r.stdout[-2000:], r.stderr[-4000:])
