"""`python bench.py --gpus N` launches its own ranks (bench.launch_ranks).  No multi-GPU node is available to these
tests, so the launcher is driven end to end on the CPU: two gloo ranks (tests/stub_rank.py: the oracle as the stepper,
the same rendezvous variables, shard ranges and timing protocol as a real rank), weak and strong, asserting one JSON
line and status 0 -- and status 1 with every child gone when a rank dies before the rendezvous (the other rank would
otherwise wait for ever).  Also the host-only pieces: device count without touching HIP, per-rank CPU sets."""
import json
import os
import subprocess
import sys
import time

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

import bench  # noqa: E402

STUB = os.path.join(HERE, "stub_rank.py")


def _launch(capsys, monkeypatch, extra, die=None):
    argv = ["--gpus", "2", "--share-gpu", "--steps", "6", "--warmup", "2", *extra]
    if die is not None:
        monkeypatch.setenv("STUB_DIE_RANK", str(die))
    args = bench.parse_args(argv)
    rc = bench.launch_ranks(args, worker_argv=[sys.executable, STUB, *argv])
    out = capsys.readouterr()
    return rc, out


@pytest.mark.parametrize("scaling", ["weak", "strong"])
def test_launcher_runs_two_gloo_ranks_and_relays_one_json_line(scaling, capsys, monkeypatch):
    rc, out = _launch(capsys, monkeypatch, ["--scaling", scaling])
    assert rc == 0, out.err
    lines = [ln for ln in out.out.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.out
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == scaling and d["total_envs"] == 12 and d["first_env_of_rank0"] == 0
    assert len(d["per_rank_ms_per_step"]) == 2 and all(t > 0 for t in d["per_rank_ms_per_step"])
    assert d["value"] > 0 and d["rank0_cpus"] and d["rank0_cpus"] >= 1


def test_launcher_fails_and_leaves_no_child_behind_when_a_rank_dies(capsys, monkeypatch):
    started = set(_children())
    t0 = time.time()
    rc, out = _launch(capsys, monkeypatch, [], die=1)
    assert rc == 1 and "rank 1 exited with status 3" in out.err
    assert not [ln for ln in out.out.splitlines() if ln.startswith("{")]
    assert time.time() - t0 < 60
    assert set(_children()) <= started  # rank 0 (waiting in the rendezvous) was stopped, nobody is left


def _children():
    me = os.getpid()
    kids = []
    for pid in os.listdir("/proc"):
        if pid.isdigit():
            try:
                with open(f"/proc/{pid}/stat") as fh:
                    if int(fh.read().rsplit(")", 1)[1].split()[1]) == me:
                        kids.append(int(pid))
            except (OSError, ValueError, IndexError):
                pass
    return kids


def test_rank_cpu_sets_partition_the_allowed_cores():
    allowed = sorted(os.sched_getaffinity(0))
    for world in (1, 2, 3, 8):
        sets = bench.rank_cpu_sets(world)
        assert len(sets) == world and all(s and set(s) <= set(allowed) for s in sets)
        if len(allowed) >= world:  # enough cores: the ranks do not share any
            flat = [c for s in sets for c in s]
            assert len(flat) == len(set(flat))


def test_device_count_comes_from_the_environment_or_sysfs_without_touching_hip(monkeypatch):
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0,2,5")
    assert bench.visible_gpu_count() == 3
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "")
    assert bench.visible_gpu_count() == 0
    monkeypatch.delenv("HIP_VISIBLE_DEVICES")
    # (no KFD topology in the build container: the fallback is a child process that asks torch and exits)
    assert bench.visible_gpu_count() >= 0
    src = open(os.path.join(ROOT, "bench.py")).read()
    body = src[src.index("def launch_ranks"):src.index("# CPU baselines")]
    assert "import torch" not in body and "hip" not in body.lower().replace("hsa_enable", "")
