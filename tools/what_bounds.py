#!/usr/bin/env python3
"""Diagnostic: what a step of a workload costs with parts of its output switched off, and at other batch sizes -- which
part of the launch the time belongs to.  One hipGraph of 100 launches per variant, replayed; staggered episodes.

    python tools/what_bounds.py [workload] [key=value engine knobs ...]
Variants: all outputs | no observation (obs = NULL: the observation wave only keeps the barriers) | observation only (no
rewards / flags / info tensors).  Batches: 1/2x, 1x, 2x the workload's envs.
"""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from dl_reference_models_amd import workloads as wl
from dl_reference_models_amd.vec_env import VecReferenceModel

args = sys.argv[1:]
name = args[0] if args and "=" not in args[0] else wl.HEADLINE
knobs = dict(a.split("=", 1) for a in args if "=" in a)
b0 = wl.WORKLOADS[name][0]


def timed(b, variant):
    cfg = wl.workload_config(name, list(range(b)))
    cfg.update(knobs)
    env = VecReferenceModel(cfg)
    env.reset()
    n, spe = cfg["num_agents"], int(cfg["steps_per_episode"])
    c = env.get_state()["counters"]
    c[:, 0] = np.arange(b) % spe
    env.set_state(counters=c)
    acts = torch.from_numpy(np.random.default_rng(1).integers(0, 5, size=(100, b, n)).astype(np.int8)).to(env.device)
    base, stride = acts.data_ptr(), b * n
    obs = env._obs.data_ptr() if variant != "no_obs" else None
    rest = [env._rewards.data_ptr(), env._terminated.data_ptr(), env._truncated.data_ptr(), env._info_all.data_ptr(),
            env._info_agent.data_ptr()] if variant != "obs_only" else [None] * 5

    def launch(t, sp):
        rc = env._lib.mapf_step(env._h, base + (t % 100) * stride, obs, *rest, None, 1, sp)
        assert rc == 0, rc

    sp = torch.cuda.current_stream().cuda_stream
    for t in range(150):
        launch(t, sp)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        cp = torch.cuda.current_stream().cuda_stream
        for t in range(100):
            launch(t, cp)
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    env.poll_error()
    info = env.launch_info()
    return 1e3 * e0.elapsed_time(e1) / 1000, info


for b in (b0 // 2, b0, 2 * b0):
    for variant in ("all", "no_obs", "obs_only"):
        us, info = timed(b, variant)
        print(json.dumps({"workload": name, "envs": b, "variant": variant, "us_per_step": round(us, 3), "blocks": info["blocks"],
                          "threads": info["threads"], "lds_bytes": info["lds_bytes"], **knobs}), flush=True)
