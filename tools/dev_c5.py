#!/usr/bin/env python3
"""Development check for the 64-lane group paths (N = 33 .. 64): parity against the oracle over many episode ends
(inline draws of 2N = 66 .. 128 cells, also on small populations where Floyd's collisions are frequent), then timings of
what VERDICT r2 item 2 names: the reset kernel at c5's shape, c5 single step, c5 with phases staggered by hand.
Usage on the GPU box: MAPF_LIB=build_diag/libc5.so python3 tools/dev_c5.py [parity|time|all]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from trace_util import EngineStepper, OracleStepper, compare_steppers, synth_grids
from dl_reference_models_amd import workloads as wl
from dl_reference_models_amd.vec_env import VecReferenceModel

what = sys.argv[1] if len(sys.argv) > 1 else "all"
if what in ("parity", "all"):
    cases = [  # B, H, W, N, density, steps_per_episode, T, extra
        (24, 64, 64, 64, 0.20, 3, 40, {}),
        (16, 12, 12, 64, 0.0, 2, 40, {}),     # F = 144, 2N = 128: almost every draw of Floyd's lies in the j range
        (16, 12, 11, 64, 0.0, 1, 30, {}),     # F = 132
        (16, 9, 9, 33, 0.1, 2, 40, {}),       # F ~ 73, 2N = 66
        (20, 40, 37, 48, 0.15, 5, 60, {"lifelong_mapf": True}),
        (9, 30, 30, 40, 0.3, 4, 50, {"include_action_mask_in_obs": False}),
    ]
    cases += [(8, 6, 7, 6, 0.1, 3, 40, {"lanes_per_env": 64}), (8, 5, 9, 10, 0.1, 4, 40, {"lanes_per_env": 64, "lifelong_mapf": True})]
    for (B, H, W, N, dens, spe, T, extra) in cases:
        cfg = {"env_name": "synthetic", "num_agents": N, "sensor_range": 2, "include_action_mask_in_obs": True, "steps_per_episode": spe}
        extra = dict(extra)
        kw = {"lanes_per_env": extra.pop("lanes_per_env")} if "lanes_per_env" in extra else {}
        cfg.update(extra)
        grids = synth_grids(B, H, W, dens, N, base_seed=90_000)
        acts = np.random.default_rng(5).integers(0, 5, size=(T, B, N)).astype(np.int8)
        seeds = list(range(700, 700 + B))
        eng = EngineStepper(grids, cfg, seeds=seeds, **kw)
        compare_steppers(eng, OracleStepper(grids, cfg, seeds=seeds), acts)
        eng.env.poll_error()
        print("parity ok", (B, H, W, N, dens, spe, T, extra), flush=True)

if what in ("time", "all"):
    name = "c5_1024x64x64_n64_lifelong"
    b = wl.WORKLOADS[name][0]
    cfg = wl.workload_config(name, list(range(b)))
    env = VecReferenceModel(cfg)
    n, spe = cfg["num_agents"], int(cfg["steps_per_episode"])
    acts = torch.from_numpy(np.random.default_rng(999).integers(0, 5, size=(100, b, n)).astype(np.int8)).to(env.device)
    base, stride = acts.data_ptr(), b * n

    def timed(fn, reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        return 1e3 * e0.elapsed_time(e1) / reps

    env.reset()
    out = {"reset_kernel_us": timed(env.reset, 20)}
    for label, stagger in (("single_us", False), ("staggered_by_hand_us", True)):
        env.reset()
        if stagger:
            c = env.get_state()["counters"]
            c[:, 0] = np.arange(b) % spe
            env.set_state(counters=c)
        sp = torch.cuda.current_stream().cuda_stream
        for t in range(100): env.step_raw(base + t * stride, sp, 1)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            cp = torch.cuda.current_stream().cuda_stream
            for t in range(100): env.step_raw(base + t * stride, cp, 1)
        g.replay()
        out[label] = timed(g.replay, 10) / 100
        env.poll_error()
    env.reset()
    f = lambda: env.step_many(acts, obs_mode=2, outputs=True)
    f(); f()
    out["fused_obs_every_step_us"] = timed(f, 5) / 100
    env.poll_error()
    print(json.dumps(out), flush=True)
