#!/usr/bin/env python3
"""Development check of the fused launches' tail pre-draw (k_step_many): fused launches of several lengths against the
ORACLE stepped one step at a time, episode phases staggered, single steps in between (so that slots drawn by the tail
are consumed by single-step launches and slices started by single steps meet a fused launch), the visible generator
words after every launch; then timing of fused c3 in phase / staggered.
Usage on the GPU box: MAPF_LIB=build_diag/libdev.so python3 tools/dev_fused.py [parity|time|all]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from trace_util import EngineStepper, OracleStepper, _eq, synth_grids
from dl_reference_models_amd import workloads as wl
from dl_reference_models_amd.vec_env import VecReferenceModel

what = sys.argv[1] if len(sys.argv) > 1 else "all"
if what in ("parity", "all"):
    for (B, H, W, N, dens, spe) in [(130, 32, 32, 8, 0.4, 50), (67, 16, 16, 8, 0.2, 5), (64, 10, 10, 8, 0.1, 1), (40, 12, 9, 8, 0.1, 13)]:
        cfg = {"env_name": "synthetic", "num_agents": N, "sensor_range": 2, "include_action_mask_in_obs": True, "steps_per_episode": spe}
        grids = synth_grids(B, H, W, dens, N, base_seed=33_000)
        seeds = list(range(900, 900 + B))
        eng, orc = EngineStepper(grids, cfg, seeds=seeds), OracleStepper(grids, cfg, seeds=seeds)
        _eq("reset", eng.reset(), orc.reset())
        counts = np.arange(B) % spe
        eng.set_step_counts(counts); orc.set_step_counts(counts)
        rng = np.random.default_rng(3)
        for rep, T in enumerate((7, 50, 1, 130, 3, 64, 20)):
            acts = rng.integers(0, 5, size=(T, B, N)).astype(np.int8)
            out = eng.env.step_many(torch.from_numpy(acts).to(eng.env.device), obs_mode=2)
            out = {k: v.cpu().numpy() for k, v in out.items()}
            for t in range(T):
                r = orc.step(acts[t])
                for k in ("obs", "rewards", "terminated", "truncated", "info_all", "info_agent"):
                    _eq(f"fused {k} rep {rep}", out[k][t], r[k], t)
            _eq("rng words", eng.rng_words(), orc.rng_words(), rep)
            _eq("positions", eng.positions(), orc.positions(), rep)
            _eq("goals", eng.goals(), orc.goals(), rep)
            for t in range(int(rng.integers(0, 12))):  # single steps in between
                a1 = rng.integers(0, 5, size=(B, N)).astype(np.int8)
                ra, rb = eng.step(a1), orc.step(a1)
                for k in ("obs", "rewards", "terminated", "truncated", "info_all", "info_agent"):
                    _eq(f"single {k} rep {rep}", ra[k], rb[k], t)
            _eq("rng words after singles", eng.rng_words(), orc.rng_words(), rep)
        eng.env.poll_error()
        print("parity ok", (B, H, W, N, dens, spe), flush=True)

if what in ("time", "all"):
    name = "c3_8192x32x32_n8"
    b = wl.WORKLOADS[name][0]
    cfg = wl.workload_config(name, list(range(b)))
    n, spe = cfg["num_agents"], int(cfg["steps_per_episode"])
    out = {}
    for label, stagger in (("fused_in_phase_us", False), ("fused_staggered_us", True)):
        env = VecReferenceModel(cfg)
        env.reset()
        if stagger:
            c = env.get_state()["counters"]
            c[:, 0] = np.arange(b) % spe
            env.set_state(counters=c)
        acts = torch.from_numpy(np.random.default_rng(999).integers(0, 5, size=(100, b, n)).astype(np.int8)).to(env.device)
        f = lambda: env.step_many(acts, obs_mode=2, outputs=True)
        f(); f(); f()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(5): f()
        e1.record(); torch.cuda.synchronize()
        out[label] = 1e3 * e0.elapsed_time(e1) / 500
        env.poll_error()
    print(json.dumps(out), flush=True)
