#!/usr/bin/env python3
"""Secondary measurements quoted in DESIGN.md (not the driver's bench line): fused T-steps-per-launch throughput
("report both"), the reference-default observation layout (mask off, L = 28), a goal-biased action
stream (more deadlocks), and the c2 / c5 shapes.  One JSON object per line."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from dl_reference_models_amd import workloads as wl
from dl_reference_models_amd.vec_env import VecReferenceModel


def timed(fn, n):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3


def run(name, over=None, fused_T=100, p=None, tag=None, envs=None, stagger=False):
    b, h, w, n, density, _ = wl.WORKLOADS[name]
    if envs is not None:  # batch-size sweep: the same per-env workload, more envs on the one GPU
        b = envs
    cfg = wl.workload_config(name, list(range(b)))
    cfg.update(over or {})
    env = VecReferenceModel(cfg)
    env.reset()
    if stagger:  # every env starts its first episode at a different step count: ~1 % of the envs reset in EVERY step
        c = env.get_state()["counters"].copy()
        c[:, 0] = np.arange(b) % int(cfg["steps_per_episode"])
        env.set_state(counters=c)
    rng = np.random.default_rng(999)
    pool = 100
    if p is None:
        a = rng.integers(0, 5, size=(pool, b, n))
    else:
        a = rng.choice(5, size=(pool, b, n), p=p)
    acts = torch.from_numpy(a.astype(np.int8)).to(env.device)
    sptr = torch.cuda.current_stream().cuda_stream
    base, stride = acts.data_ptr(), b * n

    def single():
        for t in range(pool):
            env.step_raw(base + t * stride, sptr, 1)

    g = torch.cuda.CUDAGraph()
    single()
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        cptr = torch.cuda.current_stream().cuda_stream
        for t in range(pool):
            env.step_raw(base + t * stride, cptr, 1)
    for _ in range(3):
        g.replay()
    dt = timed(g.replay, 10)
    single_rate = b * n * pool * 10 / dt
    out = {"workload": tag or name, "envs": b, "agents": n, "obs_floats": env.obs_len,
           "specialized_kernel": env.launch_info()["specialized_kernel"],
           "single_step_per_launch": {"agent_steps_per_s": single_rate, "us_per_step": 1e6 * dt / (pool * 10)}}
    bytes_step = wl.algorithmic_bytes_per_env_step(n, env.obs_len, h, w) * b
    out["single_step_per_launch"]["roofline_frac"] = bytes_step / (dt / (pool * 10)) / 8e12
    for mode, label in ((2, "fused_obs_every_step"), (1, "fused_obs_last_only")):
        f = lambda: env.step_many(acts[:fused_T], obs_mode=mode, outputs=True)
        f(); f()
        dt = timed(f, 5)
        out[label] = {"T": fused_T, "agent_steps_per_s": b * n * fused_T * 5 / dt, "us_per_step": 1e6 * dt / (fused_T * 5)}
        if mode == 2:
            out[label]["roofline_frac"] = bytes_step / (dt / (fused_T * 5)) / 8e12
            # what a fused launch really moves per env-step: actions in, the observation row and the per-step outputs
            # out (rewards f32, info_agent 2 x u8 per agent; terminated, truncated, info_all 14 x f32 per env) -- state,
            # rows and counters stay in registers / LDS for the T steps, so the 8(d) figure above overstates the traffic
            fused_bytes = (n * (1 + 4 * env.obs_len + 4 + 2) + 2 + 4 * 14) * b
            out[label]["bytes_moved_per_env_step"] = fused_bytes // b
            out[label]["frac_of_8TBs_on_bytes_moved"] = fused_bytes / (dt / (fused_T * 5)) / 8e12
    if cfg.get("include_action_mask_in_obs", False):  # T fused steps with the in-kernel masked-random policy
        f = lambda: env.step_many_sampled(fused_T, seed=5)
        f(); f()
        dt = timed(f, 5)
        out["fused_sampled_policy"] = {"T": fused_T, "agent_steps_per_s": b * n * fused_T * 5 / dt, "us_per_step": 1e6 * dt / (fused_T * 5)}
    env.poll_error()
    st = env.get_state()["counters"]
    out["episodes_finished"] = int(st[:, 9].sum())
    print(json.dumps(out), flush=True)


def run_cte(b=8192, h=16, w=16, n=4, density=0.20, lanes=0):
    """Single-agent (CTE) sibling env: full-grid observation (H*W + 5N floats per env), one launch per step."""
    from dl_reference_models_amd.vec_env_single_agent import VecSingleAgentReferenceModel
    grids = wl.synthetic_grids(list(range(b)), h, w, density, n)
    env = VecSingleAgentReferenceModel({"num_envs": b, "num_agents": n, "grid": grids, "seeds": list(range(b)),
                                        "steps_per_episode": 100, "lanes_per_env": lanes})
    env.reset()
    pool = 100
    acts = torch.from_numpy(np.random.default_rng(999).integers(0, 5, size=(pool, b, n)).astype(np.int8)).to(env.device)

    def steps():
        for t in range(pool):
            env.step(acts[t], auto_reset=True)

    steps()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        steps()
    for _ in range(3):
        g.replay()
    dt = timed(g.replay, 10)
    us = 1e6 * dt / (pool * 10)
    # algorithmic bytes per env-step: observation row written, agent records + env scalars read and written,
    # obstacle rows and actions read, reward / flags / info written
    # (round 3: this env reads and writes the 8-byte hot plane only -- it keeps no lock history; rounds 1-2 counted 2 x 48 n)
    bytes_env = 4 * env.obs_len + 2 * 8 * n + 2 * 64 + 8 * h + n + 8 + 2 + 16
    out = {"workload": f"CTE single-agent env {b} x {h}x{w} x {n} agents (obs {env.obs_len} floats/env)",
           "envs": b, "agents": n, "obs_floats": env.obs_len, "lanes_per_env": lanes,
           "single_step_per_launch": {"env_steps_per_s": b * pool * 10 / dt, "agent_steps_per_s": b * n * pool * 10 / dt,
                                      "us_per_step": us, "algorithmic_bytes_per_env_step": bytes_env,
                                      "roofline_frac": bytes_env * b / (us * 1e-6) / 8e12}}
    for mode, label in ((2, "fused_obs_every_step"), (1, "fused_obs_last_only")):  # mapf_cte_step_many, T = 100
        f = lambda: env.step_many(acts, obs_mode=mode)
        f(); f()
        dtf = timed(f, 5)
        usf = 1e6 * dtf / (pool * 5)
        out[label] = {"T": pool, "us_per_step": usf, "env_steps_per_s": b * pool * 5 / dtf}
        if mode == 2:
            out[label]["roofline_frac"] = bytes_env * b / (usf * 1e-6) / 8e12
    env.poll_error()
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "sweep":
        # c4 is 65 536 envs over 8 GPUs; here the same envs on ONE GPU: with more than one wave pair per SIMD the
        # launch cost and the dependent chain of a step are hidden by other workgroups
        for envs in (16384, 32768, 65536):
            run("c3_8192x32x32_n8", tag=f"c3 per-env workload, {envs} envs on one GPU", envs=envs, fused_T=20)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "cte":
        for lanes in (0, 8, 16, 32, 64):
            run_cte(lanes=lanes)
        for lanes in (0, 16, 32, 64):
            run_cte(b=1024, h=32, w=32, n=8, density=0.2, lanes=lanes)
        sys.exit(0)
    run("c3_8192x32x32_n8")
    run("c3_8192x32x32_n8", {"include_action_mask_in_obs": False}, tag="c3 reference-default obs (mask off, L=28)")
    run("c3_8192x32x32_n8", p=[0.1, 0.1, 0.3, 0.4, 0.1], tag="c3 goal-biased action stream")
    run("c3_8192x32x32_n8", {"force_generic_kernel": True}, tag="c3 runtime-config kernel")
    run("c3_8192x32x32_n8", stagger=True, tag="c3 with staggered episodes (some env resets in every step)")
    run("c2_1024x16x16_n4")
    run("c5_1024x64x64_n64_lifelong")
    run("ref_training_4096x32x32_n16", tag="the reference's training setup (main.py: 16 agents, 7x7 windows), 4096 envs")
    run("ref_training_4096x32x32_n16", stagger=True, tag="the reference's training setup, staggered episodes")
