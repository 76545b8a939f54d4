#!/usr/bin/env python3
"""Policy-in-the-loop cost (a measurement, not a component): what a rollout step costs when a policy
sits between two env steps, and which share of it is the env kernel.

One hipGraph of [mapf_step -> a 2-layer MLP on the observation -> masked argmax -> actions] x 20, replayed; next to it the
same graph without the env (policy only) and without the policy (env only, the actions of the last policy run).  The MLP
is the size class the reference configures for PPO (/root/reference/src/agents/ppo.py:67-75: 64 hidden units; its LSTM is
replaced by a second dense layer -- this is about launch and bandwidth cost, not about learning).  fp32 and bf16 weights.

    python tools/policy_loop.py [workload ...]        (default: the headline shape and the reference's training setup)
One JSON object per line.
"""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from dl_reference_models_amd import workloads as wl
from dl_reference_models_amd.vec_env import VecReferenceModel

K = 20  # env steps per graph (what the driver's bench window holds)


def timed(g, reps=50):
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / (reps * K)  # us per env step


def run(name, dtype):
    b, h, w, n, density, _ = wl.WORKLOADS[name]
    cfg = wl.workload_config(name, list(range(b)))
    env = VecReferenceModel(cfg)
    dev = env.device
    L = env.obs_len
    has_mask = bool(cfg.get("include_action_mask_in_obs", False))
    torch.manual_seed(0)
    w1 = (torch.randn(L, 64, device=dev) / np.sqrt(L)).to(dtype)
    b1 = torch.zeros(64, device=dev, dtype=dtype)
    w2 = (torch.randn(64, 5, device=dev) / 8.0).to(dtype)
    b2 = torch.zeros(5, device=dev, dtype=dtype)
    actions = torch.zeros((b, n), dtype=torch.int8, device=dev)
    obs = env.reset()
    c = env.get_state()["counters"]
    c[:, 0] = np.arange(b) % int(cfg["steps_per_episode"])  # staggered episode phases, as in bench.py
    env.set_state(counters=c)

    def policy(o):
        x = o.view(b * n, L)
        hdn = torch.tanh(torch.addmm(b1, x.to(dtype), w1))
        logits = torch.addmm(b2, hdn, w2).float()
        if has_mask:  # the reference's action-mask model: logits + clamp(log(mask)) (models/action_mask_model.py:52-64)
            logits = logits + torch.clamp(torch.log(x[:, L - 5:] + 1e-6), min=-1e9)
        actions.copy_(torch.argmax(logits, dim=1).to(torch.int8).view(b, n))

    def env_step():
        return env.step(actions)["obs"]

    for _ in range(3):  # warm up (rocBLAS picks its kernels outside the capture)
        policy(env_step())
    torch.cuda.synchronize()
    graphs = {}
    for kind in ("loop", "env_only", "policy_only"):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            o = env._obs
            for _ in range(K):
                if kind != "policy_only":
                    o = env_step()
                if kind != "env_only":
                    policy(o)
        graphs[kind] = g
    out = {"workload": name, "envs": b, "agents": n, "obs_floats": L, "policy": f"MLP {L}-64-5, {str(dtype).split('.')[-1]}, masked argmax" if has_mask else f"MLP {L}-64-5, {str(dtype).split('.')[-1]}, argmax",
           "steps_per_graph": K}
    for kind, g in graphs.items():
        out[kind + "_us_per_step"] = timed(g)
    env.poll_error()
    out["env_share_of_loop"] = out["env_only_us_per_step"] / out["loop_us_per_step"]
    out["agent_steps_per_s_in_loop"] = b * n / (out["loop_us_per_step"] * 1e-6)
    # the same loop without a graph: Python launches every kernel (what an eager RL loop pays)
    torch.cuda.synchronize()
    import time
    t0 = time.perf_counter()
    for _ in range(200):
        policy(env_step())
    torch.cuda.synchronize()
    out["eager_python_loop_us_per_step"] = 1e6 * (time.perf_counter() - t0) / 200
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    names = sys.argv[1:] or [wl.HEADLINE, "ref_training_4096x32x32_n16"]
    for nm in names:
        for dt in (torch.float32, torch.bfloat16):
            run(nm, dt)
