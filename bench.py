#!/usr/bin/env python3
"""Headline benchmark: agent-steps/sec of the env step hot path (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one mapf_step launch over one batch of envs (8192 envs x 32x32 x 8 agents per GPU,
density 0.40, L = 33, lock metrics on, in-kernel auto-reset), with the actions already resident in
HBM.  Per-GPU work is fixed as N grows (weak scaling); envs shard with no hot-path collective.
Rank 0 prints ONE JSON line.  Extra fields: `roofline` (algorithmic bytes / measured kernel time vs
the 8 TB/s HBM peak) and `cpu_baseline` (the parity-checked C restatement, oracle/, on this box's
host cores -- a reported baseline, not a target).
"""

from __future__ import annotations

import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=300)
    ap.add_argument("--workload", default=None, help="one of dl_reference_models_amd.workloads.WORKLOADS")
    ap.add_argument("--graph-steps", type=int, default=100,
                    help="steps captured per hipGraph (0 = plain launches)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-steps", type=int, default=700)
    ap.add_argument("--kernel-samples", type=int, default=200,
                    help="launches timed one by one with events for the roofline figure")
    return ap.parse_args()


def cpu_baseline(name, env_ids, steps, action_pool):
    """Time the C oracle (CPU restatement, parity-checked against the reference) single-threaded on a
    bounded sample of the same workload."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as orc

    from dl_reference_models_amd import workloads as wl

    cfg = wl.workload_config(name, env_ids)
    batch = orc.OracleBatch(cfg["grid"], cfg, seeds=cfg["seeds"])
    batch.reset()
    n = cfg["num_agents"]
    acts = action_pool[:, : len(env_ids), :]
    for t in range(5):
        batch.step(acts[t % acts.shape[0]], auto_reset=True, outputs=True)
    t0 = time.perf_counter()
    for t in range(steps):
        batch.step(acts[t % acts.shape[0]], auto_reset=True, outputs=True)
    dt = time.perf_counter() - t0
    return {
        "value": len(env_ids) * n * steps / dt, "unit": "agent-steps/s", "cores": 1, "kind": "port",
        "sample": f"{len(env_ids)} envs x {steps} steps of the same workload, C restatement (oracle/), 1 thread "
                  f"of {os.cpu_count()} host cpus",
    }


def main():
    args = parse_args()
    import torch
    import torch.distributed as dist

    from dl_reference_models_amd import sharding, workloads as wl
    from dl_reference_models_amd.vec_env import VecReferenceModel

    rank, local_rank, world = sharding.dist_env()
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.gpus > 1 and world == 1:
        raise SystemExit("for --gpus N > 1 launch with torch.distributed.run (one rank per GPU)")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    use_dist = world > 1 or "RANK" in os.environ  # under torchrun always go through RCCL, also at world size 1
    if use_dist:
        dist.init_process_group(backend="nccl", device_id=device)  # nccl == RCCL on ROCm

    name = args.workload or wl.HEADLINE
    b_per, h, w, n, density, _ = wl.WORKLOADS[name]
    env_ids = list(sharding.weak_range(b_per, rank))
    cfg = wl.workload_config(name, env_ids)
    cfg["device"] = str(device)
    env = VecReferenceModel(cfg)
    L = env.obs_len
    env.reset()

    # actions: uniform over {0..4}, generated once and resident in HBM (inputs, not part of the path)
    pool = max(args.graph_steps, 1) if args.graph_steps else 128
    action_pool_np = np.random.default_rng(999 + rank).integers(0, 5, size=(pool, b_per, n)).astype(np.int8)
    action_pool = torch.from_numpy(action_pool_np).to(device)
    stream = torch.cuda.current_stream(device)
    sptr = stream.cuda_stream
    step_raw = env.step_raw
    base, stride = action_pool.data_ptr(), b_per * n

    def run_plain(k0, k):
        for t in range(k0, k0 + k):
            rc = step_raw(base + (t % pool) * stride, sptr, 1)
            if rc != 0:
                raise RuntimeError(f"mapf_step failed: {rc}")

    graph = None
    use_graph = args.graph_steps > 0
    if use_graph:
        try:
            run_plain(0, 3)  # warm the code object before capture
            torch.cuda.synchronize(device)
            graph = torch.cuda.CUDAGraph()
            cap_stream = torch.cuda.Stream(device)
            with torch.cuda.graph(graph, stream=cap_stream):
                cptr = torch.cuda.current_stream(device).cuda_stream
                for t in range(args.graph_steps):
                    rc = step_raw(base + t * stride, cptr, 1)
                    if rc != 0:
                        raise RuntimeError(f"mapf_step failed during capture: {rc}")
        except Exception as exc:  # graph capture is an optimisation of the launch loop only
            print(f"[bench] hipGraph capture unavailable ({exc!r}); using plain launches", file=sys.stderr)
            graph, use_graph = None, False

    def run_steps(k):
        if graph is not None:
            full, rem = divmod(k, args.graph_steps)
            for _ in range(full):
                graph.replay()
            if rem:
                run_plain(0, rem)
        else:
            run_plain(0, k)

    def fence():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize(device)

    run_steps(args.warmup)
    fence()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record(stream)
    run_steps(args.steps)
    ev1.record(stream)
    fence()
    elapsed = time.perf_counter() - t0
    # HIP events on the launch stream over the timed region: device time per launch (kernel + the
    # back-to-back boundary; graph replays leave no host gap).  This is the roofline's kernel time.
    kernel_ms = ev0.elapsed_time(ev1) / args.steps
    env.poll_error()
    if use_dist:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    # ---- secondary: event pairs around single launches (includes ~2 us of event/launch overhead) ----
    samples = []
    for i in range(args.kernel_samples):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        step_raw(base + (i % pool) * stride, sptr, 1)
        e1.record(stream)
        samples.append((e0, e1))
    torch.cuda.synchronize(device)
    per_launch_ms = np.array([a.elapsed_time(b) for a, b in samples], dtype=np.float64)
    isolated_launch_ms = float(np.median(per_launch_ms)) if len(per_launch_ms) else None

    # off the timed path: episode statistics accumulated on the device, summed over ranks with ONE small
    # RCCL all-reduce (96 bytes; latency-bound, so one fused buffer)
    from dl_reference_models_amd.vec_env import metrics_from_sums

    sums = sharding.all_reduce_stats(env.episode_sums().astype(np.float64), device=device if use_dist else None)
    stats = [float(sums[0])]
    episode_metrics = metrics_from_sums(sums, n, bool(cfg.get("lifelong_mapf", False)))

    agent_steps = b_per * n * args.steps * world
    bytes_per_launch = wl.algorithmic_bytes_per_env_step(n, L, h, w) * b_per
    achieved = bytes_per_launch / (kernel_ms * 1e-3) / 1e9
    traffic = None
    tfile = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    if os.path.exists(tfile):
        try:
            traffic = json.load(open(tfile)).get(name, {}).get("hbm_bytes_per_launch")
        except Exception:
            traffic = None

    result = {
        "metric": "agent-steps/sec at 8192 envs x 8 agents on 32x32 grid" if name == wl.HEADLINE
        else f"agent-steps/sec ({name})",
        "value": agent_steps / elapsed,
        "unit": "agent-steps/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u8/i16 state, f32 observations",
        "data": "synthetic",
        "config": {
            "workload": name, "envs_per_gpu": b_per, "grid": [h, w], "agents": n, "obstacle_density": density,
            "obs_floats": L, "sensor_range": cfg["sensor_range"], "steps_per_episode": cfg["steps_per_episode"],
            "lock_metrics": True, "auto_reset": "in-kernel", "actions": "uniform{0..4}, device-resident",
            "launch": f"hipGraph x{args.graph_steps}" if graph is not None else "plain launches",
            "parallelism": f"env-sharded x{world}, no hot-path collective", **env.launch_info(),
        },
        "roofline": {
            "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
            "kernel": "k_step", "kernel_ms": kernel_ms, "isolated_launch_ms": isolated_launch_ms,
            "algorithmic_bytes_per_launch": bytes_per_launch,
        },
        "episodes_finished": float(stats[0]),
        "episode_metrics": episode_metrics,
    }
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(name, env_ids, args.cpu_steps, action_pool_np)
        print(json.dumps(result), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
