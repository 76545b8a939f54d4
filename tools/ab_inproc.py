#!/usr/bin/env python3
"""Paired A/B of the c3 step time over several builds of the library IN ONE PROCESS: one engine per build (the loader
caches per path), the same actions, graph replays of 100 launches timed with events, the builds taking turns round after
round -- differences of a percent that tools/ab2.sh (one process per build) loses in its run-to-run noise.  (The same library
listed twice differs by up to 0.7 % between its two engines -- where their buffers landed in memory -- so read
differences below one percent as "none".)
AB_WORKLOAD=<name> selects another workload of workloads.py (default: the headline); AB_GENERIC=1 the runtime-config kernels.
A build may be followed by engine knobs of the env config, `lib.so@key=value,key=value` (e.g. lib.so@small_group_observation=
table_walk): the same library, another kernel choice.
Usage on the GPU box: python3 tools/ab_inproc.py [--staggered] [--rounds 30] lib_a.so lib_b.so[@knob=value] ..."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from dl_reference_models_amd import workloads as wl

args = [a for a in sys.argv[1:] if not a.startswith("--")]
stag = "--staggered" in sys.argv
rounds = int(sys.argv[sys.argv.index("--rounds") + 1]) if "--rounds" in sys.argv else 30
libs = [a for a in args if a.split("@")[0].endswith(".so")]
name = os.environ.get("AB_WORKLOAD", wl.HEADLINE)
b = wl.WORKLOADS[name][0]
engines = []
for spec in libs:
    path, _, knobs = spec.partition("@")
    os.environ["MAPF_LIB"] = os.path.abspath(path)
    import importlib
    from dl_reference_models_amd import _lib, vec_env
    cfg = wl.workload_config(name, list(range(b)))
    if os.environ.get("AB_GENERIC"):  # the runtime-config kernels on the same shape
        cfg["force_generic_kernel"] = True
    for kv in filter(None, knobs.split(",")):
        k, _, v = kv.partition("=")
        cfg[k] = v
    env = vec_env.VecReferenceModel(cfg)
    env.reset()
    n, spe = cfg["num_agents"], int(cfg["steps_per_episode"])
    if stag:
        c = env.get_state()["counters"]
        c[:, 0] = np.arange(b) % spe
        env.set_state(counters=c)
    acts = torch.from_numpy(np.random.default_rng(999).integers(0, 5, size=(100, b, n)).astype(np.int8)).to(env.device)
    base, stride = acts.data_ptr(), b * n
    sp = torch.cuda.current_stream().cuda_stream
    for t in range(100): env.step_raw(base + t * stride, sp, 1)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        cp = torch.cuda.current_stream().cuda_stream
        for t in range(100): env.step_raw(base + t * stride, cp, 1)
    g.replay(); torch.cuda.synchronize()
    engines.append((spec, env, g, acts))
res = {p: [] for p in libs}
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for r in range(rounds):
    order = list(range(len(engines))) if r % 2 == 0 else list(reversed(range(len(engines))))
    for i in order:
        path, env, g, _ = engines[i]
        g.replay()
        e0.record()
        for _ in range(5): g.replay()
        e1.record(); torch.cuda.synchronize()
        res[path].append(1e3 * e0.elapsed_time(e1) / 500)
ref = np.array(res[libs[0]])
for p in libs:
    x = np.array(res[p])
    print("%-34s %s  %.4f us  +- %.4f (std of %d rounds)   vs first: %+.2f %%  (paired std %.2f %%)" % (
        p, "staggered" if stag else "in phase", x.mean(), x.std(), len(x), 100 * (x.mean() / ref.mean() - 1), 100 * (x / ref - 1).std()), flush=True)
for _, env, _, _ in engines: env.poll_error()
