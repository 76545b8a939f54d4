#!/usr/bin/env python3
"""How long a 20-launch window (one hipGraph of 20 c3 steps, fence to fence on the host clock) takes as a function of how
long the GPU sat idle before it.  Round 3: 122 us straight behind other launches, 131 us after 200 us of idleness, 138 us
after 1 ms -- which is why bench.py takes its "episodes before" count on the device and puts nothing but the fence
between its warm-up launches and the timed region.  Usage on the GPU box: python3 tools/idle_gap.py"""
import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from dl_reference_models_amd import workloads as wl
from dl_reference_models_amd.vec_env import VecReferenceModel
b = 8192
cfg = wl.workload_config(wl.HEADLINE, list(range(b)))
env = VecReferenceModel(cfg); env.reset()
n = cfg["num_agents"]
acts = torch.from_numpy(np.random.default_rng(999).integers(0, 5, size=(100, b, n)).astype(np.int8)).to(env.device)
base, stride = acts.data_ptr(), b * n
s = torch.cuda.Stream()
sp = s.cuda_stream
with torch.cuda.stream(s):
    for t in range(200): env.step_raw(base + (t % 100) * stride, sp, 1)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        for t in range(20): env.step_raw(base + t * stride, s.cuda_stream, 1)
    for gap in (0.0, 50e-6, 200e-6, 1e-3, 5e-3, 50e-3):
        r = []
        for rep in range(40):
            g.replay(); g.replay(); torch.cuda.synchronize()
            t1 = time.perf_counter()
            while time.perf_counter() - t1 < gap: pass
            t0 = time.perf_counter(); g.replay(); torch.cuda.synchronize(); r.append((time.perf_counter() - t0) * 1e6)
        print("idle gap %6.0f us before the timed graph(20): median %.1f us  min %.1f" % (gap * 1e6, np.median(r), min(r)), flush=True)
