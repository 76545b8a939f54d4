#!/bin/bash
# c5 (1024 envs x 64x64 x 64 agents, lifelong) step time over alternative builds in ONE session: bash tools/ab_c5.sh [lib ...]
run() { timeout -k 10 200 python3 - <<PY
import json, os, sys, torch, numpy as np
sys.path.insert(0, os.getcwd())
from dl_reference_models_amd import workloads as wl
from dl_reference_models_amd.vec_env import VecReferenceModel
name = os.environ.get("AB_WORKLOAD", "c5_1024x64x64_n64_lifelong")
b = wl.WORKLOADS[name][0]
cfg = wl.workload_config(name, list(range(b)))
env = VecReferenceModel(cfg); env.reset()
n = cfg["num_agents"]
acts = torch.from_numpy(np.random.default_rng(999).integers(0, 5, size=(100, b, n)).astype(np.int8)).to(env.device)
base, stride = acts.data_ptr(), b * n
sp = torch.cuda.current_stream().cuda_stream
for t in range(100): env.step_raw(base + t * stride, sp, 1)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    cp = torch.cuda.current_stream().cuda_stream
    for t in range(100): env.step_raw(base + t * stride, cp, 1)
for _ in range(3): g.replay()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): g.replay()
e1.record(); torch.cuda.synchronize()
env.poll_error()
print("%-34s %s single %.3f us" % (os.environ.get("MAPF_LIB", "shipped"), name, e0.elapsed_time(e1)))
PY
}
unset MAPF_LIB; run || exit 1
for L in "$@"; do export MAPF_LIB=$L; run || exit 1; done
unset MAPF_LIB; run
