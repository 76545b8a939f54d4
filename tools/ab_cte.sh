run() { timeout -k 10 200 python3 - <<PY 2>&1 | grep -v amdgpu.ids
import os, sys, json, numpy as np, torch
sys.path.insert(0, os.getcwd())
from dl_reference_models_amd import workloads as wl
from dl_reference_models_amd.vec_env_single_agent import VecSingleAgentReferenceModel
for (b, h, w, n, lanes) in ((8192, 16, 16, 4, 8), (1024, 32, 32, 8, 64)):
    grids = wl.synthetic_grids(list(range(b)), h, w, 0.2, n)
    env = VecSingleAgentReferenceModel({"num_envs": b, "num_agents": n, "grid": grids, "seeds": list(range(b)), "steps_per_episode": 100, "lanes_per_env": lanes})
    env.reset()
    acts = torch.from_numpy(np.random.default_rng(999).integers(0, 5, size=(100, b, n)).astype(np.int8)).to(env.device)
    def steps():
        for t in range(100): env.step(acts[t], auto_reset=True)
    steps(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g): steps()
    for _ in range(3): g.replay()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): g.replay()
    e1.record(); torch.cuda.synchronize()
    print(os.environ.get("MAPF_LIB", "shipped"), b, h, w, n, "single %.3f us" % (e0.elapsed_time(e1)))
PY
}
unset MAPF_LIB; run; export MAPF_LIB=build_diag/libcte.so; run; unset MAPF_LIB; run
