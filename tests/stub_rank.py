"""Stand-in for one rank of `bench.py --gpus N` in the CPU tests of the launcher (tests/test_launcher_gloo.py): same
rendezvous (torchrun variables, gloo), same shard of the workload (global env ids), same timing protocol (barrier, K
steps, clock read, barrier, every rank's time gathered, the job's time = their maximum, rank 0 prints ONE JSON line) -- with the C
oracle as the stepper, because the HIP engine needs a GPU.  STUB_DIE_RANK=<r>: that rank exits with status 3 before the
rendezvous (the launcher must stop the others and fail)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import bench  # noqa: E402
from dl_reference_models_amd import sharding, workloads as wl  # noqa: E402
from trace_util import OracleStepper  # noqa: E402


def main():
    args = bench.parse_args()
    rank, local_rank, world = sharding.dist_env()
    if os.environ.get("STUB_DIE_RANK") == str(rank):
        sys.exit(3)
    cpus = bench.bind_rank_cpus(local_rank, int(os.environ.get("LOCAL_WORLD_SIZE", world)))
    dist.init_process_group(backend="gloo")
    per, total = 6, 12  # envs per rank (weak) / of the whole job (strong)
    env_ids = list(sharding.shard_range(total, world, rank) if args.scaling == "strong" else sharding.weak_range(per, rank))
    cfg = {"env_name": "synthetic", "num_agents": 4, "sensor_range": 2, "steps_per_episode": 20, "include_action_mask_in_obs": True}
    st = OracleStepper(wl.synthetic_grids(env_ids, 16, 16, 0.20, 4), cfg, seeds=[int(i) for i in env_ids])
    st.reset()
    acts = np.random.default_rng(999 + rank).integers(0, 5, size=(8, len(env_ids), 4)).astype(np.int8)
    for t in range(args.warmup):
        st.step(acts[t % 8])
    dist.barrier()
    t0 = time.perf_counter()
    for t in range(args.steps):
        st.step(acts[t % 8])
    elapsed = time.perf_counter() - t0  # (this rank's own K steps; the closing barrier comes behind the clock read, as in bench.py)
    dist.barrier()
    tt = torch.zeros(world, dtype=torch.float64)
    tt[rank] = elapsed
    dist.all_reduce(tt)
    envs = torch.tensor([float(len(env_ids))], dtype=torch.float64)
    dist.all_reduce(envs)
    if rank == 0:
        print(json.dumps({"n_gpus": world, "steps": args.steps, "scaling": args.scaling, "total_envs": int(envs.item()),
                          "value": envs.item() * 4 * args.steps / float(tt.max()),
                          "per_rank_ms_per_step": [1e3 * float(x) / args.steps for x in tt.tolist()],
                          "rank0_cpus": len(cpus) if cpus else None, "first_env_of_rank0": env_ids[0]}), flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
