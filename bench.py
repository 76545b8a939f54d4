#!/usr/bin/env python3
"""Headline benchmark: agent-steps/sec of the env step hot path (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W [--scaling weak|strong]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one mapf_step launch over one batch of envs (c3: 8192 envs x 32x32 x 8 agents per GPU, density
0.40, L = 33, lock metrics on, in-kernel auto-reset), with the actions already resident in HBM.  Envs shard
over the GPUs with no hot-path collective: `--scaling weak` (default) keeps 8192 envs per GPU, `--scaling
strong` splits a fixed total (c4: 65 536) over the ranks.  Started without torchrun and with --gpus N > 1 the
script launches one fresh child process per GPU itself (before it touches a GPU) and relays rank 0's line.
Timing with N > 1: every rank leaves a barrier + synchronize, starts its clock, launches its K steps, synchronizes and stops
its clock, then joins the closing barrier; the job's time is the MAXIMUM of the ranks' times (`per_rank_ms_per_step` lists
them), so the latency of the closing barrier's own collective is in nobody's K steps.

The timed region is what an RL loop sees in steady state: episode phases are STAGGERED (env b starts at step
b mod steps_per_episode), so about 1 % of the envs finish and are re-placed inside EVERY launch; `value` is
that number.  `value_synchronised` is the same run with all episodes in phase (resets in 1 launch of 100).
All K timed launches are hipGraph replays (graphs of min(K, --graph-steps) launches plus one for the remainder) unless
--graph-steps 0 or --launch-mode plain (plain C-ABI launches back to back; --launch-mode auto times min(K, 200) launches
both ways outside warm-up and timed region and takes the faster); `config.launch_mode` says which.

Rank 0 prints ONE JSON line.  Extra fields: `roofline` -- `frac` is SURVEY 8(d)'s quantity computed from the `value`
next to it (agent-steps/s x algorithmic bytes per agent-step / 8 TB/s), `frac_kernel` the same bytes over the DEVICE
time of the timed launches (HIP events), `traffic` the PMC-counted HBM bytes per launch from profiles/hbm_traffic.json --,
`per_call_ms` / `python_api_ms_per_step` (the step without a captured graph: plain C-ABI launches, and
VecReferenceModel.step() in a Python loop), `cpu_baseline` (the parity-checked C restatement, oracle/, one thread) and
`cpu_baseline_all_cores` (one env shard per host thread) -- reported baselines, not targets.

`--workload` selects any entry of dl_reference_models_amd.workloads.WORKLOADS (c2, c3, c5, the reference's training
setup, the single-agent env), `--fused T` the T-steps-per-launch kernels: every number in DESIGN.md's evidence table is
one `bench.py` command, so that rocprofv3 can be put in front of it (tools/collect_evidence.sh).
"""

from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
C4_TOTAL_ENVS = 65536  # BASELINE.json configs[3]


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=300)
    ap.add_argument("--pre-roll", type=int, default=-1,
                    help="untimed steps run when the staggered episode phases are set up, before warmup "
                         "(default: one episode length; 0 = measure the start-up transient)")
    ap.add_argument("--workload", default=None, help="one of dl_reference_models_amd.workloads.WORKLOADS")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="weak: the workload's envs per GPU; strong: --total-envs split over the GPUs")
    ap.add_argument("--total-envs", type=int, default=C4_TOTAL_ENVS, help="strong scaling: envs of the whole job")
    ap.add_argument("--launch-mode", choices=("graph", "plain", "auto"), default="graph",
                    help="how the K timed launches reach the GPU: hipGraph replays (default), plain C-ABI launches back to back, "
                         "or (auto) whichever of the two a calibration of min(K, 200) launches outside warm-up and timed region "
                         "finds faster for this workload and K; config.launch_mode says which.  (Measured, c3: at K = 2 000 the "
                         "two are within 1 %; at the driver's K = 20 both scatter between 6.0 and 7.5 us per step from run to "
                         "run -- a window of 0.12 ms is dominated by what surrounds it -- so auto is not the default.)")
    ap.add_argument("--graph-steps", type=int, default=100,
                    help="launches captured per hipGraph (0 = plain launches)")
    ap.add_argument("--episodes", choices=("staggered", "synchronised", "both"), default="both",
                    help="episode phases of the timed region(s); `value` is the staggered run unless "
                         "'synchronised' is asked for")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=8.0, help="target duration of each CPU baseline leg")
    ap.add_argument("--fused", type=int, default=0,
                    help="T > 0: launches of T env steps each (mapf_step_many / mapf_cte_step_many, observation written "
                         "every step) instead of one step per launch; --steps and --warmup stay in env steps")
    ap.add_argument("--api-steps", type=int, default=200,
                    help="env steps of the two secondary legs per_call_ms (plain C-ABI launches, no graph) and "
                         "python_api_ms_per_step (VecReferenceModel.step in a Python loop); 0 = skip them")
    ap.add_argument("--kernel-samples", type=int, default=200,
                    help="launches timed one by one with events (isolated launch duration)")
    ap.add_argument("--lanes-per-env", type=int, default=0,
                    help="diagnostic: override the engine's choice of lanes per env (config.lanes_per_env reports what ran)")
    ap.add_argument("--share-gpu", action="store_true",
                    help="rehearsal only: ranks share the visible GPUs round-robin and meet over gloo "
                         "(the line is marked shared_gpu; not a scaling measurement)")
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------------------------
# parent: python bench.py --gpus N without torchrun -> one fresh child per GPU
# ------------------------------------------------------------------------------------------------------
def _kfd_gpu_nodes():
    """GPU nodes of the KFD topology, in node order, as dicts of their properties -- read from sysfs, so the parent
    never opens the HIP runtime before it starts its ranks (a process that has initialised the GPU must not spawn /
    exec the workers of this pool).  None when the topology is not readable."""
    root = "/sys/class/kfd/kfd/topology/nodes"
    try:
        names = sorted(os.listdir(root), key=lambda n: int(n))
    except (OSError, ValueError):
        return None
    nodes = []
    for n in names:
        try:
            props = dict(line.split()[:2] for line in open(os.path.join(root, n, "properties")) if len(line.split()) >= 2)
        except OSError:
            continue  # (a node this cgroup may not read is not ours)
        if int(props.get("simd_count", "0")) > 0:
            nodes.append(props)
    return nodes


def visible_gpu_count() -> int:
    """GPUs this job may use, WITHOUT initialising HIP in this process: the *_VISIBLE_DEVICES list if one is set, else
    the KFD topology in sysfs, else a disposable child process that asks torch and exits."""
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            return len([x for x in v.split(",") if x.strip() != ""])
    nodes = _kfd_gpu_nodes()
    if nodes is not None:
        return len(nodes)
    try:
        out = subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"],
                             capture_output=True, text=True, timeout=300)
        return int(out.stdout.strip().splitlines()[-1])
    except (subprocess.SubprocessError, ValueError, IndexError):
        return 0


def _cpulist(text: str):
    cpus = []
    for part in text.strip().split(","):
        if "-" in part:
            a, b = part.split("-")
            cpus.extend(range(int(a), int(b) + 1))
        elif part:
            cpus.append(int(part))
    return cpus


def rank_cpu_sets(world: int):
    """One CPU set per local rank: the cores of the rank's GPU's NUMA node (PCI address from the KFD topology) that this
    process may run on, divided among the ranks that share the node; an even split of the allowed cores when the
    topology does not say.  SURVEY 8(e): every GPU's launch loop gets host threads of its own, next to its GPU."""
    allowed = sorted(os.sched_getaffinity(0))
    even = [allowed[r * len(allowed) // world:(r + 1) * len(allowed) // world] or allowed for r in range(world)]
    nodes = _kfd_gpu_nodes()
    if not nodes or len(nodes) < world or any(os.environ.get(v) for v in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES")):
        return even
    local = []
    for props in nodes[:world]:
        try:
            loc, dom = int(props["location_id"]), int(props.get("domain", "0"))
            bdf = f"{dom:04x}:{(loc >> 8) & 0xFF:02x}:{(loc >> 3) & 0x1F:02x}.{loc & 7}"
            cpus = [c for c in _cpulist(open(f"/sys/bus/pci/devices/{bdf}/local_cpulist").read()) if c in set(allowed)]
        except (OSError, KeyError, ValueError):
            cpus = []
        local.append(tuple(cpus))
    if any(not c for c in local):
        return even
    sets = []
    for r in range(world):
        peers = [q for q in range(world) if local[q] == local[r]]  # ranks whose GPUs hang off the same NUMA node
        k, n = peers.index(r), len(peers)
        cpus = list(local[r])
        sets.append(cpus[k * len(cpus) // n:(k + 1) * len(cpus) // n] or cpus)
    return sets


def bind_rank_cpus(local_rank: int, local_world: int):
    """Pin this rank to its CPU set (MAPF_RANK_CPUS from the parent launcher, else computed here: under torchrun the
    ranks are somebody else's children).  Returns the set, or None when the platform has no affinity call."""
    try:
        spec = os.environ.get("MAPF_RANK_CPUS")
        cpus = _cpulist(spec) if spec else rank_cpu_sets(local_world)[local_rank]
        os.sched_setaffinity(0, cpus)
        return cpus
    except (AttributeError, OSError, IndexError, ValueError):
        return None


def launch_ranks(args, worker_argv=None) -> int:
    """Start args.gpus worker processes (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, exactly what torchrun
    would export, plus the rank's CPU set) and relay rank 0's JSON line.  Nothing in this process touches a GPU, before
    or after: the devices are counted from sysfs.  worker_argv: the command of one rank (default: this script with this
    command line); the CPU tests pass a gloo stand-in."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    if not args.share_gpu:
        have = visible_gpu_count()
        if have < args.gpus:
            print(f"[bench] --gpus {args.gpus} but this node exposes {have} GPU(s)", file=sys.stderr)
            return 1
    cpu_sets = rank_cpu_sets(args.gpus)
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(args.gpus), "LOCAL_WORLD_SIZE": str(args.gpus),
                    "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "HSA_ENABLE_IPC_MODE_LEGACY": "0",
                    "MAPF_RANK_CPUS": ",".join(str(c) for c in cpu_sets[r])})
        procs.append(subprocess.Popen(worker_argv or [sys.executable, os.path.abspath(__file__), *sys.argv[1:]], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    # a rank that dies early (bad device, OOM) would leave the others waiting in the rendezvous for ever: watch all of
    # them, and when one fails stop the rest (by pid) and fail
    import threading

    out_chunks = []
    reader = threading.Thread(target=lambda: out_chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    failed = None
    while True:
        codes = [p.poll() for p in procs]
        failed = next(((r, c) for r, c in enumerate(codes) if c not in (None, 0)), None)
        if failed or all(c == 0 for c in codes):
            break
        time.sleep(0.2)
    if failed:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=20)
            except subprocess.TimeoutExpired:
                p.kill()
    reader.join(timeout=20)
    out = b"".join(c for c in out_chunks if c)
    for line in out.decode().splitlines():  # rank 0's stdout may also carry library chatter (gloo prints there)
        if line.startswith("{"):
            print(line, flush=True)
        elif line.strip():
            print(line, file=sys.stderr)
    if failed:
        print(f"[bench] rank {failed[0]} exited with status {failed[1]}; the other ranks were stopped", file=sys.stderr)
        return 1
    return 0


# ------------------------------------------------------------------------------------------------------
# CPU baselines (oracle/: test infrastructure, used here only as the reported baseline)
# ------------------------------------------------------------------------------------------------------
def host_threads() -> int:
    """Threads this process may really use: affinity mask, capped by the cgroup cpu quota when there is one."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def _oracle_shard(name, env_ids, stagger):
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as orc

    from dl_reference_models_amd import workloads as wl

    cfg = wl.workload_config(name, env_ids)
    spe = int(cfg["steps_per_episode"])
    if wl.is_single_agent(name):
        batch = orc.OracleCteBatch(cfg["grid"], cfg, seeds=cfg["seeds"])
        batch.reset()
        if stagger:
            batch.set_step_counts([int(i) % spe for i in env_ids])
        return batch, cfg
    batch = orc.OracleBatch(cfg["grid"], cfg, seeds=cfg["seeds"])
    batch.reset()
    if stagger:
        for i, e in zip(env_ids, batch.envs):
            e.step_count = int(i) % spe
    return batch, cfg


def _oracle_run(batch, acts, steps):
    if hasattr(batch, "run"):  # (single-agent env: the whole loop is one C call)
        batch.run(acts, steps)
        return
    for t in range(steps):
        batch.step(acts[t % acts.shape[0]], auto_reset=True, outputs=True)


def cpu_baselines(name, env_ids, action_pool, seconds, stagger):
    """The C oracle (CPU restatement, parity-checked against the reference) on a bounded sample of the same
    workload (same grids, seeds, actions, episode phases): one thread, then one env shard per thread.  No thread is
    pinned: the shards run on whatever cores the scheduler gives the process (the usable count is in `sample`)."""
    from concurrent.futures import ThreadPoolExecutor

    res = {}
    n_cpu = os.cpu_count()
    phases = "staggered episodes" if stagger else "episodes in phase"
    # ---- one thread: a sample of up to 1024 envs, step count sized from a short probe
    ids1 = list(env_ids[:1024])
    batch, cfg = _oracle_shard(name, ids1, stagger)
    n = cfg["num_agents"]
    acts = np.ascontiguousarray(action_pool[:, : len(ids1), :])
    t0 = time.perf_counter()
    _oracle_run(batch, acts, 10)
    probe = (time.perf_counter() - t0) / 10
    steps = int(min(max(seconds / max(probe, 1e-6), 20), 5000))
    t0 = time.perf_counter()
    _oracle_run(batch, acts, steps)
    dt = time.perf_counter() - t0
    res["cpu_baseline"] = {
        "value": len(ids1) * n * steps / dt, "unit": "agent-steps/s", "cores": 1, "kind": "port",
        "sample": f"{len(ids1)} envs x {steps} steps of the same workload ({phases}), C restatement "
                  f"(oracle/), 1 thread of {n_cpu} host cpus, not pinned",
    }
    # ---- all cores: one shard of the batch per thread (the C step loop runs outside the GIL)
    threads = min(host_threads(), 128)
    per = max(len(env_ids) // threads, 8)
    shards = []
    for k in range(threads):
        ids = list(env_ids[k * per:(k + 1) * per])
        if ids:
            shards.append((ids, k * per))
    made = [(_oracle_shard(name, ids, stagger)[0], np.ascontiguousarray(action_pool[:, off:off + len(ids), :])) for ids, off in shards]
    steps_mt = int(min(max(seconds / max(probe * per / len(ids1), 1e-6), 20), 5000))
    with ThreadPoolExecutor(len(made)) as ex:
        t0 = time.perf_counter()
        list(ex.map(lambda ba: _oracle_run(ba[0], ba[1], steps_mt), made))
        dt = time.perf_counter() - t0
    total = sum(b.B for b, _ in made)
    res["cpu_baseline_all_cores"] = {
        "value": total * n * steps_mt / dt, "unit": "agent-steps/s", "cores": len(made), "kind": "port",
        "sample": f"{total} envs x {steps_mt} steps, one shard of {per} envs per thread, {len(made)} threads, not pinned "
                  f"(os.cpu_count() = {n_cpu}, usable = {host_threads()})",
    }
    return res


# ------------------------------------------------------------------------------------------------------
# what is launched: one of four runners behind the same few calls
# ------------------------------------------------------------------------------------------------------
class _MultiAgentRunner:
    """mapf_step (one env step per launch) or mapf_step_many (--fused T: T env steps per launch, observation every step)."""

    def __init__(self, name, env_ids, device, rank, args):
        import torch

        from dl_reference_models_amd import workloads as wl
        from dl_reference_models_amd.vec_env import VecReferenceModel

        self.torch = torch
        _b, self.h, self.w, self.n, self.density, _ = wl.WORKLOADS[name]
        self.env_ids = env_ids
        self.b = len(env_ids)
        cfg = wl.workload_config(name, env_ids)
        cfg["device"] = str(device)
        self.cfg = cfg
        self.env = VecReferenceModel(cfg)
        self.spe = int(cfg["steps_per_episode"])
        self.lifelong = bool(cfg.get("lifelong_mapf", False))
        self.obs_len = self.env.obs_len
        self.T = int(args.fused)
        self.steps_per_launch = self.T or 1
        self.bytes_per_env_step = wl.algorithmic_bytes_per_env_step(self.n, self.obs_len, self.h, self.w)
        # what a fused launch really moves per env-step: the action byte, the observation row and the per-step outputs
        # (state, rows and counters stay in registers / LDS for the T steps)
        self.bytes_moved_per_env_step = (self.n * (1 + 4 * self.obs_len + 4 + 2) + 2 + 4 * 14) if self.T else None
        # actions: uniform over {0..4}, generated once and resident in HBM (inputs, not part of the path)
        self.pool = (2 * self.T) if self.T else (max(args.graph_steps, 1) if args.graph_steps else 128)
        self.action_pool_np = np.random.default_rng(999 + rank).integers(0, 5, size=(self.pool, self.b, self.n)).astype(np.int8)
        self.action_pool = torch.from_numpy(self.action_pool_np).to(device)
        self._base, self._stride = self.action_pool.data_ptr(), self.b * self.n
        info = self.env.launch_info()
        three = info["threads"] == 192  # k_step3 (small groups) / k_stepw (64-lane groups): three waves per workgroup
        self.kernel = "k_step_many" if self.T else (("k_stepw" if info["lanes_per_env"] == 64 else "k_step3") if three else "k_step")
        if self.T:
            B, N, Lo, T = self.b, self.n, self.obs_len, self.T
            from dl_reference_models_amd import _lib as L
            self._out = [torch.empty((T, B, N, Lo), dtype=torch.float32, device=device),
                         torch.empty((T, B, N), dtype=torch.float32, device=device),
                         torch.empty((T, B), dtype=torch.uint8, device=device), torch.empty((T, B), dtype=torch.uint8, device=device),
                         torch.empty((T, B, L.INFO_ALL), dtype=torch.float32, device=device),
                         torch.empty((T, B, N, 2), dtype=torch.uint8, device=device)]
            self._out_ptrs = [t.data_ptr() for t in self._out]

    def launch(self, i, sptr):
        if not self.T:
            return self.env.step_raw(self._base + (i % self.pool) * self._stride, sptr, 1)
        o = self._out_ptrs
        return self.env._lib.mapf_step_many(self.env._h, self.T, self._base + (i % 2) * self.T * self._stride, o[0], 2, o[1],
                                            o[2], o[3], o[4], o[5], sptr)

    def python_call(self, i):
        """The same launch through the public Python API (tensor in, dict of tensors out)."""
        if not self.T:
            return self.env.step(self.action_pool[i % self.pool])
        k = (i % 2) * self.T
        return self.env.step_many(self.action_pool[k:k + self.T], obs_mode=2)

    def set_phases(self, staggered):
        from dl_reference_models_amd import _lib as L

        self.env.reset()
        if staggered:  # env with global index i is i mod steps_per_episode steps into its episode
            c = self.env.get_state()["counters"]
            c[:, L.CTR_STEP_COUNT] = np.asarray(self.env_ids, dtype=np.int64) % self.spe
            self.env.set_state(counters=c)

    def episodes_device(self, out):
        self.env.episode_sums_device(out)

    def episode_sums(self):
        return self.env.episode_sums()

    def poll_error(self):
        self.env.poll_error()

    def launch_info(self):
        return self.env.launch_info()


class _SingleAgentRunner:
    """The single-agent (CTE) sibling env: mapf_cte_step / mapf_cte_step_many."""

    def __init__(self, name, env_ids, device, rank, args):
        import torch

        from dl_reference_models_amd import workloads as wl
        from dl_reference_models_amd.vec_env_single_agent import VecSingleAgentReferenceModel

        self.torch = torch
        _b, self.h, self.w, self.n, self.density, _ = wl.WORKLOADS[name]
        self.env_ids = env_ids
        self.b = len(env_ids)
        cfg = wl.workload_config(name, env_ids)
        cfg["device"] = str(device)
        if args.lanes_per_env:
            cfg["lanes_per_env"] = args.lanes_per_env
        self.cfg = cfg
        self.env = VecSingleAgentReferenceModel(cfg)
        self.spe = int(cfg["steps_per_episode"])
        self.lifelong = False
        self.obs_len = self.env.obs_len
        self.T = int(args.fused)
        self.steps_per_launch = self.T or 1
        self.bytes_per_env_step = wl.cte_algorithmic_bytes_per_env_step(self.n, self.h, self.w)
        self.bytes_moved_per_env_step = (4 * self.obs_len + self.n + 8 + 2 + 16) if self.T else None
        self.pool = (2 * self.T) if self.T else (max(args.graph_steps, 1) if args.graph_steps else 128)
        self.action_pool_np = np.random.default_rng(999 + rank).integers(0, 5, size=(self.pool, self.b, self.n)).astype(np.int8)
        self.action_pool = torch.from_numpy(self.action_pool_np).to(device)
        self.kernel = "k_cte_step"
        if self.T:
            B, T = self.b, self.T
            self._out = [torch.empty((T, B, self.obs_len), dtype=torch.float32, device=device),
                         torch.empty((T, B), dtype=torch.float64, device=device),
                         torch.empty((T, B), dtype=torch.uint8, device=device), torch.empty((T, B), dtype=torch.uint8, device=device),
                         torch.empty((T, B, 4), dtype=torch.float32, device=device)]
            self._out_ptrs = [t.data_ptr() for t in self._out]

    def launch(self, i, sptr):
        e = self.env
        stride = self.b * self.n
        if not self.T:
            return e._lib.mapf_cte_step(e._h, self.action_pool.data_ptr() + (i % self.pool) * stride, e._obs.data_ptr(),
                                        e._reward.data_ptr(), e._terminated.data_ptr(), e._truncated.data_ptr(),
                                        e._info.data_ptr(), None, 1, sptr)
        o = self._out_ptrs
        return e._lib.mapf_cte_step_many(e._h, self.T, self.action_pool.data_ptr() + (i % 2) * self.T * stride, o[0], 2, o[1],
                                         o[2], o[3], o[4], sptr)

    def python_call(self, i):
        if not self.T:
            return self.env.step(self.action_pool[i % self.pool])
        k = (i % 2) * self.T
        return self.env.step_many(self.action_pool[k:k + self.T], obs_mode=2)

    def set_phases(self, staggered):
        self.env.reset()
        if staggered:
            self.env.set_step_counts(np.asarray(self.env_ids, dtype=np.int64) % self.spe)

    def episodes_device(self, out):  # (this env keeps no episode statistics on the device)
        return None

    def episode_sums(self):
        return None

    def poll_error(self):
        self.env.poll_error()

    def launch_info(self):
        return self.env.launch_info(fused=self.T > 1)


# ------------------------------------------------------------------------------------------------------
# worker: one rank = one GPU
# ------------------------------------------------------------------------------------------------------
def worker(args) -> int:
    import torch
    import torch.distributed as dist

    from dl_reference_models_amd import _lib as L
    from dl_reference_models_amd import sharding, workloads as wl
    from dl_reference_models_amd.vec_env import metrics_from_sums

    rank, local_rank, world = sharding.dist_env()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    rank_cpus = bind_rank_cpus(local_rank, int(os.environ.get("LOCAL_WORLD_SIZE", world))) if world > 1 else None
    n_dev = torch.cuda.device_count()
    if n_dev == 0 or (not args.share_gpu and local_rank >= n_dev):  # fail fast: the launcher stops the other ranks
        raise SystemExit(f"[bench] rank {rank}: no GPU for local rank {local_rank} ({n_dev} visible)")
    dev_index = local_rank % n_dev if args.share_gpu else local_rank
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    use_dist = world > 1 or "RANK" in os.environ  # under torchrun always go through RCCL, also at world size 1
    if use_dist:
        if args.share_gpu:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=device)  # nccl == RCCL on ROCm

    name = args.workload or wl.HEADLINE
    b_weak, h, w, n, density, _ = wl.WORKLOADS[name]
    if args.scaling == "strong":
        env_ids = list(sharding.shard_range(args.total_envs, world, rank))
        total_envs = args.total_envs
    else:
        env_ids = list(sharding.weak_range(b_weak, rank))
        total_envs = b_weak * world
    b_per = len(env_ids)
    run = (_SingleAgentRunner if wl.is_single_agent(name) else _MultiAgentRunner)(name, env_ids, device, rank, args)
    spl = run.steps_per_launch
    if args.steps % spl or args.warmup % spl:
        raise SystemExit(f"--fused {spl}: --steps and --warmup must be multiples of it")
    spe = run.spe
    stream = torch.cuda.current_stream(device)
    sptr = stream.cuda_stream

    def run_plain(k, ptr=sptr, first=0):
        """k launches through the C ABI (k * spl env steps)."""
        for i in range(first, first + k):
            rc = run.launch(i, ptr)
            if rc != 0:
                raise RuntimeError(f"launch failed: {rc}")

    graphs = {}  # launches per graph -> captured graph

    def graph_of(k):
        if k not in graphs:
            g = torch.cuda.CUDAGraph()
            cap_stream = torch.cuda.Stream(device)
            with torch.cuda.graph(g, stream=cap_stream):
                run_plain(k, torch.cuda.current_stream(device).cuda_stream)
            graphs[k] = g
        return graphs[k]

    launches = args.steps // spl
    G = min(args.graph_steps, max(launches, 1)) if args.graph_steps > 0 else 0

    def plan(k):
        """(graph, replays) pairs that make exactly k launches."""
        if G == 0 or k == 0:
            return []
        full, rem = divmod(k, G)
        return ([(graph_of(G), full)] if full else []) + ([(graph_of(rem), 1)] if rem else [])

    mode = {"chosen": "plain" if (G == 0 or args.launch_mode == "plain") else "graph", "requested": args.launch_mode}

    def run_launches(k, pl):
        if mode["chosen"] == "plain":
            run_plain(k)
        else:
            for g, reps in pl:
                for _ in range(reps):
                    g.replay()

    def fence():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize(device)

    def timed_region(staggered):
        run.set_phases(staggered)
        run_plain(3)  # code object resident before any capture
        torch.cuda.synchronize(device)
        # pre-roll (state preparation, like the reset above; not warmup and not timed; reported in config): by
        # default one episode length of steps straight before the warmup, so that a SHORT timed window (the driver's
        # --steps 20 is 0.12 ms of GPU time) sees the steady state of a long run -- every env has been through a reset
        # since its phase was set, the placements of the episodes ending in the window have been pre-drawn in the
        # background -- and a GPU that has been busy, instead of the start-up transient after the idle of the capture
        pre = args.pre_roll if args.pre_roll >= 0 else spe
        pre_l = (pre + spl - 1) // spl
        pl_w, pl_t, pl_p = plan(args.warmup // spl), plan(launches), plan(pre_l)  # capture happens here, outside the timed region
        if args.launch_mode == "auto" and G > 0 and "calibration_us_per_step" not in mode:
            # Launch-mode calibration (state preparation like the pre-roll: not warm-up, not timed): the same fence-to-fence
            # region as the timed one, min(K, 200) launches, once as graph replays and once as plain launches.  Every rank
            # takes the job's (slowest rank's) times, so all ranks choose alike.
            c = min(launches, 200)
            pl_c = plan(c)
            t_us = {}
            for m in ("graph", "plain", "graph", "plain"):
                mode["chosen"] = m
                fence()
                t0 = time.perf_counter()
                run_launches(c, pl_c)
                fence()
                dt = 1e6 * (time.perf_counter() - t0) / (c * spl)
                t_us[m] = min(t_us.get(m, dt), dt)
            if use_dist:
                tt = torch.tensor([t_us["graph"], t_us["plain"]], dtype=torch.float64, device=None if args.share_gpu else device)
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                t_us = {"graph": float(tt[0]), "plain": float(tt[1])}
            mode["chosen"] = "plain" if t_us["plain"] < t_us["graph"] else "graph"
            mode["calibration_us_per_step"] = {k: round(v, 4) for k, v in t_us.items()}
            mode["calibrated_on_launches"] = c
            run.poll_error()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record(stream)  # (torch creates the HIP event at its first record: not inside the timed region)
        ev1.record(stream)
        ep_before = torch.zeros(L.NUM_EPISODE_ACC, dtype=torch.int64, device=device)
        fence()
        # Nothing but the fence stands between the warm-up launches and the timed region: the episode count before the
        # region is taken ON THE DEVICE, behind the last warm-up launch (mapf_episode_stats_async), and read back after
        # the region.  (Round 3: the host-side read that used to sit here -- 393 KB over PCIe plus a sum -- left the GPU
        # idle for a millisecond, and a 20-launch window that starts on a GPU idle for >= 1 ms takes 138 us instead of
        # 122: tools/idle_gap.py.  It is the warm-up's job to have the GPU warm when the clock starts.)
        run_launches(pre_l, pl_p)
        run_launches(args.warmup // spl, pl_w)
        with torch.cuda.stream(stream):
            run.episodes_device(ep_before)
        fence()
        t0 = time.perf_counter()
        ev0.record(stream)
        run_launches(launches, pl_t)
        ev1.record(stream)
        # Closing side of the bracket: this rank's clock stops when ITS K steps are done (synchronize); the closing barrier comes
        # behind the clock read -- the job's time is the MAXIMUM over the ranks' times (all-reduced below), i.e. the moment the
        # slowest rank finished after the common start, and the latency of the barrier collective itself (an RCCL all-reduce
        # behind the launches: tens of microseconds, a fifth of a 20-step window) is not part of anybody's K steps.
        torch.cuda.synchronize(device)
        elapsed = time.perf_counter() - t0
        if use_dist:
            dist.barrier()
        elapsed_incl_barrier = time.perf_counter() - t0  # (reported beside it: ms_per_step_incl_closing_barrier)
        # HIP events on the launch stream over the timed region: device time per launch (graph replays leave no
        # host gap).  This is the kernel time of roofline.frac_kernel.
        kernel_ms = ev0.elapsed_time(ev1) / launches
        run.poll_error()
        sums = run.episode_sums()
        resets = None if sums is None else int(sums[L.ACC_EPISODES]) - int(ep_before[L.ACC_EPISODES].item())
        per_rank = [elapsed]
        if use_dist:
            tt = torch.zeros(world, dtype=torch.float64, device=None if args.share_gpu else device)
            tt[rank] = elapsed
            dist.all_reduce(tt, op=dist.ReduceOp.SUM)  # (every rank's own time; the job's time is their maximum)
            per_rank = [float(x) for x in tt.tolist()]
            elapsed = max(per_rank)
        return {"elapsed": elapsed, "kernel_ms": kernel_ms, "resets": resets, "per_rank": per_rank,
                "elapsed_incl_barrier": elapsed_incl_barrier}

    legs = {}
    if args.episodes in ("staggered", "both"):
        legs["staggered"] = timed_region(True)
    if args.episodes in ("synchronised", "both"):
        legs["synchronised"] = timed_region(False)
    head = legs["staggered"] if "staggered" in legs else legs["synchronised"]

    # ---- secondary legs, same state as the last timed region left (what a caller that does NOT capture a graph pays) ----
    # per_call_ms: plain launches through the C ABI, back to back, fence to fence (the host's launch rate or the GPU's
    # step time, whichever is slower); python_api_ms_per_step: the same through VecReferenceModel.step() -- argument
    # checks, ctypes marshalling, the dict of output tensors -- i.e. what a Python RL loop pays per env step before its
    # policy has run.  Both per ENV STEP (a fused launch counts its T steps).
    per_call_ms = python_api_ms = None
    if args.api_steps > 0:
        k = max(args.api_steps // spl, 1)
        run_plain(min(k, 8))
        fence()
        t0 = time.perf_counter()
        run_plain(k)
        fence()
        per_call_ms = 1e3 * (time.perf_counter() - t0) / (k * spl)
        for i in range(min(k, 8)):
            run.python_call(i)
        fence()
        t0 = time.perf_counter()
        for i in range(k):
            run.python_call(i)
        fence()
        python_api_ms = 1e3 * (time.perf_counter() - t0) / (k * spl)
        run.poll_error()

    # ---- secondary: event pairs around single launches (includes ~2 us of event/launch overhead) ----
    samples = []
    for i in range(args.kernel_samples if spl == 1 else 0):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        run.launch(i, sptr)
        e1.record(stream)
        samples.append((e0, e1))
    torch.cuda.synchronize(device)
    per_launch_ms = np.array([a.elapsed_time(b) for a, b in samples], dtype=np.float64)
    isolated_launch_ms = float(np.median(per_launch_ms)) if len(per_launch_ms) else None

    # off the timed path: episode statistics accumulated on the device, summed over ranks with ONE small
    # all-reduce (96 bytes; latency-bound, so one fused buffer; RCCL on GPUs)
    sums = run.episode_sums()
    episode_metrics = None
    if sums is not None:
        sums = sharding.all_reduce_stats(sums.astype(np.float64), device=device if (use_dist and not args.share_gpu) else None)
        episode_metrics = metrics_from_sums(sums, n, run.lifelong)

    agent_steps = total_envs * n * args.steps
    value = agent_steps / head["elapsed"]
    # Roofline: roofline.frac = agent_steps_per_s x algorithmic bytes per agent-step / 8 TB/s, on the `value`
    # this line reports, per GPU.  frac_kernel is the same bytes over the DEVICE time of the timed launches (HIP events on
    # the launch stream, no host-side cost): what the kernel itself reaches; rocprofv3's per-kernel average under
    # profiles/ is to be compared with kernel_ms.
    bytes_env_step = run.bytes_per_env_step
    bytes_per_launch = bytes_env_step * b_per * spl
    achieved = (value / world) * (bytes_env_step / n) / 1e9
    achieved_kernel = bytes_per_launch / (head["kernel_ms"] * 1e-3) / 1e9
    key = name + (f"+fused{spl}" if spl > 1 else "")
    traffic = None
    tfile = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    if os.path.exists(tfile) and b_per == b_weak:
        try:
            traffic = json.load(open(tfile)).get(key, {}).get("hbm_bytes_per_launch")
        except (OSError, ValueError):
            traffic = None

    full, rem = divmod(launches, G) if G else (0, 0)
    launch = ((f"hipGraph: {full} x {G} launches" + (f" + 1 x {rem}" if rem else "")) if mode["chosen"] == "graph"
              else f"{launches} plain launches through the C ABI, back to back")
    if spl > 1:
        launch += f", {spl} env steps per launch (fused, observation written every step)"
    phases = "staggered" if "staggered" in legs else "synchronised"
    result = {
        "metric": "agent-steps/sec at 8192 envs x 8 agents on 32x32 grid" if (name == wl.HEADLINE and b_per == b_weak and spl == 1)
        else f"agent-steps/sec ({key}, {b_per} envs per GPU)",
        "value": value,
        "unit": "agent-steps/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * head["elapsed"] / args.steps,
        "higher_is_better": True,
        "scaling": args.scaling,
        "vs_baseline": None,
        "dtype": "u8/i16 state, f32 observations",
        "data": "synthetic",
        "config": {
            "workload": key, "envs_per_gpu": b_per, "total_envs": total_envs, "grid": [h, w], "agents": n,
            "obstacle_density": density, "obs_floats": run.obs_len, "sensor_range": run.cfg.get("sensor_range"),
            "steps_per_episode": spe, "lock_metrics": not wl.is_single_agent(name), "auto_reset": "in-kernel",
            "episode_phases": phases, "resets_in_timed_region": head["resets"],
            "pre_roll_steps": args.pre_roll if args.pre_roll >= 0 else spe,
            "actions": "uniform{0..4}, device-resident", "launch": launch, "launch_mode": mode, "env_steps_per_launch": spl,
            "parallelism": f"env-sharded x{world}, no hot-path collective", **run.launch_info(),
        },
        "roofline": {
            "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
            "traffic_source": "profiles/hbm_traffic.json (rocprofv3 PMC passes of this kernel, committed; not measured in this run)" if traffic else None,
            "definition": "frac = value / n_gpus x algorithmic bytes per agent-step / peak (SURVEY 8(d)); "
                          "frac_kernel = algorithmic bytes per launch / kernel_ms (HIP events over the timed launches) / peak",
            "kernel": run.kernel, "kernel_ms": head["kernel_ms"], "achieved_kernel": achieved_kernel,
            "frac_kernel": achieved_kernel / HBM_PEAK_GBS, "isolated_launch_ms": isolated_launch_ms,
            "algorithmic_bytes_per_launch": bytes_per_launch, "algorithmic_bytes_per_agent_step": bytes_env_step / n,
        },
        "per_call_ms": per_call_ms,
        "python_api_ms_per_step": python_api_ms,
        "episodes_finished": None if sums is None else float(sums[L.ACC_EPISODES]),
        "episode_metrics": episode_metrics,
    }
    if run.bytes_moved_per_env_step:
        moved = run.bytes_moved_per_env_step * b_per * spl
        result["roofline"]["bytes_moved_per_launch"] = moved
        result["roofline"]["frac_kernel_on_bytes_moved"] = moved / (head["kernel_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS
    if wl.is_single_agent(name):
        result["env_steps_per_s"] = value / n
    if "synchronised" in legs and "staggered" in legs:
        s = legs["synchronised"]
        result["value_synchronised"] = agent_steps / s["elapsed"]
        result["ms_per_step_synchronised"] = 1e3 * s["elapsed"] / args.steps
        result["resets_in_timed_region_synchronised"] = s["resets"]
        result["roofline"]["kernel_ms_synchronised"] = s["kernel_ms"]
        result["roofline"]["frac_synchronised"] = (agent_steps / s["elapsed"] / world) * (bytes_env_step / n) / 1e9 / HBM_PEAK_GBS
        result["roofline"]["frac_kernel_synchronised"] = bytes_per_launch / (s["kernel_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS
    if world > 1:
        result["per_rank_ms_per_step"] = [1e3 * t / args.steps for t in head["per_rank"]]
        result["ms_per_step_incl_closing_barrier"] = 1e3 * head["elapsed_incl_barrier"] / args.steps  # (rank 0's clock)
        result["config"]["rank0_cpus"] = len(rank_cpus) if rank_cpus else None
    if args.share_gpu:
        result["shared_gpu"] = True
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            result.update(cpu_baselines(name, env_ids, run.action_pool_np, args.cpu_seconds, phases == "staggered"))
        print(json.dumps(result), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    return 0


def main():
    args = parse_args()
    if args.gpus > 1 and "RANK" not in os.environ:
        return launch_ranks(args)
    return worker(args)


if __name__ == "__main__":
    sys.exit(main())
