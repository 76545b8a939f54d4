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

The timed region is what an RL loop sees in steady state: episode phases are STAGGERED (env b starts at step
b mod steps_per_episode), so about 1 % of the envs finish and are re-placed inside EVERY launch; `value` is
that number.  `value_synchronised` is the same run with all episodes in phase (resets in 1 launch of 100).
All K timed launches are hipGraph replays (graphs of min(K, --graph-steps) launches plus one for the
remainder) unless --graph-steps 0.

Rank 0 prints ONE JSON line.  Extra fields: `roofline` (algorithmic bytes / measured kernel time vs the
8 TB/s HBM peak), `cpu_baseline` (the parity-checked C restatement, oracle/, one thread) and
`cpu_baseline_all_cores` (one env shard per host thread) -- reported baselines, not targets.
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
    ap.add_argument("--graph-steps", type=int, default=100,
                    help="launches captured per hipGraph (0 = plain launches)")
    ap.add_argument("--episodes", choices=("staggered", "synchronised", "both"), default="both",
                    help="episode phases of the timed region(s); `value` is the staggered run unless "
                         "'synchronised' is asked for")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=8.0, help="target duration of each CPU baseline leg")
    ap.add_argument("--kernel-samples", type=int, default=200,
                    help="launches timed one by one with events (isolated launch duration)")
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
    batch = orc.OracleBatch(cfg["grid"], cfg, seeds=cfg["seeds"])
    batch.reset()
    if stagger:
        for i, e in zip(env_ids, batch.envs):
            e.step_count = int(i) % int(cfg["steps_per_episode"])
    return batch, cfg


def _oracle_run(batch, acts, steps):
    for t in range(steps):
        batch.step(acts[t % acts.shape[0]], auto_reset=True, outputs=True)


def cpu_baselines(name, env_ids, action_pool, seconds):
    """The C oracle (CPU restatement, parity-checked against the reference) on a bounded sample of the same
    workload (same grids, seeds, actions, staggered episode phases): one thread, then one env shard per thread."""
    from concurrent.futures import ThreadPoolExecutor

    res = {}
    n_cpu = os.cpu_count()
    # ---- one thread: a 1024-env sample, step count sized from a short probe
    ids1 = list(env_ids[:1024])
    batch, cfg = _oracle_shard(name, ids1, True)
    n = cfg["num_agents"]
    acts = action_pool[:, : len(ids1), :]
    t0 = time.perf_counter()
    _oracle_run(batch, acts, 10)
    probe = (time.perf_counter() - t0) / 10
    steps = int(min(max(seconds / max(probe, 1e-6), 20), 5000))
    t0 = time.perf_counter()
    _oracle_run(batch, acts, steps)
    dt = time.perf_counter() - t0
    res["cpu_baseline"] = {
        "value": len(ids1) * n * steps / dt, "unit": "agent-steps/s", "cores": 1, "kind": "port",
        "sample": f"{len(ids1)} envs x {steps} steps of the same workload (staggered episodes), C restatement "
                  f"(oracle/), 1 thread of {n_cpu} host cpus",
    }
    # ---- all cores: one shard of the batch per thread (the C step loop runs outside the GIL)
    threads = min(host_threads(), 128)
    per = max(len(env_ids) // threads, 8)
    shards = []
    for k in range(threads):
        ids = list(env_ids[k * per:(k + 1) * per])
        if ids:
            shards.append((ids, k * per))
    made = [(_oracle_shard(name, ids, True)[0], action_pool[:, off:off + len(ids), :]) for ids, off in shards]
    steps_mt = int(min(max(seconds / max(probe * per / len(ids1), 1e-6), 20), 5000))
    with ThreadPoolExecutor(len(made)) as ex:
        t0 = time.perf_counter()
        list(ex.map(lambda ba: _oracle_run(ba[0], ba[1], steps_mt), made))
        dt = time.perf_counter() - t0
    total = sum(b.B for b, _ in made)
    res["cpu_baseline_all_cores"] = {
        "value": total * n * steps_mt / dt, "unit": "agent-steps/s", "cores": len(made), "kind": "port",
        "sample": f"{total} envs x {steps_mt} steps, one shard of {per} envs per thread, {len(made)} threads "
                  f"(os.cpu_count() = {n_cpu}, usable = {host_threads()})",
    }
    return res


# ------------------------------------------------------------------------------------------------------
# worker: one rank = one GPU
# ------------------------------------------------------------------------------------------------------
def worker(args) -> int:
    import torch
    import torch.distributed as dist

    from dl_reference_models_amd import _lib as L
    from dl_reference_models_amd import sharding, workloads as wl
    from dl_reference_models_amd.vec_env import VecReferenceModel, metrics_from_sums

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
    cfg = wl.workload_config(name, env_ids)
    cfg["device"] = str(device)
    if os.environ.get("MAPF_JIT_PREBUILT_TOO"):  # development A/B: step kernels compiled from the source tree's mapf_kernels.inl
        cfg["jit_specialize"] = True
    if os.environ.get("MAPF_SEPARATE_OUTPUTS"):  # A/B knob of VecReferenceModel (output tensors in separate allocations)
        cfg["separate_output_tensors"] = True
    env = VecReferenceModel(cfg)
    L_obs = env.obs_len
    spe = int(cfg["steps_per_episode"])

    # actions: uniform over {0..4}, generated once and resident in HBM (inputs, not part of the path)
    pool = max(args.graph_steps, 1) if args.graph_steps else 128
    action_pool_np = np.random.default_rng(999 + rank).integers(0, 5, size=(pool, b_per, n)).astype(np.int8)
    action_pool = torch.from_numpy(action_pool_np).to(device)
    stream = torch.cuda.current_stream(device)
    sptr = stream.cuda_stream
    step_raw = env.step_raw
    base, stride = action_pool.data_ptr(), b_per * n

    def run_plain(k, ptr=sptr):
        for t in range(k):
            rc = step_raw(base + (t % pool) * stride, ptr, 1)
            if rc != 0:
                raise RuntimeError(f"mapf_step failed: {rc}")

    graphs = {}  # launches per graph -> captured graph

    def graph_of(k):
        if k not in graphs:
            g = torch.cuda.CUDAGraph()
            cap_stream = torch.cuda.Stream(device)
            with torch.cuda.graph(g, stream=cap_stream):
                run_plain(k, torch.cuda.current_stream(device).cuda_stream)
            graphs[k] = g
        return graphs[k]

    G = min(args.graph_steps, max(args.steps, 1)) if args.graph_steps > 0 else 0

    def plan(k):
        """(graph, replays) pairs that make exactly k launches."""
        if G == 0 or k == 0:
            return []
        full, rem = divmod(k, G)
        return ([(graph_of(G), full)] if full else []) + ([(graph_of(rem), 1)] if rem else [])

    def run_steps(k, pl):
        if G == 0:
            run_plain(k)
        else:
            for g, reps in pl:
                for _ in range(reps):
                    g.replay()

    def fence():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize(device)

    def set_phases(staggered):
        env.reset()
        if staggered:  # env with global index i is i mod steps_per_episode steps into its episode
            c = env.get_state()["counters"]
            c[:, L.CTR_STEP_COUNT] = np.asarray(env_ids, dtype=np.int64) % spe
            env.set_state(counters=c)

    def timed_region(staggered):
        set_phases(staggered)
        run_plain(3)  # code object resident before any capture
        torch.cuda.synchronize(device)
        pl_w, pl_t = plan(args.warmup), plan(args.steps)  # capture happens here, outside the timed region
        # pre-roll (state preparation, like the reset above; not warmup and not timed; reported in config): by
        # default one episode length of steps straight before the warmup, so that a SHORT timed window (the driver's
        # --steps 20 is 0.12 ms of GPU time) sees the steady state of a long run -- every env has been through a reset
        # since its phase was set, the placements of the episodes ending in the window have been pre-drawn in the
        # background -- and a GPU that has been busy, instead of the start-up transient after the idle of the capture
        pre = args.pre_roll if args.pre_roll >= 0 else spe
        pl_p = plan(pre)
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
        run_steps(pre, pl_p)
        run_steps(args.warmup, pl_w)
        with torch.cuda.stream(stream):
            env.episode_sums_device(ep_before)
        fence()
        t0 = time.perf_counter()
        ev0.record(stream)
        run_steps(args.steps, pl_t)
        ev1.record(stream)
        fence()
        elapsed = time.perf_counter() - t0
        # HIP events on the launch stream over the timed region: device time per launch (graph replays leave no
        # host gap).  This is the roofline's kernel time.
        kernel_ms = ev0.elapsed_time(ev1) / args.steps
        env.poll_error()
        ep0 = int(ep_before[L.ACC_EPISODES].item())
        resets = int(env.episode_sums()[L.ACC_EPISODES]) - ep0
        per_rank = [elapsed]
        if use_dist:
            tt = torch.zeros(world, dtype=torch.float64, device=None if args.share_gpu else device)
            tt[rank] = elapsed
            dist.all_reduce(tt, op=dist.ReduceOp.SUM)  # (every rank's own time; the job's time is their maximum)
            per_rank = [float(x) for x in tt.tolist()]
            elapsed = max(per_rank)
        return {"elapsed": elapsed, "kernel_ms": kernel_ms, "resets": resets, "per_rank": per_rank}

    legs = {}
    if args.episodes in ("staggered", "both"):
        legs["staggered"] = timed_region(True)
    if args.episodes in ("synchronised", "both"):
        legs["synchronised"] = timed_region(False)
    head = legs["staggered"] if "staggered" in legs else legs["synchronised"]

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
    # all-reduce (96 bytes; latency-bound, so one fused buffer; RCCL on GPUs)
    sums = sharding.all_reduce_stats(env.episode_sums().astype(np.float64),
                                     device=device if (use_dist and not args.share_gpu) else None)
    episode_metrics = metrics_from_sums(sums, n, bool(cfg.get("lifelong_mapf", False)))

    agent_steps = total_envs * n * args.steps
    bytes_per_launch = wl.algorithmic_bytes_per_env_step(n, L_obs, h, w) * b_per
    achieved = bytes_per_launch / (head["kernel_ms"] * 1e-3) / 1e9
    traffic = None
    tfile = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    if os.path.exists(tfile) and b_per == b_weak:
        try:
            traffic = json.load(open(tfile)).get(name, {}).get("hbm_bytes_per_launch")
        except (OSError, ValueError):
            traffic = None

    full, rem = divmod(args.steps, G) if G else (0, 0)
    launch = (f"hipGraph: {full} x {G} launches" + (f" + 1 x {rem}" if rem else "")) if G else "plain launches"
    result = {
        "metric": "agent-steps/sec at 8192 envs x 8 agents on 32x32 grid" if (name == wl.HEADLINE and b_per == b_weak)
        else f"agent-steps/sec ({name}, {b_per} envs per GPU)",
        "value": agent_steps / head["elapsed"],
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
            "workload": name, "envs_per_gpu": b_per, "total_envs": total_envs, "grid": [h, w], "agents": n,
            "obstacle_density": density, "obs_floats": L_obs, "sensor_range": cfg["sensor_range"],
            "steps_per_episode": spe, "lock_metrics": True, "auto_reset": "in-kernel",
            "episode_phases": "staggered" if "staggered" in legs else "synchronised",
            "resets_in_timed_region": head["resets"],
            "pre_roll_steps": args.pre_roll if args.pre_roll >= 0 else spe,
            "actions": "uniform{0..4}, device-resident", "launch": launch,
            "parallelism": f"env-sharded x{world}, no hot-path collective", **env.launch_info(),
        },
        "roofline": {
            "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
            "traffic_source": "profiles/hbm_traffic.json (rocprofv3 PMC passes of this kernel, committed; not measured in this run)" if traffic else None,
            "kernel": "k_step", "kernel_ms": head["kernel_ms"], "isolated_launch_ms": isolated_launch_ms,
            "algorithmic_bytes_per_launch": bytes_per_launch,
        },
        "episodes_finished": float(sums[L.ACC_EPISODES]),
        "episode_metrics": episode_metrics,
    }
    if "synchronised" in legs and "staggered" in legs:
        s = legs["synchronised"]
        result["value_synchronised"] = agent_steps / s["elapsed"]
        result["ms_per_step_synchronised"] = 1e3 * s["elapsed"] / args.steps
        result["resets_in_timed_region_synchronised"] = s["resets"]
        result["roofline"]["kernel_ms_synchronised"] = s["kernel_ms"]
        result["roofline"]["frac_synchronised"] = bytes_per_launch / (s["kernel_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS
    if world > 1:
        result["per_rank_ms_per_step"] = [1e3 * t / args.steps for t in head["per_rank"]]
        result["config"]["rank0_cpus"] = len(rank_cpus) if rank_cpus else None
    if args.share_gpu:
        result["shared_gpu"] = True
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            result.update(cpu_baselines(name, env_ids, action_pool_np, args.cpu_seconds))
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
