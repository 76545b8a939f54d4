#!/usr/bin/env python3
"""Diagnostic: per-phase wave time of k_step from in-kernel s_memtime stamps.

Builds a SEPARATE library with -DMAPF_STAMPS (never the shipped one; MAPF_STAMPS_VARIANT = a build.py variant), runs the workload and prints
the median cycles each phase takes.  Shares only; the stamped build's run time is not representative.
"""
import ctypes as C, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
so = os.environ.get("MAPF_STAMPS_LIB") or os.path.join(ROOT, "build_diag", "libstamps.so")
if not os.environ.get("MAPF_STAMPS_LIB"):  # else: a stamps build made beforehand (hipcc cross-compiles without a GPU):
    # python -m dl_reference_models_amd.build --variant dev_c5 -DMAPF_STAMPS --out build_diag/libstamps_c5.so
    from dl_reference_models_amd import build as hip_build
    hip_build.build_variants([os.environ.get("MAPF_STAMPS_VARIANT", "full")], extra=["-DMAPF_STAMPS"], out=so)
os.environ["MAPF_LIB"] = so
import torch
from dl_reference_models_amd import workloads as wl
from dl_reference_models_amd.vec_env import VecReferenceModel
args = [x for x in sys.argv[1:] if not x.startswith("--")]
stagger = "--stagger" in sys.argv  # spread the episode boundaries: some env of the batch resets in every step
name = args[0] if args else wl.HEADLINE
b = wl.WORKLOADS[name][0]
cfg = wl.workload_config(name, list(range(b)))
env = VecReferenceModel(cfg)
env.reset()
if stagger:
    c = env.get_state()["counters"].copy()
    c[:, 0] = np.arange(b) % int(cfg["steps_per_episode"])
    env.set_state(counters=c)
acts = torch.randint(0, 5, (64, b, cfg["num_agents"]), dtype=torch.int8, device=env.device)
for t in range(130):
    env.step(acts[t % 64])
torch.cuda.synchronize()
blocks = env.launch_info()["blocks"]
ROW = 32  # words per workgroup (kDbgRow)
rows, srows = [], []
for t in range(20):
    env.step(acts[t % 64])
    buf = np.zeros((blocks + 4096) * ROW, dtype=np.uint64)
    n = env._lib.mapf_debug_stamps(env._h, buf.ctypes.data_as(C.c_void_p), buf.size)
    assert n >= blocks * ROW, n
    rows.append(buf[: blocks * ROW].reshape(blocks, ROW).astype(np.int64))
    srows.append(buf[blocks * ROW: n].reshape(-1, ROW).astype(np.int64))
full = np.stack(rows)  # [T, blocks, 16]
samp = np.stack(srows)  # [T, sampler workgroups, 16]
st = full[:, :, :10]
d = np.diff(st, axis=2)
names = ["loads+sync", "move", "goal/lock/term/table", "pair loop+emit", "flush obs", "lock detector", "outputs",
         "state stores", "drain stores"]
tot = st[:, :, 9] - st[:, :, 0]
print(f"workload {name}: {blocks} waves; s_memtime ticks are shader cycles (100 MHz-based clock on gfx950: see guide)")
for k, nme in enumerate(names):
    print(f"  {nme:22s} median {np.median(d[:, :, k]):9.0f}  mean {d[:, :, k].mean():9.0f}  p95 {np.percentile(d[:, :, k], 95):9.0f}")
print(f"  {'wave total':22s} median {np.median(tot):9.0f}  mean {tot.mean():9.0f}  p99 {np.percentile(tot, 99):9.0f}  "
      f"slowest wave of a launch (median over launches) {np.median(tot.max(axis=1)):9.0f}")
slow = tot.argmax(axis=1)
dd = np.stack([d[t, slow[t]] for t in range(d.shape[0])])
print("  phases of the slowest wave (median over launches): " + " ".join(f"{np.median(dd[:, k]):.0f}" for k in range(dd.shape[1])))
# (stamps of different XCDs come from different counters: only differences inside one wave/workgroup are meaningful)
if env.launch_info()["threads"] >= 128:  # observation wave, relative to the state wave's first stamp
    w1 = full[:, :, 10:15] - st[:, :, 0:1]
    for k, nme in enumerate(["obs wave: rows loaded (B0)", "obs wave: released (B1)", "obs wave: observation staged",
                             "obs wave: stores issued", "obs wave: stores drained"]):
        print(f"  {nme:30s} at median {np.median(w1[:, :, k]):9.0f}  p95 {np.percentile(w1[:, :, k], 95):9.0f}")
    cum = st - st[:, :, 0:1]
    print("  state wave stamps at median " + " ".join(f"{np.median(cum[:, :, k]):.0f}" for k in range(10)))
# sub-stamps (slots 16..20), relative to the state wave's first stamp
sub = {16: "target cell known (move phase)", 19: "observation wave released (after B1)", 17: "rewards / flags issued", 18: "agent records issued",
       20: "info / counters staged"}
for k, nme in sub.items():
    v = full[:, :, k] - st[:, :, 0]
    v = v[full[:, :, k] > 0]
    if v.size:
        print(f"  sub-stamp {nme:38s} at median {np.median(v):7.0f}  p95 {np.percentile(v, 95):7.0f}")
# the same sub-stamps for the slowest state wave of each launch (a wave in which an episode ends, typically)
seq = [0, 1, 16, 2, 19, 3, 4, 5, 17, 18, 6, 20, 7, 8, 9]
lab = ["loads", "target", "move", "B1", "goal/lock", "pairs", "flush", "rewards", "records", "lock det", "info staged", "info out", "state st", "drain"]
def deltas(rows):  # rows: [n, ROW]
    pts = [rows[:, k] if k != 0 else rows[:, 0] for k in seq]
    return [np.median(pts[i + 1] - pts[i]) for i in range(len(seq) - 1)]
allw = full.reshape(-1, ROW)
sloww = np.stack([full[t, slow[t]] for t in range(full.shape[0])])
if (allw[:, 19] > 0).all():
    print("  phase (sub-stamps)    " + " ".join(f"{x:>11s}" for x in lab))
    print("  all waves, median     " + " ".join(f"{x:11.0f}" for x in deltas(allw)))
    print("  slowest wave / launch " + " ".join(f"{x:11.0f}" for x in deltas(sloww)))
# lifelong respawn sub-stamps (24..27), for the waves that had an arrival in the step
if (full[:, :, 24] > 0).any():
    m = full[:, :, 27] > full[:, :, 2]
    if m.any():
        seq2 = [2, 24, 25, 26, 27, 3]
        lab2 = ["move end -> stream + ranks", "bounded draw", "rank fixed point", "cell of rank", "rest of goal logic (incl. further arrivals)"]
        rows2 = full[m]
        print(f"  respawn (waves with an arrival: {m.mean() * 100:.0f} % of wave-steps), median cycles (last arrival of the step):")
        for i, nme in enumerate(lab2):
            print(f"    {nme:52s} {np.median(rows2[:, seq2[i + 1]] - rows2[:, seq2[i]]):7.0f}")
# sliced background draw in the tail of the observation wave (slot 23 = slice kind run, 21 / 22 = start / end)
kind = full[:, :, 23]
if (kind > 0).any():
    names = {1: "outputs, 2nd", 2: "outputs, 1st", 3: "bounded draws", 4: "Floyd, 1st", 5: "Floyd, 2nd", 6: "shuffle", 7: "gather"}
    print(f"  observation waves running a draw slice per launch (median): {np.median((kind > 0).sum(axis=1)):.0f} of {blocks}")
    for k in range(1, 8):
        m = kind == k
        if m.any():
            dur = (full[:, :, 22] - full[:, :, 21])[m]
            end = (full[:, :, 22] - st[:, :, 0])[m]
            print(f"    slice {k} ({names[k]:13s}): duration median {np.median(dur):6.0f}  p95 {np.percentile(dur, 95):6.0f};  ends at "
                  f"{np.median(end):6.0f} (p95 {np.percentile(end, 95):6.0f}) after the state wave's first stamp")
pre = st[:, :, 0] - full[:, :, 15]
print(f"  wave entry -> first stamp (scalar loads: kernel arguments + Params): median {np.median(pre):.0f}  p95 {np.percentile(pre, 95):.0f}")

# sampler workgroups (appended to the grid).  s_memtime counters of different workgroups are not comparable (see
# above), so only durations inside a sampler wave are reported, next to the duration of an env workgroup's state wave
# (entry -> last store drained) which a sampler wave has to stay under to be invisible.
if samp.shape[1]:
    act = samp[:, :, 4] == 1
    env_dur = full[:, :, 9] - full[:, :, 15]
    print(f"  sampler workgroups: {samp.shape[1]}, active per launch (median) {np.median(act.sum(axis=1)):.0f}")
    print(f"  env state wave, entry -> done: median {np.median(env_dur):.0f}  p95 {np.percentile(env_dur, 95):.0f}  "
          f"slowest of a launch (median) {np.median(env_dur.max(axis=1)):.0f}")
    d01 = samp[:, :, 1] - samp[:, :, 0]
    print(f"  sampler wave: entry -> need known  median {np.median(d01):.0f}  p95 {np.percentile(d01, 95):.0f}  (idle waves end here)")
    if act.any():
        d12 = (samp[:, :, 2] - samp[:, :, 1])[act]
        d23 = (samp[:, :, 3] - samp[:, :, 2])[act]
        d03 = (samp[:, :, 3] - samp[:, :, 0])[act]
        if (samp[:, :, 8][act] > 0).any():  # first-half sub-stamps: prefetched state fetched (5), outputs (6), bounded draws (7), stream stored (8)
            seq = [1, 5, 6, 7, 8, 2]
            parts = [np.median((samp[:, :, seq[i + 1]] - samp[:, :, seq[i]])[act]) for i in range(len(seq) - 1)]
            print("  first half, median cycles: shfl %.0f | jump-ahead outputs %.0f | bounded draws %.0f | stream store %.0f | stage store %.0f" % tuple(parts))
        print(f"  ACTIVE sampler wave: draw {np.median(d12):.0f} (p95 {np.percentile(d12, 95):.0f})  gather+store {np.median(d23):.0f} "
              f"(p95 {np.percentile(d23, 95):.0f})  entry -> slot stored median {np.median(d03):.0f}  p95 {np.percentile(d03, 95):.0f}  max {d03.max():.0f}")

# inline resets (reset_groups): stamps 24 entry, 25 placement drawn, 26 state image, 27 past B2, 28 reset observation built
rg = full[:, :, 24:29]
have = (rg[:, :, 0] > 0) & (rg[:, :, 4] > rg[:, :, 0])
if have.any():
    dd = np.diff(rg[have], axis=1)
    print(f"  inline resets seen in {int(have.sum())} wave-launches: draw {np.median(dd[:, 0]):.0f}  gather+image {np.median(dd[:, 1]):.0f}  "
          f"B2 wait {np.median(dd[:, 2]):.0f}  reset observation {np.median(dd[:, 3]):.0f}  (median cycles)")
