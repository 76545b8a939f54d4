#!/usr/bin/env python3
"""Diagnostic: timeline of the three-wave step kernel (k_step3) from in-kernel s_memtime stamps, in cycles after the
state wave's entry (median / p95 over workgroups and launches, and for the slowest workgroup of each launch).

Needs a -DMAPF_STAMPS library (never the shipped one): MAPF_STAMPS_LIB=<path> [STAMPS_KNOBS=key=value,...] python tools/stamps3.py [workload] [--stagger]
"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["MAPF_LIB"] = os.environ["MAPF_STAMPS_LIB"]
import torch
from dl_reference_models_amd import workloads as wl
from dl_reference_models_amd.vec_env import VecReferenceModel
args = [x for x in sys.argv[1:] if not x.startswith("--")]
stagger = "--stagger" in sys.argv
name = args[0] if args else wl.HEADLINE
b = wl.WORKLOADS[name][0]
cfg = wl.workload_config(name, list(range(b)))
for kv in filter(None, os.environ.get("STAMPS_KNOBS", "").split(",")):  # engine knobs of the env config: key=value,key=value
    cfg[kv.partition("=")[0]] = kv.partition("=")[2]
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
info = env.launch_info()
blocks, ROW = info["blocks"], 32
rows, srows = [], []
for t in range(20):
    env.step(acts[t % 64])
    buf = np.zeros((blocks + 4096) * ROW, dtype=np.uint64)
    n = env._lib.mapf_debug_stamps(env._h, buf.ctypes.data_as(C.c_void_p), buf.size)
    rows.append(buf[: blocks * ROW].reshape(blocks, ROW).astype(np.int64))
    srows.append(buf[blocks * ROW: n].reshape(-1, ROW).astype(np.int64))
full = np.stack(rows)
samp = np.stack(srows)
rel = full - full[:, :, 15:16]
names = {4: "W0 first 16 B of records + actions", 5: "W0 whole records", 0: "W0 state in registers (stamp 0)", 16: "W0 target cell known", 2: "W0 moves resolved", 19: "W0 past B1", 3: "W0 goal logic + blocking done",
         17: "W0 rewards / flags issued", 18: "W0 records issued", 8: "W0 body done", 9: "W0 stores drained",
         10: "W1 rows in LDS", 11: "W1 past B1", 12: "W1 observation staged", 13: "W1 stream issued", 14: "W1 stream drained",
         6: "W1 wave entry", 7: "W2 wave entry",
         25: "W1 window rows in registers / k_step3: first candidate prepared", 26: "W1 window masks built / k_step3: arrives at B1", 27: "W1 turn-dependent cells resolved",
         21: "W2 state in registers", 20: "W2 arrives at B1", 22: "W2 past B1", 24: "W2 rewards / flags / hot plane issued", 29: "W2 lock detector done", 30: "W2 info / counters stored", 31: "W2 end (slice incl.)"}
order = [0, 16, 2, 19, 8, 9, 6, 10, 25, 26, 11, 27, 12, 13, 14, 7, 21, 20, 22, 24, 29, 30, 31]
print(f"workload {name} ({'staggered' if stagger else 'synchronised'}): {blocks} workgroups x {info['threads']} threads; cycles after the state wave's entry")
end = np.max(np.stack([rel[:, :, 9], rel[:, :, 14], rel[:, :, 31]]), axis=0)
slow = end.argmax(axis=1)
for k in order:
    v = rel[:, :, k][full[:, :, k] > 0]
    if v.size == 0:
        continue
    sl = np.array([rel[t, slow[t], k] for t in range(rel.shape[0])])
    print(f"  {names[k]:34s} median {np.median(v):7.0f}  p95 {np.percentile(v, 95):7.0f}   slowest workgroup {np.median(sl):7.0f}")
print(f"  workgroup end (last of the three)  median {np.median(end):7.0f}  p95 {np.percentile(end, 95):7.0f}   slowest workgroup {np.median(end.max(axis=1)):7.0f}")
three = np.stack([rel[:, :, 9], rel[:, :, 14], rel[:, :, 31]])
who = three.argmax(axis=0)
print(f"  workgroup end: p99 {np.percentile(end, 99):.0f}  p99.9 {np.percentile(end, 99.9):.0f}; last wave of a workgroup: state {np.mean(who == 0):.2f} / obs {np.mean(who == 1):.2f} / aux {np.mean(who == 2):.2f}; "
      f"of the slowest 1 %: state {np.mean(who[end >= np.percentile(end, 99)] == 0):.2f} / obs {np.mean(who[end >= np.percentile(end, 99)] == 1):.2f} / aux {np.mean(who[end >= np.percentile(end, 99)] == 2):.2f}")
kind = np.where(full[:, :, 4] > full[:, :, 15], full[:, :, 23], 0)  # (the buffer keeps a launch's stamps: a slice of THIS launch started after this launch's entry)
if (kind > 0).any():
    print(f"  state waves running a draw slice (after B1) per launch (median): {np.median((kind > 0).sum(axis=1)):.0f}")
    for k in range(1, 8):
        m = kind == k
        if m.any():
            dur = (full[:, :, 8] - full[:, :, 4])[m]
            end = (full[:, :, 8] - full[:, :, 15])[m]
            print(f"    slice {k}: duration median {np.median(dur):6.0f}  p95 {np.percentile(dur, 95):6.0f}; ends at {np.median(end):6.0f} (p95 {np.percentile(end, 95):6.0f}) after entry")

if samp.shape[1]:  # sampler workgroups (their own clock origin: only durations inside a wave mean something)
    act = samp[:, :, 4] > 0
    d01 = samp[:, :, 1] - samp[:, :, 0]
    print(f"  sampler workgroups: {samp.shape[1]}, active per launch (median) {np.median(act.sum(axis=1)):.0f};  entry -> need known: median {np.median(d01):.0f}  p95 {np.percentile(d01, 95):.0f}")
    if act.any():
        d03 = (samp[:, :, 3] - samp[:, :, 0])[act]
        d12 = (samp[:, :, 2] - samp[:, :, 1])[act]
        print(f"  active sampler wave (wave 0 of its workgroup): draw {np.median(d12):.0f} (p95 {np.percentile(d12, 95):.0f}), entry -> stored median {np.median(d03):.0f}  p95 {np.percentile(d03, 95):.0f}  max {d03.max():.0f}")

if info["lanes_per_env"] == 64 and full[:, :, 3].any():  # k_stepw stamps build: HW_REG_HW_ID of the three waves (gfx9: simd_id bits 5:4, cu_id 11:8, sh 12, se 15:13; xcc in the high bits)
    hw = full[-1, :, 3:6]
    simd = (hw >> 4) & 3
    cu = ((hw >> 8) & 0xFF) | ((hw >> 32) << 8)  # cu_id, sh_id, se_id of HW_ID and the XCC id: one value per CU
    print("  wave placement (last launch): SIMD ids of (W0, W1, W2), first 12 workgroups:", [tuple(int(x) for x in simd[i]) for i in range(12)])
    same_cu = (cu[:, 0] == cu[:, 1]) & (cu[:, 1] == cu[:, 2])
    print(f"  workgroups whose three waves report the same CU: {same_cu.mean():.2f}")
    import collections
    per = collections.Counter()
    kinds = collections.defaultdict(list)
    for w in range(hw.shape[0]):
        for k in range(3):
            key = (int(cu[w, k]), int(simd[w, k]))
            per[key] += 1
            kinds[key].append(k)
    cnt = collections.Counter(per.values())
    print("  waves per (CU, SIMD): histogram", dict(sorted(cnt.items())), " (CU, SIMD) pairs used:", len(per))
    mix = collections.Counter(tuple(sorted(v)) for v in kinds.values())
    print("  kinds of waves sharing a SIMD (0 state, 1 obs, 2 aux), most common:", mix.most_common(8))
