#!/usr/bin/env python3
"""Condenses one tools/collect_evidence.sh directory into summary.json: the bench line's headline fields, the rocprofv3
--stats row of the timed kernel, the per-launch HBM bytes from the two PMC passes.

Corrections (MI355X_MICROARCH.md, HBM section): the counters are in KiB; on gfx950 FETCH_SIZE reports half of the bytes
of a wide coalesced streaming read, so it is doubled; WRITE_SIZE is exact for 16-B-per-lane streaming stores.  The
kernels' traffic is 16-B-per-lane loads / stores except a few scalar-sized items, so the corrected figure is an estimate
for those (stated in DESIGN.md).  Also appends the traffic to gpurun_out/hbm_traffic.json under the workload key (copy it
to profiles/hbm_traffic.json to have bench.py report it as roofline.traffic)."""
import glob, json, os, sys
import pandas as pd

key, phases, out = sys.argv[1], sys.argv[2], sys.argv[3]
bench = json.loads(open(os.path.join(out, "bench.json")).read().strip().splitlines()[-1])
kernel = bench["roofline"]["kernel"]
res = {"key": key, "episode_phases": phases, "kernel": kernel,
       "bench": {k: bench.get(k) for k in ("value", "ms_per_step", "steps", "per_call_ms", "python_api_ms_per_step")},
       "roofline": bench["roofline"], "cpu_baseline": bench.get("cpu_baseline"), "launch": bench["config"].get("launch")}
ks = pd.read_csv(os.path.join(out, "kernel_stats.csv"))
rows = ks[ks["Name"].str.contains(kernel + "<") | ks["Name"].str.contains(kernel + "I")]  # (demangled or mangled template)
if len(rows) == 0:
    rows = ks[ks["Name"].str.contains(kernel)]
row = rows.sort_values("TotalDurationNs", ascending=False).iloc[0]
res["rocprof_kernel_stats"] = {"name": row["Name"][:120], "calls": int(row["Calls"]), "avg_ns": float(row["AverageNs"]),
                               "min_ns": float(row["MinNs"]), "max_ns": float(row["MaxNs"])}
spl = bench["config"].get("env_steps_per_launch", 1)
res["rocprof_us_per_env_step"] = float(row["AverageNs"]) / 1e3 / spl
vals = {}
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    fs = glob.glob(os.path.join(out, "pmc_" + ctr, "**", "*counter_collection.csv"), recursive=True)
    if not fs:
        continue
    df = pd.read_csv(fs[0])
    df = df[df["Kernel_Name"].str.contains(kernel) & (df["Counter_Name"] == ctr)]
    if len(df):
        vals[ctr] = float(df["Counter_Value"].mean())
        vals[ctr + "_dispatches"] = int(len(df))
if "FETCH_SIZE" in vals and "WRITE_SIZE" in vals:
    t = {"FETCH_SIZE_KiB_raw": vals["FETCH_SIZE"], "WRITE_SIZE_KiB_raw": vals["WRITE_SIZE"],
         "dispatches": [vals["FETCH_SIZE_dispatches"], vals["WRITE_SIZE_dispatches"]],
         "fetch_bytes_corrected": 2.0 * vals["FETCH_SIZE"] * 1024.0, "write_bytes": vals["WRITE_SIZE"] * 1024.0}
    t["hbm_bytes_per_launch"] = t["fetch_bytes_corrected"] + t["write_bytes"]
    t["traffic_over_algorithmic"] = t["hbm_bytes_per_launch"] / bench["roofline"]["algorithmic_bytes_per_launch"]
    res["hbm_traffic"] = t
    path = os.path.join("gpurun_out", "hbm_traffic.json")
    allr = json.load(open(path)) if os.path.exists(path) else (json.load(open("profiles/hbm_traffic.json")) if os.path.exists("profiles/hbm_traffic.json") else {})
    allr[key if phases == "staggered" else key + ":" + phases] = t
    json.dump(allr, open(path, "w"), indent=1)
json.dump(res, open(os.path.join(out, "summary.json"), "w"), indent=1)
print(json.dumps({k: res[k] for k in ("key", "episode_phases", "rocprof_kernel_stats", "rocprof_us_per_env_step")}
                 | {"frac": bench["roofline"]["frac"], "frac_kernel": bench["roofline"]["frac_kernel"],
                    "traffic_over_algorithmic": res.get("hbm_traffic", {}).get("traffic_over_algorithmic")}))
