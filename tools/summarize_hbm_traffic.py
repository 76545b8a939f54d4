#!/usr/bin/env python3
"""Turn the two rocprofv3 --pmc passes of tools/collect_hbm_traffic.sh into bytes per k_step launch.

Corrections (MI355X_MICROARCH.md, HBM section): the counters are in KiB; on gfx950 FETCH_SIZE reports half
of the bytes of a wide coalesced streaming read, so it is doubled; WRITE_SIZE is exact for 16-B-per-lane
streaming stores.  The step kernel's traffic is 16-B-per-lane loads/stores except a few scalar-sized items,
so the corrected figure is an estimate for those (stated in DESIGN.md)."""
import glob, json, os, sys
import pandas as pd

name, out = sys.argv[1], sys.argv[2]
vals = {}
for sub, ctr in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    f = glob.glob(os.path.join(out, sub, "**", "*counter_collection.csv"), recursive=True)[0]
    df = pd.read_csv(f)
    df = df[df["Kernel_Name"].str.contains("k_step") & (df["Counter_Name"] == ctr)]
    vals[ctr] = float(df["Counter_Value"].mean())
    vals[ctr + "_dispatches"] = int(len(df))
res = {
    "FETCH_SIZE_KiB_raw": vals["FETCH_SIZE"], "WRITE_SIZE_KiB_raw": vals["WRITE_SIZE"],
    "dispatches": [vals["FETCH_SIZE_dispatches"], vals["WRITE_SIZE_dispatches"]],
    "fetch_bytes_corrected": 2.0 * vals["FETCH_SIZE"] * 1024.0, "write_bytes": vals["WRITE_SIZE"] * 1024.0,
}
res["hbm_bytes_per_launch"] = res["fetch_bytes_corrected"] + res["write_bytes"]
path = os.path.join("profiles", "hbm_traffic.json")
allr = json.load(open(path)) if os.path.exists(path) else {}
allr[name] = res
os.makedirs("gpurun_out", exist_ok=True)
json.dump(allr, open(os.path.join("gpurun_out", "hbm_traffic.json"), "w"), indent=1)
print(json.dumps({name: res}, indent=1))
