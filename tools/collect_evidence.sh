#!/bin/bash
# Evidence for one build: rocprofv3 kernel-trace stats of the default bench command, the bench line itself,
# SQ counters and HBM traffic (each --pmc set in its own pass).  Usage on the GPU box: bash tools/collect_evidence.sh <tag>
set -e
TAG=${1:-x}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/ev_$TAG; rm -rf "$O"; mkdir -p "$O"
python3 bench.py > "$O/bench.json" 2> "$O/bench.err"
# kernel-trace stats per leg of the default bench (the headline `value` / roofline is the staggered leg; the default
# command runs both legs in one process, which would mix their launches in one average)
for leg in staggered synchronised; do
  rocprofv3 --kernel-trace --stats --output-format csv -d "$O/kt_$leg" -- python3 bench.py --no-cpu-baseline --episodes $leg > "$O/kt_bench_$leg.json" 2> "$O/kt_$leg.err"
  cp $(find "$O/kt_$leg" -name "*kernel_stats.csv" | head -1) "$O/kernel_stats_$leg.csv"
done
cp "$O/kernel_stats_staggered.csv" "$O/kernel_stats.csv"
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" "SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM"; do
  n=$(echo $set | cut -d' ' -f1)
  rocprofv3 --pmc $set --output-format csv -d "$O/pmc_$n" -- python3 bench.py --steps 200 --warmup 50 --no-cpu-baseline --graph-steps 0 --kernel-samples 0 --episodes staggered > /dev/null 2> "$O/pmc_$n.err" || echo "pmc set failed: $set"
done
python3 - "$O" <<'PY'
import glob, json, sys
import pandas as pd
o = sys.argv[1]
res = {}
for f in glob.glob(o + "/pmc_*/**/*counter_collection.csv", recursive=True):
    df = pd.read_csv(f)
    df = df[df["Kernel_Name"].str.contains("k_step")]
    for c, g in df.groupby("Counter_Name"):
        res[c] = float(g["Counter_Value"].mean())
if "SQ_WAVES" in res:
    res["per_wave"] = {k: v / res["SQ_WAVES"] for k, v in res.items() if k != "SQ_WAVES"}
json.dump(res, open(o + "/pmc_sq.json", "w"), indent=1)
print(json.dumps(res.get("per_wave", res), indent=1))
PY
bash tools/collect_hbm_traffic.sh
cat "$O/bench.json"
grep k_step "$O/kernel_stats.csv" | head -3
