#!/bin/bash
# SQ instruction / cycle counters of the timed kernel of one bench.py configuration, four --pmc passes (no tracing flags):
#   bash tools/pmc_quick.sh <tag> [workload] [episodes] [lib] [extra bench flags ...]
TAG=$1; W=${2:-c3_8192x32x32_n8}; EP=${3:-synchronised}; LIB=$4; shift; shift; shift; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
[ -n "$LIB" ] && [ "$LIB" != "-" ] && export MAPF_LIB=$LIB
O=gpurun_out/pmc_$TAG; rm -rf $O; mkdir -p $O
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES" "SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM_WR"; do
  n=$(echo $set | cut -d' ' -f1)
  rocprofv3 --pmc $set --output-format csv -d "$O/$n" -- python3 bench.py --workload $W --steps 100 --warmup 20 --no-cpu-baseline --graph-steps 0 --kernel-samples 0 --api-steps 0 --episodes $EP "$@" > /dev/null 2> "$O/$n.err" || echo "pmc set failed: $set"
done
python3 - "$O" <<'PY'
import glob, json, sys
import pandas as pd
o = sys.argv[1]
res = {}
for f in glob.glob(o + "/**/*counter_collection.csv", recursive=True):
    df = pd.read_csv(f)
    df = df[df["Kernel_Name"].str.contains("k_step|k_cte_step")]
    for c, g in df.groupby("Counter_Name"):
        res[c] = float(g["Counter_Value"].mean())
w = res.get("SQ_WAVES", 1)
out = {k: round(v / w, 1) for k, v in res.items()}
out["SQ_WAVES"] = w
json.dump(out, open(o + "/per_wave.json", "w"), indent=0)
print(json.dumps(out, indent=0))
PY
rm -rf $O/SQ_*/
