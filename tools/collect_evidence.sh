#!/bin/bash
# Evidence for ONE bench.py configuration of the current build, the same four things for every workload:
#   bench line (with cpu_baseline), rocprofv3 --kernel-trace --stats of the same command, FETCH_SIZE and WRITE_SIZE PMC
#   passes (one counter per pass, no tracing flags with --pmc), all under gpurun_out/ev_<tag>/<key>/ plus one summary.json.
# Usage on the GPU box, from the repo root:
#   bash tools/collect_evidence.sh <tag> <workload> [episodes=staggered] [fused T=0] [extra bench flags ...]
# The program follows `--` directly (python3 bench.py ...): no env / bash -c hop under the profiler.
set -e
TAG=${1:?tag}; W=${2:?workload}; EP=${3:-staggered}; T=${4:-0}; shift; shift; shift || true; shift || true
EXTRA="$@"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
KEY=$W; [ "$T" != "0" ] && KEY="$W+fused$T"
[ -n "$EV_KEY" ] && KEY=$EV_KEY   # (a configuration that differs by extra flags, e.g. c4 on one GPU: --scaling strong --gpus 1)
O=gpurun_out/ev_$TAG/${KEY}_$EP; rm -rf "$O"; mkdir -p "$O"
FUSED=""; STEPS="--steps 2000 --warmup 300"; PSTEPS="--steps 200 --warmup 50"
if [ "$T" != "0" ]; then FUSED="--fused $T"; STEPS="--steps $((T * 20)) --warmup $((T * 3))"; PSTEPS="--steps $((T * 4)) --warmup $T"; fi
python3 bench.py --workload $W $FUSED $STEPS --episodes $EP $EXTRA > "$O/bench.json" 2> "$O/bench.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/kt" -- python3 bench.py --workload $W $FUSED $STEPS --episodes $EP --no-cpu-baseline --api-steps 0 --kernel-samples 0 $EXTRA > "$O/kt_bench.json" 2> "$O/kt.err"
cp $(find "$O/kt" -name "*kernel_stats.csv" | head -1) "$O/kernel_stats.csv"
for ctr in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $ctr --output-format csv -d "$O/pmc_$ctr" -- python3 bench.py --workload $W $FUSED $PSTEPS --episodes $EP --no-cpu-baseline --graph-steps 0 --api-steps 0 --kernel-samples 0 $EXTRA > "$O/pmc_$ctr.json" 2> "$O/pmc_$ctr.err" || echo "pmc pass failed: $ctr"
done
python3 tools/summarize_evidence.py "$KEY" "$EP" "$O"
rm -rf "$O/kt" "$O"/pmc_FETCH_SIZE "$O"/pmc_WRITE_SIZE   # (raw traces: large; the csv / json summaries stay)
