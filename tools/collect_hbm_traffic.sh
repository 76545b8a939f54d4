#!/bin/bash
# HBM traffic of the step kernel from the L2 memory-side counters, one counter per pass
# (MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE do not fit one pass; no tracing flags with --pmc).
# Usage (on the GPU box, from the repo root):  bash tools/collect_hbm_traffic.sh [workload]
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
W=${1:-c3_8192x32x32_n8}
OUT=gpurun_out/traffic_$W
rm -rf "$OUT"; mkdir -p "$OUT/fetch" "$OUT/write"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -- python3 bench.py --workload "$W" --steps 200 --warmup 50 --no-cpu-baseline --graph-steps 0 --kernel-samples 0 --episodes staggered > "$OUT/fetch/bench.json" 2> "$OUT/fetch/err.log"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -- python3 bench.py --workload "$W" --steps 200 --warmup 50 --no-cpu-baseline --graph-steps 0 --kernel-samples 0 --episodes staggered > "$OUT/write/bench.json" 2> "$OUT/write/err.log"
python3 tools/summarize_hbm_traffic.py "$W" "$OUT"
