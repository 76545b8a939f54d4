#!/bin/bash
# Diagnostic: the single-agent env at 32x32 cells x 8 agents over batch sizes, single-step and fused (T = 100), in phase.
#   bash tools/cte_lanes_sweep.sh            the engine's own choice of lanes per env   -> gpurun_out/cte_batch_sweep.jsonl
#   (WL=<workload> BATCHES="..." APPEND=1 select another shape / batch list and append to the file)
#   bash tools/cte_lanes_sweep.sh 8 16 32 64 every listed group width                   -> gpurun_out/cte_lanes_sweep.jsonl
# (the heuristics in mapf_create -- mapf_step.hip -- were picked from the second form: profiles/r04/cte_lanes_sweep.jsonl)
set -e
LANES="${*:-0}"
OUT=gpurun_out/cte_batch_sweep.jsonl; [ "$LANES" != "0" ] && OUT=gpurun_out/cte_lanes_sweep.jsonl
WL=${WL:-cte_1024x32x32_n8}
mkdir -p gpurun_out; [ -z "$APPEND" ] && : > $OUT
for B in ${BATCHES:-1024 2048 4096 8192 16384}; do for LPE in $LANES; do for F in 0 100; do
  python bench.py --workload $WL --scaling strong --gpus 1 --total-envs $B --episodes synchronised --lanes-per-env $LPE --fused $F --steps 2000 --warmup 200 --no-cpu-baseline --api-steps 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print(json.dumps({'workload': '$WL', 'envs': $B, 'lanes_requested': $LPE, 'fused': $F, 'us_per_step': round(d['ms_per_step']*1e3,3), 'env_steps_per_s': d['env_steps_per_s'], 'frac': round(r['frac'],3), 'frac_kernel': round(r['frac_kernel'],3), 'lanes_per_env': d['config']['lanes_per_env'], 'blocks': d['config']['blocks']}))" >> $OUT
done; done; done
cat $OUT
