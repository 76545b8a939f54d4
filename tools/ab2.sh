#!/bin/bash
# A/B of the headline bench over alternative builds in ONE session: bash tools/ab2.sh [lib ...]
# (the shipped library runs first and last; prints staggered and synchronised step times)
run() {
  timeout -k 10 200 python3 bench.py --no-cpu-baseline --steps 1000 --warmup 200 --kernel-samples 0 ${BENCH_ARGS} 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('%-34s staggered %.3f us  synchronised %.3f us' % ('${MAPF_LIB:-shipped}', r['kernel_ms']*1e3, r.get('kernel_ms_synchronised', 0)*1e3))"
}
unset MAPF_LIB; run || exit 1
for L in "$@"; do export MAPF_LIB=$L; run || exit 1; done
unset MAPF_LIB; run
