#!/bin/bash
# A/B several builds of the library in one GPU session: bash tools/ab.sh lib1.so lib2.so ...  (3 interleaved rounds)
for r in 1 2 3; do for L in "$@"; do
  MAPF_LIB=$L timeout -k 10 200 python bench.py --no-cpu-baseline --steps 3000 --warmup 500 --kernel-samples 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$L round $r kernel_us %.3f value %.4g' % (1e3*d['roofline']['kernel_ms'], d['value']))"
done; done
