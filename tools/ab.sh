#!/bin/bash
# A/B of the headline bench over alternative builds: bash tools/ab.sh [lib ...]   (the shipped library runs first)
run() { timeout -k 10 200 python bench.py --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('${MAPF_LIB:-shipped}', round(d['value']/1e9, 3), 'G', round(d['ms_per_step']*1e3, 3), 'us')"; }
unset MAPF_LIB; run || exit 1
for L in "$@"; do export MAPF_LIB=$L; run || exit 1; done
