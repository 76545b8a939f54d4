#!/bin/bash
run() { timeout -k 10 200 python3 bench.py --no-cpu-baseline --steps 1000 --warmup 200 --kernel-samples 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('%-34s staggered %.3f us  synchronised %.3f us' % ('$1', r['kernel_ms']*1e3, r.get('kernel_ms_synchronised', 0)*1e3))"; }
run base
export HIP_FORCE_DEV_KERNARG=1; run devkernarg; unset HIP_FORCE_DEV_KERNARG
run base2
export HIP_FORCE_DEV_KERNARG=0; run devkernarg0; unset HIP_FORCE_DEV_KERNARG
