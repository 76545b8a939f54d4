run() { timeout -k 10 200 python3 bench.py --no-cpu-baseline --steps 1000 --warmup 200 --kernel-samples 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('%-10s staggered %.3f us  synchronised %.3f us  jit=%s' % ('$1', r['kernel_ms']*1e3, r.get('kernel_ms_synchronised', 0)*1e3, d['config']['jit']))"; }
unset MAPF_JIT_PREBUILT_TOO; run prebuilt; export MAPF_JIT_PREBUILT_TOO=1; run jit; unset MAPF_JIT_PREBUILT_TOO; run prebuilt; export MAPF_JIT_PREBUILT_TOO=1; run jit
