#!/bin/bash
# One GPU round trip: parity tests, headline bench, stamps.  Usage: bash tools/gpu_check.sh <tag> [extra]
TAG=${1:-x}
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -q --timeout 300 -p no:cacheprovider -x > gpurun_out/test_$TAG.log 2>&1
rc=$?; echo "pytest exit=$rc"; tail -5 gpurun_out/test_$TAG.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err; rc=$?; echo "bench exit=$rc"
[ $rc -eq 0 ] || exit $rc
python -c "
import json; d=json.load(open('gpurun_out/bench_$TAG.json')); r=d['roofline']; print('value %.4g  ms/step %.5f  kernel_ms %.5f  frac %.4f' % (d['value'], d['ms_per_step'], r['kernel_ms'], r['frac']))"
if [ "$2" = "extra" ]; then
  timeout -k 10 400 python tools/extra_benchmarks.py > gpurun_out/extra_$TAG.jsonl 2> gpurun_out/extra_$TAG.err; echo "extra exit=$?"
  python - <<PY
import json
for line in open('gpurun_out/extra_$TAG.jsonl'):
    d = json.loads(line)
    print('%-44s single %.2f us  fused %.2f us  fused-last %.2f us' % (d['workload'], d['single_step_per_launch']['us_per_step'], d['fused_obs_every_step']['us_per_step'], d['fused_obs_last_only']['us_per_step']))
PY
fi
timeout -k 10 300 python tools/stamps_profile.py 2>&1 | tail -19
