#!/bin/bash
# One GPU round trip: parity tests, headline bench, stamps.  Usage: bash tools/gpu_check.sh <tag>
TAG=${1:-x}
mkdir -p gpurun_out
timeout -k 10 700 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider -x > gpurun_out/test_$TAG.log 2>&1; echo "pytest exit=$?"; tail -3 gpurun_out/test_$TAG.log
timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err; echo "bench exit=$?"
python -c "
import json; d=json.load(open('gpurun_out/bench_$TAG.json')); r=d['roofline']; print('value %.4g  ms/step %.5f  kernel_ms %.5f  frac %.4f' % (d['value'], d['ms_per_step'], r['kernel_ms'], r['frac']))"
timeout -k 10 300 python tools/stamps_profile.py 2>&1 | tail -12 | head -11
