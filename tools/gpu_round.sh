#!/bin/bash
# One GPU round trip of the whole repo: the GPU suite, smoke(), then a short bench line of every runner kind.
# Usage on the GPU box: bash tools/gpu_round.sh <tag> [tests|notests]
TAG=${1:-x}; MODE=${2:-tests}
O=gpurun_out/$TAG; mkdir -p "$O"
if [ "$MODE" = "tests" ]; then
  timeout -k 10 900 python -m pytest tests -m gpu -x -q > "$O/tests.log" 2>&1; rc=$?
  tail -5 "$O/tests.log"
  [ $rc -ne 0 ] && exit $rc
  timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" > "$O/smoke.log" 2>&1 || { tail -5 "$O/smoke.log"; exit 1; }
fi
for spec in "c3_8192x32x32_n8 0" "c3_8192x32x32_n8 100" "c2_1024x16x16_n4 0" "c5_1024x64x64_n64_lifelong 0" "ref_training_4096x32x32_n16 0" "cte_8192x16x16_n4 0" "cte_1024x32x32_n8 100"; do
  set -- $spec
  F=""; S="--steps 400 --warmup 100"; [ "$2" != "0" ] && { F="--fused $2"; S="--steps 400 --warmup 100"; }
  timeout -k 10 300 python bench.py --workload $1 $F $S --cpu-seconds 2 > "$O/bench_$1_$2.json" 2> "$O/bench_$1_$2.err" || { echo "bench failed: $spec"; tail -5 "$O/bench_$1_$2.err"; exit 1; }
  python - "$O/bench_$1_$2.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
print("%-34s %7.3f G  %.3f us/step  frac %.3f  frac_kernel %.3f  per_call %.2f us  python %.2f us  cpu1 %.2f M" % (
    d["config"]["workload"], d["value"] / 1e9, d["ms_per_step"] * 1e3, r["frac"], r["frac_kernel"],
    (d["per_call_ms"] or 0) * 1e3, (d["python_api_ms_per_step"] or 0) * 1e3, d.get("cpu_baseline", {}).get("value", 0) / 1e6))
PY
done
timeout -k 10 200 python bench.py --steps 20 --warmup 5 > "$O/bench_driver_style.json" 2> "$O/bench_driver_style.err" && python -c "
import json; d=json.loads(open('$O/bench_driver_style.json').read().strip().splitlines()[-1]); print('driver-style: %.3f G, frac %.3f, frac_kernel %.3f' % (d['value']/1e9, d['roofline']['frac'], d['roofline']['frac_kernel']))"
