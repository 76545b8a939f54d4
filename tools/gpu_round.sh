#!/bin/bash
# One GPU round trip of this round's checks.  Usage: bash tools/gpu_round.sh <tag> [tests|notests] [extra]
TAG=${1:-x}
O=gpurun_out/$TAG; mkdir -p $O
python3 -c "import os; print('cpu_count', os.cpu_count(), 'affinity', len(os.sched_getaffinity(0))); print(open('/sys/fs/cgroup/cpu.max').read() if os.path.exists('/sys/fs/cgroup/cpu.max') else 'no cpu.max')" > $O/box.txt 2>&1
rocm-smi --showid 2>/dev/null | grep -c "GPU\[" >> $O/box.txt
if [ "$2" != "notests" ]; then
  timeout -k 10 900 python3 -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider -x > $O/test.log 2>&1
  rc=$?; echo "pytest exit=$rc"; tail -5 $O/test.log
  [ $rc -eq 0 ] || exit $rc
fi
timeout -k 10 300 python3 bench.py > $O/bench.json 2> $O/bench.err; rc=$?; echo "bench exit=$rc"; tail -3 $O/bench.err
[ $rc -eq 0 ] || exit $rc
timeout -k 10 120 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_driver.json 2> $O/bench_driver.err; echo "driver-style bench exit=$?"
python3 - <<PY
import json
for f in ("bench", "bench_driver"):
    d = json.load(open("$O/%s.json" % f)); r = d["roofline"]
    print("%-12s value %.4g (sync %.4g)  ms/step %.5f (sync %.5f)  kernel_ms %.5f  frac %.3f (sync %.3f) resets %s launch %s" % (
        f, d["value"], d.get("value_synchronised", 0), d["ms_per_step"], d.get("ms_per_step_synchronised", 0), r["kernel_ms"], r["frac"],
        r.get("frac_synchronised", 0), d["config"]["resets_in_timed_region"], d["config"]["launch"]))
    for k in ("cpu_baseline", "cpu_baseline_all_cores"):
        if k in d: print("   ", k, "%.4g" % d[k]["value"], d[k]["cores"], d[k]["sample"])
PY
if [ "$3" = "extra" ]; then
  timeout -k 10 200 python3 bench.py --gpus 2 --share-gpu --steps 200 --warmup 50 --no-cpu-baseline > $O/bench_share2.json 2> $O/bench_share2.err; echo "2-rank shared-gpu rehearsal exit=$?"
  timeout -k 10 200 python3 bench.py --scaling strong --steps 300 --warmup 50 --no-cpu-baseline > $O/bench_strong1.json 2> $O/bench_strong1.err; echo "strong N=1 (65536 envs) exit=$?"
  timeout -k 10 400 python3 tools/extra_benchmarks.py > $O/extra.jsonl 2> $O/extra.err; echo "extra exit=$?"
  python3 - <<PY
import json
for f in ("bench_share2", "bench_strong1"):
    try:
        d = json.load(open("$O/%s.json" % f)); print(f, "value %.4g ms/step %.5f n_gpus %d scaling %s envs/gpu %d" % (d["value"], d["ms_per_step"], d["n_gpus"], d["scaling"], d["config"]["envs_per_gpu"]))
    except Exception as e: print(f, "unreadable", e)
for line in open('$O/extra.jsonl'):
    d = json.loads(line)
    print('%-60s single %.2f us  fused %.2f us  fused-last %.2f us  fused-sampled %s' % (d['workload'], d['single_step_per_launch']['us_per_step'], d['fused_obs_every_step']['us_per_step'], d['fused_obs_last_only']['us_per_step'], ('%.2f us' % d['fused_sampled_policy']['us_per_step']) if 'fused_sampled_policy' in d else '-'))
PY
fi
