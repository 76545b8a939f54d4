#!/bin/bash
# Every BASELINE configuration (and the secondary workloads) under the same evidence, one tools/collect_evidence.sh call each
# (VERDICT r3 item 1): bash tools/collect_all_evidence.sh <tag>   -> gpurun_out/ev_<tag>/<key>_<phases>/summary.json
TAG=${1:?tag}
run() {  # (a configuration whose summary exists already is skipped: a failed collection can be resumed)
  local key=$1; [ "${3:-0}" != "0" ] && key="$1+fused$3"; [ -n "$EV_KEY" ] && key=$EV_KEY
  [ -f "gpurun_out/ev_$TAG/${key}_${2:-staggered}/summary.json" ] && return 0
  bash tools/collect_evidence.sh "$TAG" "$@" || { echo "FAILED: $@"; exit 1; }
}
run c3_8192x32x32_n8 staggered
run c3_8192x32x32_n8 synchronised
run c2_1024x16x16_n4 staggered
run c5_1024x64x64_n64_lifelong synchronised
run c5_1024x64x64_n64_lifelong staggered
run ref_training_4096x32x32_n16 staggered
run ref_training_4096x32x32_n16 synchronised
run c3_8192x32x32_n8 staggered 100
run c5_1024x64x64_n64_lifelong synchronised 64
run cte_8192x16x16_n4 staggered
run cte_8192x16x16_n4 synchronised
run cte_8192x16x16_n4 synchronised 100
run cte_1024x32x32_n8 synchronised
run cte_1024x32x32_n8 staggered
run cte_1024x32x32_n8 synchronised 100
EV_KEY=c4_65536_on_one_gpu run c3_8192x32x32_n8 staggered 0 --scaling strong --gpus 1
python3 - "$TAG" <<'PY'
import glob, json, sys
rows = []
for f in sorted(glob.glob(f"gpurun_out/ev_{sys.argv[1]}/*/summary.json")):
    d = json.load(open(f))
    t = d.get("hbm_traffic", {})
    rows.append((d["key"], d["episode_phases"], d["kernel"], d["rocprof_kernel_stats"]["calls"], d["rocprof_us_per_env_step"],
                 d["bench"]["ms_per_step"] * 1e3, d["roofline"]["frac"], d["roofline"]["frac_kernel"], t.get("traffic_over_algorithmic")))
print("| configuration | phases | kernel | rocprof calls | rocprof us / env step | bench us / step | frac (on value) | frac_kernel | traffic / algorithmic |")
print("|---|---|---|---|---|---|---|---|---|")
for r in rows:
    print("| %s | %s | %s | %d | %.3f | %.3f | %.3f | %.3f | %s |" % (r[0], r[1], r[2], r[3], r[4], r[5], r[6], r[7], ("%.2f" % r[8]) if r[8] else "-"))
PY
