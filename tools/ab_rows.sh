set -e
L=dl_reference_models_amd/csrc/libmapfstep.so
for WL in c3_8192x32x32_n8 c2_1024x16x16_n4; do
AB_WORKLOAD=$WL timeout -k 10 300 python tools/ab_inproc.py --staggered --rounds 20 $L@small_group_rows=off $L@small_group_observation=table_walk $L 2>&1 | tail -3
AB_WORKLOAD=$WL timeout -k 10 300 python tools/ab_inproc.py --rounds 20 $L@small_group_rows=off $L@small_group_observation=table_walk $L 2>&1 | tail -3
done
