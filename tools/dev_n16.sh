#!/bin/bash
# Development round trip of the 16-lane three-wave kernel (the reference's training setup) on a --variant dev_n16 build:
#   python -m dl_reference_models_amd.build --variant dev_n16 && \
#   python -m dl_reference_models_amd.build --variant dev_n16 -DMAPF_STAMPS --out build_diag/libn16_stamps.so
#   bash tools/dev_n16.sh [parity] [ab] [stamps]      (default: all three)
set -e
WHAT="${*:-parity ab stamps}"
WL=ref_training_4096x32x32_n16
export MAPF_LIB=$PWD/build_diag/libn16.so
mkdir -p gpurun_out
if [[ $WHAT == *parity* ]]; then
  timeout -k 10 300 python -m pytest tests/test_engine_parity_gpu.py -x -q -m gpu -k "reference_training_setup" 2>&1 | tail -3
  SOAK_N=16 timeout -k 10 300 python tools/soak_specialized.py 5 12 2>&1 | tail -2
fi
if [[ $WHAT == *ab* ]]; then
  AB_WORKLOAD=$WL timeout -k 10 200 python tools/ab_inproc.py --staggered --rounds 20 build_diag/libn16.so@small_group_rows=off build_diag/libn16.so@small_group_observation=table_walk build_diag/libn16.so 2>&1 | tail -3
  AB_WORKLOAD=$WL timeout -k 10 200 python tools/ab_inproc.py --rounds 20 build_diag/libn16.so@small_group_rows=off build_diag/libn16.so@small_group_observation=table_walk build_diag/libn16.so 2>&1 | tail -3
fi
if [[ $WHAT == *stamps* ]]; then
  export MAPF_STAMPS_LIB=$PWD/build_diag/libn16_stamps.so
  timeout -k 10 200 python tools/stamps3.py $WL --stagger 2>&1 | tail -n +2 > gpurun_out/st_new.txt
  STAMPS_KNOBS=small_group_rows=off timeout -k 10 200 python tools/stamps3.py $WL --stagger 2>&1 | tail -n +2 > gpurun_out/st_old.txt
  head -26 gpurun_out/st_new.txt
fi
