#!/bin/bash
# Development round trip for the c3 step kernel on a -DMAPF_DEV_C3 library (build_diag/libdev.so): soak of the N = 8,
# L = 33 specialisation against the oracle, then the headline bench A/B against the shipped library.
# Usage: bash tools/dev_c3.sh [soak cases] [extra libs ...]
CASES=${1:-40}; shift
export SOAK_N=8 SOAK_MASK=1
MAPF_LIB=build_diag/libdev.so timeout -k 10 400 python3 tools/soak_specialized.py 2026 $CASES > gpurun_out/dev_soak.log 2>&1; rc=$?
tail -4 gpurun_out/dev_soak.log; echo "soak exit=$rc"
[ $rc -eq 0 ] || exit $rc
unset SOAK_N SOAK_MASK
bash tools/ab2.sh build_diag/libdev.so "$@"
