#!/bin/bash
# On the GPU box: the 64-lane development build (-DMAPF_DEV_C5, built beforehand into build_diag/) through tools/dev_c5.py,
# then the stamps build's reset phases.  bash tools/dev_c5.sh
set -e
export MAPF_LIB=build_diag/libc5.so
timeout -k 10 400 python3 tools/dev_c5.py all
if [ -f build_diag/libc5_stamps.so ]; then
  MAPF_STAMPS_LIB=build_diag/libc5_stamps.so timeout -k 10 200 python3 tools/stamps_profile.py c5_1024x64x64_n64_lifelong --stagger | tail -12
fi
