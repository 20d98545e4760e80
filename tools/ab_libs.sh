#!/bin/bash
# Development aid, run on the GPU box: tools/ab_modes.py (one mode) for several library builds, interleaved ROUNDS times on the one box.
#   tools/ab_libs.sh <rounds> <mode> <variant|default> ...   (variants: carparkingmaps_amd/csrc/libcpm_hip_<variant>.so)
cd ${GRAFT_REPO_ROOT:-$(pwd)}
ROUNDS=$1; MODE=$2; shift 2
for r in $(seq $ROUNDS); do
  for v in "$@"; do
    if [ "$v" = default ]; then unset CPM_LIB_PATH; else export CPM_LIB_PATH=$PWD/carparkingmaps_amd/csrc/libcpm_hip_$v.so; fi
    echo "$v $(timeout -k 10 120 python tools/ab_modes.py --modes $MODE --steps 200 --rounds 1 $AB_ARGS 2>/dev/null | grep median)"
  done
done
