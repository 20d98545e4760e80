#!/bin/bash
# Development aid: builds tuning / diagnostic variants of the library next to the product one (never shipped, git-ignored):
#   tools/build_variants.sh name1 "-DFLAG=1 ..." name2 "..."   ->  carparkingmaps_amd/csrc/libcpm_hip_<name>.so
# tools/kbench.py / bench.py pick one with CPM_LIB_PATH.
set -e
cd "$(dirname "$0")/../carparkingmaps_amd/csrc"
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function -mllvm -amdgpu-atomic-optimizer-strategy=None $flags -shared -o libcpm_hip_$name.so cpm_api.hip &
done
wait
ls -la libcpm_hip_*.so
