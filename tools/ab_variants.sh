#!/bin/bash
# A/B of library variants (tools/build_variants.sh) under rocprofv3 on the GPU box: kernel stats of kbench per variant.
#   tools/ab_variants.sh <out tag> <variant|default> ...      (KB_ARGS: extra kbench arguments)
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$1; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  if [ "$v" = default ]; then unset CPM_LIB_PATH; else export CPM_LIB_PATH=$ROOT/carparkingmaps_amd/csrc/libcpm_hip_$v.so; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$v -- python3 $ROOT/tools/kbench.py --steps 10 --configs grouped $KB_ARGS > $OUT/$v.log 2>&1
  echo "== $v"; grep resample $OUT/$v.log | cut -c1-130
  f=$(find $OUT/$v -name '*_kernel_stats.csv' | head -1)
  python3 - "$f" <<'PY'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:3]:
    print(f"   {r['Name'][:60]:60s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:8.1f} us  {float(r['Percentage']):5.1f}%")
PY
done
