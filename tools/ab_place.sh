#!/bin/bash
# A/B of kbench configurations under rocprofv3 (GPU box, repo root): one kernel-stats table per config.
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$1; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for cfg in "$@"; do
  tag=$(echo $cfg | tr ':' '_')
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$tag -- python3 $ROOT/tools/kbench.py --steps 6 --configs $cfg $KB_ARGS > $OUT/$tag.log 2>&1
  echo "== $cfg"; grep resample $OUT/$tag.log | cut -c1-150
  f=$(find $OUT/$tag -name '*_kernel_stats.csv' | head -1)
  python3 - "$f" <<'PY'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:6]:
    print(f"   {r['Name'][:60]:60s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:8.1f} us  {float(r['Percentage']):5.1f}%")
PY
done
