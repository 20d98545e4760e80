#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out/r4z
python -m pytest tests -m gpu -x -q > gpurun_out/r4z/gputest_cpt.log 2>&1 || { tail -30 gpurun_out/r4z/gputest_cpt.log; exit 1; }
tail -1 gpurun_out/r4z/gputest_cpt.log
tools/ab_bench.sh 3 default oldcpt 2>&1 | tail -2
for shape in "--zones 4096 --cpz 500" "--zones 4096 --cpz 200" "--zones 4096 --cpz 300" "--zones 4096 --cpz 2000" "--zones 2357 --cpz 1000" "--zones 2357 --cpz 1000 --melbourne" "--zones 2357 --cpz 500 --melbourne" "--zones 8192 --cpz 500"; do
  echo "== $shape"
  AB_ARGS="$shape" tools/ab_libs.sh 2 5 default oldcpt
done
