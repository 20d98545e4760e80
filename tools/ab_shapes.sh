#!/bin/bash
# Development aid, run on the GPU box: GPU tests of the product library, then the headline and a set of other shapes for the product
# library against a variant (tools/build_variants.sh), interleaved on the one box.
#   tools/ab_shapes.sh <variant>
cd ${GRAFT_REPO_ROOT:-$(pwd)}
V=${1:-pow2}
mkdir -p gpurun_out/r4z
python -m pytest tests -m gpu -x -q > gpurun_out/r4z/gputest_shapes.log 2>&1 || { tail -30 gpurun_out/r4z/gputest_shapes.log; exit 1; }
tail -1 gpurun_out/r4z/gputest_shapes.log
tools/ab_bench.sh 4 default $V 2>&1 | tail -2
for shape in "--zones 2357 --cpz 1000 --melbourne" "--zones 2357 --cpz 100 --melbourne" "--zones 2357 --cpz 500 --melbourne" "--zones 2357 --cpz 1000" "--zones 3000 --cpz 1000" "--zones 8192 --cpz 500" "--zones 4096 --cpz 1000 --skew 32"; do
  echo "== $shape"
  AB_ARGS="$shape --steps 100" tools/ab_libs.sh 2 5 default $V
done
