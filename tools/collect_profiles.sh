#!/bin/bash
# Run on the GPU box (via gpurun) from the repo root: rocprofv3 kernel trace + stats of the bench
# command, then the HBM traffic counters in their own passes (FETCH_SIZE and WRITE_SIZE do not fit
# one pass; PMC is never combined with other trace domains).  Output under gpurun_out/$1.
set -e
TAG=${1:-prof}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 5 --warmup 2 --no-cpu-baseline ${BENCH_ARGS}"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py $ARGS > $OUT/bench_trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $ROOT/bench.py $ARGS > $OUT/bench_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $ROOT/bench.py $ARGS > $OUT/bench_write.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_l2 -- python3 $ROOT/bench.py $ARGS > $OUT/bench_l2.log 2>&1 || true
tail -1 $OUT/bench_trace.log | cut -c1-300
ls $OUT/*
