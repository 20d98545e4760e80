#!/bin/bash
# Run on the GPU box (via gpurun) from the repo root: rocprofv3 kernel trace + stats of the bench command, then the HBM traffic
# counters in their own passes (FETCH_SIZE and WRITE_SIZE do not fit one pass; PMC is never combined with other trace domains),
# then kernel stats of the model-selection sweep (Z = 2,357, travel times on: the travel kernel and the table builders).
# Output under gpurun_out/$1; tools/summarize_profiles.py turns it into profiles/<name>_*.
set -e
TAG=${1:-prof}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 5 --warmup 2 --no-cpu-baseline --no-pair --skip-side melbourne,per_rank_emulated ${BENCH_ARGS}"  # (the side records stay in: table_build, per_dataset and skewed put their kernels into the same trace;
# melbourne and per_rank_emulated are left out of the counter passes: rocprofv3 --pmc segfaulted inside the first launch of the melbourne record's x 100 sampler, and the emulated ranks are 13 GB of tables per pass)
# (the trace pass with more steps than the counter passes: the first launches of a run are cold and a kernel's AVERAGE is set against bench.py's own)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --steps 40 --warmup 3 --no-cpu-baseline --no-pair ${BENCH_ARGS} > $OUT/bench_trace.log 2>&1
echo trace done >> $OUT/progress.log
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $ROOT/bench.py $ARGS > $OUT/bench_fetch.log 2>&1
echo fetch done >> $OUT/progress.log
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $ROOT/bench.py $ARGS > $OUT/bench_write.log 2>&1
echo write done >> $OUT/progress.log
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_l2 -- python3 $ROOT/bench.py $ARGS > $OUT/bench_l2.log 2>&1 || true
echo l2 done >> $OUT/progress.log
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/sweep -- python3 $ROOT/tools/sweep_bench.py --lanes 1 > $OUT/sweep.log 2>&1 || true
echo sweep done >> $OUT/progress.log
tail -1 $OUT/bench_trace.log | cut -c1-300
grep "grid points" $OUT/sweep.log || true
