#!/bin/bash
# SQ counters of the hourly kernels of the bench command (GPU box, repo root), one rocprofv3 --pmc pass per group (PMC is never
# combined with a trace domain).  Prints per-kernel averages per launch; output under gpurun_out/$1.
set -e
TAG=${1:-sq}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 4 --warmup 2 --no-cpu-baseline --no-pair --no-side ${BENCH_ARGS}"
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" \
           "SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_IDX_ACTIVE" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $OUT/p$i -- python3 $ROOT/bench.py $ARGS > $OUT/p$i.log 2>&1 || echo "pass $i failed"
  echo "pass $i done" >> $OUT/progress.log
done
python3 - $OUT <<'PY'
import csv,glob,sys,collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1]+'/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name']
        if "k_grouped_sample" in k or "k_grouped_place" in k or "k_grouped_hour" in k:
            acc[k[:70]][r['Counter_Name']].append(float(r['Counter_Value']))
out=[]
for k,v in sorted(acc.items()):
    out.append(k)
    for c,vals in sorted(v.items()):
        out.append(f'   {c:26s} {sum(vals)/len(vals):14.0f}   (n={len(vals)})')
open(sys.argv[1]+'/sq_summary.txt','w').write('\n'.join(out)+'\n')
print('\n'.join(out))
PY
