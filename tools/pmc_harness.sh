#!/bin/bash
# SQ counters of the sampler harness (GPU box, repo root).  Separate passes: the counter file holds at most a few per pass.
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$1; shift
BIN=$1; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR" "SQ_WAVES SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC" "GRBM_GUI_ACTIVE SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_FLAT"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $OUT/p$i -- $ROOT/tools/micro/$BIN "$@" > $OUT/p$i.log 2>&1 || echo "pass $i failed"
done
python3 - $OUT <<'PY'
import csv,glob,sys,collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1]+'/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r['Kernel_Name'][:60]][r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in acc.items():
    print(k)
    for c,vals in sorted(v.items()):
        vals=vals[len(vals)//2:]
        print(f'   {c:26s} {sum(vals)/len(vals):14.0f}')
PY
