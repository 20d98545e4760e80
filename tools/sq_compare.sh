#!/bin/bash
# one PMC pass (instruction counts) of the bench command for the tree in $1 -> $2
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --output-format csv -d $2 -- python3 $1/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-pair $3 > $2.log 2>&1
python3 - $2 <<'PY'
import csv,glob,sys,collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1]+'/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r['Kernel_Name'][:60]][r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in acc.items():
    if 'grouped' in k:
        print(k, {c: round(sum(x)/len(x)) for c,x in v.items()}, 'launches', len(next(iter(v.values()))))
PY
