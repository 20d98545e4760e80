#!/bin/bash
# Development aid, run on the GPU box: bench.py's headline (no side records) for several library variants, interleaved ROUNDS times
# on the one box (boxes differ by 1-3 %, runs on a box by ~1 %): prints every run and the median per variant.
#   tools/ab_bench.sh <rounds> <variant|default> ...     (variants: tools/build_variants.sh; BENCH_ARGS: extra bench.py arguments)
cd ${GRAFT_REPO_ROOT:-$(pwd)}
ROUNDS=$1; shift
for r in $(seq $ROUNDS); do
  for v in "$@"; do
    if [ "$v" = default ]; then unset CPM_LIB_PATH; else export CPM_LIB_PATH=$PWD/carparkingmaps_amd/csrc/libcpm_hip_$v.so; fi
    timeout -k 10 120 python bench.py --steps 200 --no-side --no-cpu-baseline --no-pair $BENCH_ARGS 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
d=d.get('not_a_measurement', d)   # (a variant library's line is marked invalid: its timing is still what the A/B wants)
print('$v', round(d['ms_per_step'],4), round(d['roofline']['avg_launch_ms']*1e3,2))"
  done
done | tee /tmp/ab_bench.out
python3 - <<'PY'
import statistics
from collections import defaultdict
a = defaultdict(list)
for l in open('/tmp/ab_bench.out'):
    v, ms, us = l.split()
    a[v].append((float(ms), float(us)))
for v, x in a.items():
    print(f"{v:10s} median ms/step {statistics.median(m for m, _ in x):.4f}  min {min(m for m, _ in x):.4f}  dominant launch median {statistics.median(u for _, u in x):.2f} us")
PY
