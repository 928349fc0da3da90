#!/bin/bash
# usage (GPU box): tools/pmc_quick.sh [bench args]   -> per-wave dynamic instruction counts of the step kernel
export TMPDIR=/tmp
rm -rf /tmp/pq
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_SMEM --output-format csv -d /tmp/pq -o pmc -- python3 bench.py --steps 100 --warmup 20 --no-cpu-baseline "$@" > /tmp/pq.log 2>&1
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('/tmp/pq/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r['Kernel_Name'][:60]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in acc.items():
    if 'step_kernel' in k or 'policy' in k or 'rollout' in k:
        w = sum(d['SQ_WAVES']) / len(d['SQ_WAVES'])
        print(k, 'waves', w, {c: round(sum(v) / len(v) / w, 1) for c, v in d.items() if c != 'SQ_WAVES'})
PY
