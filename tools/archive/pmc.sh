#!/bin/bash
# usage: tools/pmc.sh <outdir> <counters...>   (run on the GPU box; one counter group per pass)
out=$1; shift
mkdir -p "$out"
export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$out" -o pmc -- python3 bench.py --steps 60 --warmup 10 --no-cpu-baseline > "$out/bench.log" 2>&1
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for f in glob.glob(out + '/**/*counter_collection.csv', recursive=True):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for row in csv.DictReader(open(f)):
        acc[row['Kernel_Name'][:40]][row['Counter_Name']].append(float(row['Counter_Value']))
    for k, d in acc.items():
        if 'step_kernel' in k or 'reset_kernel' in k:
            print(k, {c: round(sum(v) / len(v), 1) for c, v in d.items()}, 'n=%d' % len(next(iter(d.values()))))
PY
