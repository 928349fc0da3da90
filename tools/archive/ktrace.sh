#!/bin/bash
# usage: tools/ktrace.sh <outdir> [bench args]   kernel-trace + stats summary as CSV
out=$1; shift
mkdir -p "$out"
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -o kt -- python3 bench.py --no-cpu-baseline "$@" > "$out/bench.log" 2>&1
head -8 "$out"/kt_kernel_stats.csv | cut -c1-200
grep '"metric"' "$out/bench.log" | cut -c1-330
