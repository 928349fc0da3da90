#!/bin/bash
# config-3 bench line (kernel time per 32-step launch, env-steps/s) for each engine build given as argument
cd "$(dirname "$0")/.."
for lib in "$@"; do
  MATE_ENGINE_LIB=$PWD/$lib python3 bench.py --workload MATE-8v8-9.yaml --batch 8192 --policy greedy --steps 1024 --warmup 128 --no-cpu-baseline --no-extras --reps 3 2>/dev/null | tail -1 |
    python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib', 'kernel us', round(d['roofline']['kernel_avg_us'],1), 'frac', round(d['roofline']['frac'],3), 'value %.3g' % d['value'])"
done
