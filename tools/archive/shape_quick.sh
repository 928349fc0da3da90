#!/bin/bash
# quick numbers for one workload/batch: fused random rollout, per-step launches (usage: tools/shape_quick.sh MATE-8v8-9.yaml 8192)
cd "$(dirname "$0")/.."
w=${1:-MATE-8v8-9.yaml}; b=${2:-8192}
for mode in "--rollout 32" "--rollout 0"; do
  python3 bench.py --workload $w --batch $b $mode --steps 1024 --warmup 128 --no-cpu-baseline --no-extras --reps 3 2>/dev/null | tail -1 |
    python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$w $b $mode', 'kernel us', round(d['roofline']['kernel_avg_us'],1), 'frac', round(d['roofline']['frac'],3), 'value %.3g' % d['value'])"
done
