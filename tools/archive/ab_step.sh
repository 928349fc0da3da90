#!/bin/bash
# A/B of engine builds on the same box: per-step launches and fused rollouts (usage: [W=MATE-8v8-9.yaml B=8192 R=32] tools/ab_step.sh libA.so libB.so ...)
cd "$(dirname "$0")/.."
for rep in 1 2; do for lib in "$@"; do for mode in "--rollout 0" "--rollout ${R:-128}"; do
  MATE_ENGINE_LIB=$PWD/$lib python3 bench.py --workload ${W:-MATE-4v8-9.yaml} --batch ${B:-4096} $mode --steps 1024 --warmup 128 --no-cpu-baseline --no-extras --reps 3 2>/dev/null | tail -1 |
    python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib $mode', 'kernel us', round(d['roofline']['kernel_avg_us'],2), 'value %.4g' % d['value'])"
done; done; done
