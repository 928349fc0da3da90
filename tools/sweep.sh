#!/bin/bash
# env-steps/s of the step path for the BASELINE.json scenarios (random policy; per-GPU shard sizes)
for spec in "MATE-4v2-9.yaml 4096" "MATE-4v8-9.yaml 4096" "MATE-4v8-9.yaml 16384" "MATE-8v8-9.yaml 8192" "MATE-4v8-0.yaml 8192" "MATE-4v8-0.yaml 65536" "MATE-Navigation.yaml 4096" "MATE-Navigation.yaml 32768"; do
  set -- $spec
  python bench.py --workload $1 --batch $2 --steps 500 --warmup 50 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('%-22s batch %6d  %8.1f M env-steps/s  %7.2f us/step  kernel %7.2f us  alg %6.0f GB/s' % ('$1', $2, d['value']/1e6, d['ms_per_step']*1e3, d['roofline']['kernel_avg_us'], d['roofline']['achieved']))"
done
