#!/bin/bash
# A/B on several workloads: tools/ab3.sh libA.so libB.so   (files under mate_amd/lib)
libs="$@"
for round in 1 2; do
for lib in $libs; do
while read -r wl batch policy; do
MATE_ENGINE_LIB=$PWD/mate_amd/lib/$lib timeout 300 python bench.py --workload $wl --batch $batch --policy $policy --steps 1024 --warmup 128 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; p=d.get('per_step_launch') or {}
print('$lib', '$wl', $batch, '$policy', round(d['value']/1e6,1), 'M/s  kernel', round(r['kernel_avg_us'],2), 'us  per-step', round(p.get('value',0)/1e6,1), 'M/s', round(p.get('kernel_avg_us',0),2), 'us')"
done <<'SPECS'
MATE-4v8-9.yaml 4096 random
MATE-4v8-9.yaml 8192 random
MATE-8v8-9.yaml 8192 random
MATE-8v8-9.yaml 8192 greedy
SPECS
done; done
