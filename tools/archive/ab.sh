#!/bin/bash
# A/B between builds of the engine library: tools/ab.sh libA.so libB.so ... (files under mate_amd/lib), alternating, 3 rounds.
# Prints the default bench (fused rollouts) and the per-step-launch rate of the same run.
libs=${@:-libmate_engine.so}
for round in 1 2 3; do
for lib in $libs; do
for b in ${AB_BATCHES:-4096}; do
MATE_ENGINE_LIB=$PWD/mate_amd/lib/$lib timeout 300 python bench.py --batch $b --steps 2048 --warmup 128 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; p=d.get('per_step_launch') or {}
print('$lib', $b, 'rollout', round(d['value']/1e6,1), 'M/s  frac', round(r['frac'],3), ' per-step', round(p.get('value',0)/1e6,1), 'M/s kernel', round(p.get('kernel_avg_us',0),2), 'us')"
done; done; done
