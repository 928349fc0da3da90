#!/bin/bash
# A/B between builds of the engine library: tools/ab.sh libA.so libB.so ... (files under mate_amd/lib), alternating, 3 rounds
libs=${@:-libmate_engine.so}
for round in 1 2 3; do
for lib in $libs; do
for b in 4096 65536; do
MATE_ENGINE_LIB=$PWD/mate_amd/lib/$lib timeout 300 python bench.py --batch $b --steps 2000 --warmup 200 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$lib', $b, round(d['value']/1e6,1), 'M/s  kernel', round(r['kernel_avg_us'],2), 'us')"
done; done; done
