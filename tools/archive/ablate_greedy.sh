#!/bin/bash
# Weighs the phases of rollout_greedy_kernel (config 3: MATE-8v8-9 x 8192, Greedy vs Greedy): the engine built with one
# phase of the fused loop compiled out (-DMATE_ABLATE=bit: 1 draws, 2 cameras, 4 targets, 8 view, 16 goals/score,
# 32 scratch, 64 pack, 128 the agents' whole step, 256 the zoom solve cut to one iteration) and the kernel time per
# 32-step launch for each.  Build here (tools/ablate_greedy.sh build), run on the GPU box (tools/ablate_greedy.sh run).
# A build with a phase missing is not a simulation: the deltas over-attribute what the missing phase feeds.
cd "$(dirname "$0")/.."
VARIANTS="0 1 2 4 8 16 32 64 128 256"
if [ "$1" = build ]; then
  for m in $VARIANTS; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -mllvm -disable-machine-licm -DMATE_ABLATE=$m \
      -o mate_amd/lib/libmate_engine_abl$m.so mate_amd/csrc/mate_engine.hip &
    if [ $(jobs -r | wc -l) -ge 5 ]; then wait -n; fi
  done; wait; ls mate_amd/lib/libmate_engine_abl*.so
else
  for m in $VARIANTS; do
    MATE_ENGINE_LIB=$PWD/mate_amd/lib/libmate_engine_abl$m.so python3 bench.py --workload MATE-8v8-9.yaml --batch 8192 --policy greedy \
      --steps 1024 --warmup 128 --no-cpu-baseline --no-extras --reps 3 2>/dev/null | tail -1 > /tmp/pq.json
    python3 - $m <<'PY'
import json, sys
try:
    d = json.load(open('/tmp/pq.json'))
    print('ablate', sys.argv[1], 'kernel us per launch', round(d['roofline']['kernel_avg_us'], 1), 'value', '%.3g' % d['value'])
except Exception as exc:
    print('ablate', sys.argv[1], 'failed', exc)
PY
  done
fi
