#!/bin/bash
# Weighs the phases of rollout_kernel: builds the engine with one phase of the fused loop compiled out (-DMATE_ABLATE=bit:
# 1 draws, 2 cameras, 4 targets, 8 view, 16 goals/score, 32 scratch, 64 pack) and reports dynamic instruction counts per
# environment-step and the kernel time for each.  Build here (tools/ablate_rollout.sh build), run on the GPU box
# (tools/ablate_rollout.sh run).  Results of a build with a phase missing are meaningless as a simulation.
cd "$(dirname "$0")/.."
if [ "$1" = build ]; then
  for m in 0 1 2 4 8 16 32 64; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -mllvm -disable-machine-licm -DMATE_ABLATE=$m \
      -o mate_amd/lib/libmate_engine_abl$m.so mate_amd/csrc/mate_engine.hip &
  done; wait; ls mate_amd/lib/libmate_engine_abl*.so
else
  export TMPDIR=/tmp
  for m in 0 1 2 4 8 16 32 64; do
    rm -rf /tmp/pq
    MATE_ENGINE_LIB=$PWD/mate_amd/lib/libmate_engine_abl$m.so rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d /tmp/pq -o pmc -- python3 bench.py --steps 512 --warmup 128 --no-cpu-baseline --no-extras --reps 1 > /tmp/pq.log 2>&1
    MATE_ENGINE_LIB=$PWD/mate_amd/lib/libmate_engine_abl$m.so python3 bench.py --steps 1024 --warmup 128 --no-cpu-baseline --no-extras --reps 3 2>/dev/null | tail -1 > /tmp/pq.json
    python3 - $m <<'PY'
import csv, glob, collections, json, sys
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('/tmp/pq/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r['Kernel_Name'][:60]][r['Counter_Name']].append(float(r['Counter_Value']))
try:
    us = json.load(open('/tmp/pq.json'))['roofline']['kernel_avg_us']
except Exception:
    us = float('nan')
for k, d in acc.items():
    if 'rollout' in k:
        w = sum(d['SQ_WAVES']) / len(d['SQ_WAVES'])
        print('ablate', sys.argv[1], {c: round(sum(v) / len(v) / w / 64, 1) for c, v in d.items() if c != 'SQ_WAVES'}, 'kernel us per 128 steps', round(us, 1))
PY
  done
fi
