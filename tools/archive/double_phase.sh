#!/bin/bash
# Exact dynamic instruction share of the idempotent phases of rollout_kernel: the engine rebuilt with one phase executed
# TWICE per step (-DMATE_DOUBLE=bit: 1 draws, 2 cameras, 8 visibility, 32 row-image blocks, 64 row-image store); the
# difference of the per-environment-step counters against the plain build is what the phase costs on real data.
# Build here (tools/double_phase.sh build), run on the GPU box (tools/double_phase.sh run).
cd "$(dirname "$0")/.."
if [ "$1" = build ]; then
  for m in 0 8; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -mllvm -disable-machine-licm -DMATE_DOUBLE=$m \
      -o mate_amd/lib/libmate_engine_dbl$m.so mate_amd/csrc/mate_engine.hip 2>/dev/null &
    if [ $m = 1 ] || [ $m = 32 ]; then wait; fi
  done; wait; ls mate_amd/lib/libmate_engine_dbl*.so
else
  export TMPDIR=/tmp
  for m in 0 8; do
    rm -rf /tmp/pq
    MATE_ENGINE_LIB=$PWD/mate_amd/lib/libmate_engine_dbl$m.so rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVES --output-format csv -d /tmp/pq -o pmc -- python3 bench.py --rollout 256 --steps 1024 --warmup 256 --no-cpu-baseline --no-extras --no-other-configs --reps 1 --rep-warmup 1 > /tmp/pq.log 2>&1
    python3 - $m <<'PY'
import csv, glob, collections, sys
acc = collections.defaultdict(list)
for f in glob.glob('/tmp/pq/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'rollout_kernel' in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
n = 4096 * 256
print('double', sys.argv[1], {c: round(sum(v) / len(v) / n, 1) for c, v in acc.items() if c != 'SQ_WAVES'}, 'launches', len(acc['SQ_WAVES']))
PY
  done
fi
