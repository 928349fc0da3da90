#!/bin/bash
# per-kernel rocprofv3 stats for two builds of the engine library (A/B)
cd /tmp && export TMPDIR=/tmp
for lib in libmate_engine_base.so libmate_engine.so; do
  export MATE_ENGINE_LIB=$GRAFT_REPO_ROOT/mate_amd/lib/$lib
  rm -rf /tmp/prof_$lib
  rocprofv3 --kernel-trace --stats -d /tmp/prof_$lib -o out -- python3 $GRAFT_REPO_ROOT/bench.py --batch 4096 --steps 2000 --warmup 200 --no-cpu-baseline > /tmp/prof_$lib.log 2>&1
  echo "== $lib"; tail -1 /tmp/prof_$lib.log | cut -c1-200
  f=$(find /tmp/prof_$lib -name "*kernel_stats.csv" | head -1); head -6 $f | cut -c1-220
done
