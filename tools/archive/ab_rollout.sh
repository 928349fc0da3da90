#!/bin/bash
# Interleaved A/B of engine builds on the fused rollout (ABBA order, so that a drift of the box's clocks does not favour one):
#   [W=MATE-4v8-9.yaml B=4096 R=256 N=3 P="--policy greedy"] tools/ab_rollout.sh libA.so libB.so
# prints every run's kernel time and, at the end, the median per build.
cd "$(dirname "$0")/.."
A=$1; B=$2; : > /tmp/ab_rollout.txt
for rep in $(seq 1 ${N:-3}); do for lib in $A $B $B $A; do
  MATE_ENGINE_LIB=$PWD/$lib python3 bench.py --workload ${W:-MATE-4v8-9.yaml} --batch ${B_:-${BATCH:-4096}} --rollout ${R:-256} $P --steps $((4 * ${R:-256})) --warmup ${R:-256} --no-cpu-baseline --no-extras --no-other-configs --reps 3 2>/dev/null | tail -1 |
    python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib', round(d['roofline']['kernel_avg_us'],2), '%.4g' % d['value'])" | tee -a /tmp/ab_rollout.txt
done; done
python3 - <<'PY'
import collections, statistics
t = collections.defaultdict(list)
for line in open('/tmp/ab_rollout.txt'):
    lib, us, v = line.split()
    t[lib].append(float(us))
for lib, xs in t.items():
    print(lib, 'median kernel us %.1f' % statistics.median(xs), 'min %.1f' % min(xs), 'n', len(xs))
PY
