#!/bin/bash
# kernel trace of the config-3 bench: per-launch durations of reset_kernel by grid size (which phase costs what)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
rm -rf /tmp/c3t
rocprofv3 --kernel-trace --output-format csv -d /tmp/c3t -o t -- python3 bench.py --workload MATE-8v8-9.yaml --batch 8192 --policy greedy --steps 1024 --warmup 128 --no-cpu-baseline --no-extras --reps 2 > /tmp/c3t.log 2>&1
python3 - <<'PY'
import csv, glob, collections
rows = []
for f in glob.glob('/tmp/c3t/**/*kernel_trace.csv', recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
print(len(rows), 'launches')
agg = collections.defaultdict(list)
for r in rows:
    name = r['Kernel_Name'].split('(')[0][-40:]
    key = (name, r.get('Grid_Size_X', r.get('Grid_Size', '?')), r.get('Workgroup_Size_X', '?'), r.get('LDS_Block_Size', r.get('LDS_Block_Size_v', '?')))
    agg[key].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1000.0)
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    v2 = sorted(v)
    print(k, 'n', len(v), 'total ms %.2f' % (sum(v) / 1000), 'median us %.1f' % v2[len(v2) // 2], 'max %.1f' % v2[-1])
# the tail of the timeline: a reset group between two rollouts
tail = rows[-40:]
t0 = int(tail[0]['Start_Timestamp'])
for r in tail:
    print('%9.1f %8.1f  %s grid %s' % ((int(r['Start_Timestamp']) - t0) / 1000.0, (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1000.0, r['Kernel_Name'].split('(')[0][-36:], r.get('Grid_Size_X', r.get('Grid_Size', '?'))))
PY
