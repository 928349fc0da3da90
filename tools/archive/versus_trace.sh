#!/bin/bash
# kernel trace of the learner-versus-greedy per-step flow (HIP-graph replays): what a restart group between two steps costs
# tools/versus_trace.sh [batch] [interval]
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
B=${1:-4096}; I=${2:-64}
rm -rf /tmp/vt
rocprofv3 --kernel-trace --output-format csv -d /tmp/vt -o t -- python3 tools/learner_flows_probe.py MATE-4v8-9.yaml $B $I versus > /tmp/vt.log 2>&1
tail -3 /tmp/vt.log
python3 - <<'PY'
import csv, glob, collections
rows = []
for f in glob.glob('/tmp/vt/**/*kernel_trace.csv', recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
print(len(rows), 'launches')
def nm(r):
    return r['Kernel_Name'].split('(')[0][-36:]
# find reset groups inside the graph-replayed region (first 60 % of the trace), print three of them with their neighbours
idx = [i for i, r in enumerate(rows[:int(len(rows) * 0.6)]) if 'reset_kernel' in r['Kernel_Name']]
groups, last = [], -10
for i in idx:
    if i - last > 3: groups.append(i)
    last = i
import statistics
spans = []
for gi in groups[len(groups) // 2: len(groups) // 2 + 40]:
    j = gi
    while j < len(rows) and ('reset_kernel' in rows[j]['Kernel_Name'] or 'fillBuffer' in rows[j]['Kernel_Name']): j += 1
    start = gi
    while start > 0 and 'fillBuffer' in rows[start - 1]['Kernel_Name']: start -= 1
    prev_end = int(rows[start - 1]['End_Timestamp']); next_start = int(rows[j]['Start_Timestamp'])
    spans.append((next_start - prev_end) / 1000.0)
print('restart group, end of the step before -> start of the policy kernel behind: median %.1f us, min %.1f, max %.1f over %d groups' % (statistics.median(spans), min(spans), max(spans), len(spans)))
g = groups[len(groups) // 2]
t0 = int(rows[g - 4]['Start_Timestamp'])
for r in rows[g - 4: g + 12]:
    print('%9.1f %8.1f  %s grid %s wg %s' % ((int(r['Start_Timestamp']) - t0) / 1000.0, (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1000.0, nm(r), r.get('Grid_Size_X', r.get('Grid_Size', '?')), r.get('Workgroup_Size_X', '?')))
PY
