export TMPDIR=/tmp
rm -rf /tmp/kt
rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -o b -- python3 bench.py --no-cpu-baseline --no-extras > /tmp/kt.log 2>&1
python3 - <<'PY'
import csv, glob, json
f = glob.glob('/tmp/kt/**/*kernel_trace.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if 'rollout_kernel' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
d = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in rows]
gap = [(int(rows[i + 1]['Start_Timestamp']) - int(rows[i]['End_Timestamp'])) / 1e3 for i in range(len(rows) - 1)]
print('launches', len(d), 'avg all', sum(d) / len(d))
print('first 2', d[:2], 'priming avg', sum(d[2:-80]) / len(d[2:-80]), 'timed (last 80) avg', sum(d[-80:]) / 80)
print('timed per rep avg', [round(sum(d[-80 + 16 * i:-80 + 16 * i + 16] if -80 + 16 * i + 16 != 0 else d[-16:]) / 16, 1) for i in range(5)])
print('gaps between consecutive rollout launches in the last rep (us):', [round(g, 1) for g in gap[-15:]])
line = [l for l in open('/tmp/kt.log') if l.startswith('{')][-1]
j = json.loads(line); print('bench kernel_avg_us', j['roofline']['kernel_avg_us'], 'rep_ms', j['rep_ms'])
PY
