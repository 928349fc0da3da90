#!/bin/bash
# Collects the per-round rocprofv3 evidence on the GPU box (run through gpurun):
#   tools/profile_round.sh gpurun_out/profiles_rNN
# 1) kernel-trace + stats of the default bench command, 2) PMC passes (separate runs, --kernel-trace only).
out=${1:-gpurun_out/profiles}
mkdir -p "$out"
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/kt" -o bench -- python3 bench.py --no-cpu-baseline > "$out/kt_bench.log" 2>&1
# the driver's own command line as well (one 20-step launch per repetition)
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/kt20" -o bench -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > "$out/kt20_bench.log" 2>&1
grep '"metric"' "$out/kt20_bench.log" > "$out/bench_steps20_under_rocprof.json"
cp "$out"/kt20/bench_kernel_stats.csv "$out/kernel_stats_steps20.csv" 2>/dev/null
grep '"metric"' "$out/kt_bench.log" > "$out/bench_under_rocprof.json"
cp "$out"/kt/bench_kernel_stats.csv "$out/kernel_stats.csv" 2>/dev/null
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY" "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE"; do
  tag=$(echo $pass | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $pass --output-format csv -d "$out/pmc_$tag" -o pmc -- python3 bench.py --steps 1024 --warmup 128 --reps 1 --no-cpu-baseline --no-extras > "$out/pmc_$tag.log" 2>&1
  rocprofv3 --kernel-trace --pmc $pass --output-format csv -d "$out/pmc_${tag}_step" -o pmc -- python3 bench.py --rollout 0 --steps 512 --warmup 64 --reps 1 --no-cpu-baseline --no-extras > "$out/pmc_${tag}_step.log" 2>&1
done
python3 - "$out" <<'PY'
import csv, glob, json, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + '/pmc_*/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        name = row['Kernel_Name']
        key = next((k for k in ('step_kernel', 'rollout_kernel', 'reset_kernel') if k in name), None)
        if key:
            acc[key][row['Counter_Name']].append(float(row['Counter_Value']))
summary = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in acc.items()}
for k, d in summary.items():
    d['launches_sampled'] = len(next(iter(acc[k].values())))
if 'rollout_kernel' in summary:
    summary['rollout_kernel']['env_steps_per_launch'] = 4096 * 128      # bench.py --steps 1024: eight 128-step launches per repetition
if 'step_kernel' in summary:
    summary['step_kernel']['env_steps_per_launch'] = 4096
json.dump(summary, open(out + '/pmc_summary.json', 'w'), indent=1, sort_keys=True)
print(json.dumps(summary.get('rollout_kernel', summary.get('step_kernel', {})), sort_keys=True))
PY
rm -rf "$out"/kt "$out"/kt20 "$out"/pmc_*/
head -4 "$out/kernel_stats.csv" | cut -c1-180
cat "$out/bench_under_rocprof.json" | cut -c1-200
