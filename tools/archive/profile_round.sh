#!/bin/bash
# Collects the per-round rocprofv3 evidence on the GPU box (run through gpurun):
#   tools/profile_round.sh gpurun_out/profiles_rNN
# 1) kernel-trace + stats of the default bench command and of the driver's own command line (--steps 20 --warmup 5);
# 2) PMC counters of every dominant kernel at the launch shape it is benchmarked at (tools/pmc_collect.py: separate
#    passes with --kernel-trace only, the program directly behind `--`): rollout_kernel at 256 and 20 steps per launch,
#    step_kernel, rollout_greedy_kernel (config 3), and the config-4 / config-5 shards.
out=${1:-gpurun_out/profiles}
mkdir -p "$out"
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/kt" -o bench -- python3 bench.py --no-cpu-baseline --no-other-configs > "$out/kt_bench.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/kt20" -o bench -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-other-configs > "$out/kt20_bench.log" 2>&1
grep '"metric"' "$out/kt20_bench.log" > "$out/bench_steps20_under_rocprof.json"
find "$out/kt20" -name '*kernel_stats.csv' -exec cp {} "$out/kernel_stats_steps20.csv" \;
grep '"metric"' "$out/kt_bench.log" > "$out/bench_under_rocprof.json"
find "$out/kt" -name '*kernel_stats.csv' -exec cp {} "$out/kernel_stats.csv" \;
rm -rf "$out"/kt "$out"/kt20
python3 tools/pmc_collect.py "$out" headline256 headline20 step c3 c4shard c5shard > "$out/pmc_collect.log" 2>&1
head -4 "$out/kernel_stats.csv" | cut -c1-180
cut -c1-200 "$out/bench_under_rocprof.json"
