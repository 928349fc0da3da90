#!/bin/bash
# Round 5's evidence in one call on the GPU box: tools/round5_evidence.sh gpurun_out/evidence5   (copied to profiles/r05_* afterwards)
out=${1:-gpurun_out/evidence5}
mkdir -p "$out"
export TMPDIR=/tmp
python3 bench.py > "$out/bench_default.log" 2>&1; grep '"metric"' "$out/bench_default.log" > "$out/bench_default.json"
python3 bench.py --gpus 1 --steps 20 --warmup 5 > "$out/bench_steps20.log" 2>&1; grep '"metric"' "$out/bench_steps20.log" > "$out/bench_steps20.json"
# kernel trace + stats of the two commands (learner flows included in the first), then the PMC passes at the launch shapes benchmarked
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/kt" -o bench -- python3 bench.py --no-cpu-baseline --no-other-configs --no-side-measurements > "$out/kt_bench.log" 2>&1
grep '"metric"' "$out/kt_bench.log" > "$out/bench_under_rocprof.json"
find "$out/kt" -name '*kernel_stats.csv' -exec cp {} "$out/kernel_stats.csv" \;
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/kt20" -o bench -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-other-configs --no-side-measurements > "$out/kt20_bench.log" 2>&1
grep '"metric"' "$out/kt20_bench.log" > "$out/bench_steps20_under_rocprof.json"
find "$out/kt20" -name '*kernel_stats.csv' -exec cp {} "$out/kernel_stats_steps20.csv" \;
rm -rf "$out"/kt "$out"/kt20
python3 tools/pmc_collect.py "$out" headline256 headline20 step step16k versus versus16k c3 c4shard c5shard > "$out/pmc_collect.log" 2>&1
# config 3
python3 bench.py --workload MATE-8v8-9.yaml --batch 8192 --policy greedy --steps 1024 --warmup 128 --reps 3 --no-cpu-baseline --no-other-configs > "$out/c3.log" 2>&1; grep '"metric"' "$out/c3.log" > "$out/c3_bench.json"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/c3kt" -o c3 -- python3 bench.py --workload MATE-8v8-9.yaml --batch 8192 --policy greedy --steps 1024 --warmup 128 --reps 3 --no-cpu-baseline --no-other-configs > "$out/c3kt.log" 2>&1
find "$out/c3kt" -name '*kernel_stats.csv' -exec cp {} "$out/c3_kernel_stats.csv" \;
rm -rf "$out/c3kt"
cp mate_amd/lib/kernel_resources.json "$out/kernel_resources.json"
# the learner-versus-greedy step: per-wave phases (profiling build), the flows probe
if [ -f mate_amd/lib/libmate_engine_prof.so ]; then
  MATE_ENGINE_LIB=$PWD/mate_amd/lib/libmate_engine_prof.so python3 tools/versus_phases.py > "$out/versus_phases.txt" 2>&1
  MATE_ENGINE_LIB=$PWD/mate_amd/lib/libmate_engine_prof.so python3 tools/versus_phases.py MATE-4v8-9.yaml 16384 >> "$out/versus_phases.txt" 2>&1
  MATE_ENGINE_LIB=$PWD/mate_amd/lib/libmate_engine_prof.so python3 tools/phase_profile.py > "$out/step_phases.txt" 2>&1
fi
# parity soak (every shape family against the CPU oracle)
bash tools/soak_round.sh > /dev/null 2>&1; cp gpurun_out/soak_final.txt "$out/soak.txt"
ls -la "$out"
