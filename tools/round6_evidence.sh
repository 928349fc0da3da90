#!/bin/bash
# Round 6's evidence in one call on the GPU box: tools/round6_evidence.sh gpurun_out/evidence6   (copied to profiles/r06_* afterwards)
out=${1:-gpurun_out/evidence6}
mkdir -p "$out"
export TMPDIR=/tmp
# the two bench commands: the default run and the driver's own (the LAST stdout line is the headline; the full record goes to --details)
python3 bench.py --details "$out/bench_default_details.json" > "$out/bench_default.out" 2> "$out/bench_default.err"; tail -n 1 "$out/bench_default.out" > "$out/bench_default.json"
python3 bench.py --gpus 1 --steps 20 --warmup 5 --details "$out/bench_steps20_details.json" > "$out/bench_steps20.out" 2> "$out/bench_steps20.err"; tail -n 1 "$out/bench_steps20.out" > "$out/bench_steps20.json"
# kernel trace + stats of the same two commands, then the PMC passes at the launch shapes benchmarked
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/kt" -o bench -- python3 bench.py --no-cpu-baseline --no-other-configs --no-side-measurements --details "$out/kt_details.json" > "$out/kt_bench.log" 2>&1
grep '^{"metric"' "$out/kt_bench.log" > "$out/bench_under_rocprof.json"
find "$out/kt" -name '*kernel_stats.csv' -exec cp {} "$out/kernel_stats.csv" \;
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/kt20" -o bench -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-other-configs --no-side-measurements --details "$out/kt20_details.json" > "$out/kt20_bench.log" 2>&1
grep '^{"metric"' "$out/kt20_bench.log" > "$out/bench_steps20_under_rocprof.json"
find "$out/kt20" -name '*kernel_stats.csv' -exec cp {} "$out/kernel_stats_steps20.csv" \;
# the target trainers' flow (MATE-2v4-0, FrameSkip(10), 16384 environments), four environments per wave and one
for on in 1 0; do
  rocprofv3 --kernel-trace --stats --output-format csv -d "$out/ktsub$on" -o sub -- python3 tools/subwave_target.py target10 MATE-2v4-0.yaml 16384 200 $on > "$out/ktsub$on.log" 2>&1
  find "$out/ktsub$on" -name '*kernel_stats.csv' -exec cp {} "$out/kernel_stats_target10_sub$on.csv" \;
done
rm -rf "$out"/kt "$out"/kt20 "$out"/ktsub0 "$out"/ktsub1
python3 tools/pmc_collect.py "$out" headline256 headline20 step step16k versus versus16k c3 c4shard c5shard sub_target10_one sub_target10_four sub_4v2_one sub_4v2_four > "$out/pmc_collect.log" 2>&1
rm -f "$out"/*_[0-9].log
# four per wave against one, every small scenario, both policies
python3 tools/subwave_probe.py --batches 8192,16384,65536 --out "$out/subwave_probe.json" > "$out/subwave_probe.txt" 2>&1
# config 3
python3 bench.py --workload MATE-8v8-9.yaml --batch 8192 --policy greedy --steps 1024 --warmup 128 --reps 3 --no-cpu-baseline --no-other-configs --details "$out/c3_details.json" > "$out/c3.out" 2> "$out/c3.err"; tail -n 1 "$out/c3.out" > "$out/c3_bench.json"
cp mate_amd/lib/kernel_resources.json "$out/kernel_resources.json"
if [ -f mate_amd/lib/libmate_engine_prof.so ]; then
  MATE_ENGINE_LIB=$PWD/mate_amd/lib/libmate_engine_prof.so python3 tools/subwave_phases.py MATE-2v4-0.yaml 16384 10 target > "$out/subwave_phases.txt" 2>&1
fi
# parity soak (every shape family against the CPU oracle)
bash tools/soak_round.sh > /dev/null 2>&1; cp gpurun_out/soak_final.txt "$out/soak.txt"
ls -la "$out"
