#!/bin/bash
# Everything profiles/ holds for a round, in one call on the GPU box: tools/round_evidence.sh gpurun_out/evidence
out=${1:-gpurun_out/evidence}
mkdir -p "$out"
export TMPDIR=/tmp
python3 bench.py > "$out/bench_default.log" 2>&1; grep '"metric"' "$out/bench_default.log" > "$out/bench_default.json"
python3 bench.py --steps 20 --warmup 5 > "$out/bench_steps20.log" 2>&1; grep '"metric"' "$out/bench_steps20.log" > "$out/bench_steps20.json"
bash tools/profile_round.sh "$out/prof" > "$out/profile_round.log" 2>&1
# config 3: Greedy vs Greedy on the device
python3 bench.py --workload MATE-8v8-9.yaml --batch 8192 --policy greedy --steps 1024 --warmup 128 --reps 3 --no-cpu-baseline --no-other-configs > "$out/c3.log" 2>&1; grep '"metric"' "$out/c3.log" > "$out/c3_bench.json"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/c3kt" -o c3 -- python3 bench.py --workload MATE-8v8-9.yaml --batch 8192 --policy greedy --steps 1024 --warmup 128 --reps 3 --no-cpu-baseline --no-other-configs > "$out/c3kt.log" 2>&1
cp "$out"/c3kt/c3_kernel_stats.csv "$out/c3_kernel_stats.csv" 2>/dev/null || find "$out/c3kt" -name '*kernel_stats.csv' -exec cp {} "$out/c3_kernel_stats.csv" \;
rm -rf "$out/c3kt"
bash tools/sweep.sh "$out/sweep.jsonl" 2> "$out/sweep.err"
cp mate_amd/lib/kernel_resources.json "$out/kernel_resources.json"
# what the observation stores cost by themselves, and what the placement of the blocks does to them (binaries built by mate_amd.build --tools)
[ -x tools/store_roof ] && tools/store_roof 4096 256 6 > "$out/store_roof.txt" 2>&1
[ -x tools/store_vmm ] && tools/store_vmm 8 2 > "$out/store_vmm.txt" 2>&1
[ -x tools/store_contig ] && tools/store_contig 4 > "$out/store_contig.txt" 2>&1
[ -x tools/store_bits ] && tools/store_bits > "$out/store_bits.txt" 2>&1
# per-wave phase clocks (profiling build), the single-step and learner-versus-greedy probes
if [ -f mate_amd/lib/libmate_engine_prof.so ]; then
  MATE_ENGINE_LIB=$PWD/mate_amd/lib/libmate_engine_prof.so python3 tools/rollout_phases.py MATE-4v8-9.yaml 4096 256 > "$out/rollout_phases_4096.txt" 2>&1
  MATE_ENGINE_LIB=$PWD/mate_amd/lib/libmate_engine_prof.so python3 tools/rollout_phases.py MATE-4v8-9.yaml 1024 256 > "$out/rollout_phases_1024.txt" 2>&1
  MATE_ENGINE_LIB=$PWD/mate_amd/lib/libmate_engine_prof.so python3 tools/greedy_phases.py MATE-8v8-9.yaml 8192 32 > "$out/greedy_phases.txt" 2>&1
fi
python3 tools/k1_probe.py > "$out/k1_probe.txt" 2>&1
python3 tools/versus_probe.py > "$out/versus_probe.txt" 2>&1
python3 tools/alloc_probe2.py 5 > "$out/blocks_scattered.txt" 2>&1
MATE_PLAIN_BLOCKS=1 python3 tools/alloc_probe2.py 5 > "$out/blocks_plain.txt" 2>&1
ls -la "$out"
