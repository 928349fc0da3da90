#!/bin/bash
# Vector-instruction mix and wait breakdown of the fused rollout kernel (GPU box): three --pmc passes, per wave per environment-step.
export TMPDIR=/tmp
for pass in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64" \
            "SQ_WAVES SQ_INSTS_VALU_CVT SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU" \
            "SQ_WAVES SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_INSTS_BRANCH" \
            "SQ_WAVES SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM SQ_INST_CYCLES_SMEM SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_ANY"; do
  rm -rf /tmp/pq
  rocprofv3 --kernel-trace --pmc $pass --output-format csv -d /tmp/pq -o pmc -- python3 bench.py --steps 512 --warmup 128 --no-cpu-baseline --no-extras --reps 1 > /tmp/pq.log 2>&1
  python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('/tmp/pq/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r['Kernel_Name'][:40]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in acc.items():
    if 'rollout' in k:
        w = sum(d['SQ_WAVES']) / len(d['SQ_WAVES'])
        print({c: round(sum(v) / len(v) / w / 128, 1) for c, v in d.items() if c != 'SQ_WAVES'})
PY
done
