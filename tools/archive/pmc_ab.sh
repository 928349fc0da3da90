#!/bin/bash
# Dynamic instruction counters of the headline rollout for each engine build given (one SQ counter pass each):
#   tools/pmc_ab.sh libA.so libB.so ...   ->  vector / scalar / LDS instructions per environment-step, VALU-busy share
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
for lib in "$@"; do
  d=/tmp/pmc_ab_$$; rm -rf $d
  MATE_ENGINE_LIB=$PWD/$lib rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY \
    --output-format csv -d $d -o pmc -- python3 bench.py --reps 1 --rep-warmup 1 --no-cpu-baseline --no-extras --no-other-configs \
    --workload ${W:-MATE-4v8-9.yaml} --batch ${BATCH:-4096} $P --rollout ${R:-256} --steps $((2 * ${R:-256})) --warmup ${R:-256} > /tmp/pmc_ab.log 2>&1
  python3 - "$d" "$lib" "${K:-rollout_kernel}" "${R:-256}" "${BATCH:-4096}" <<'PY'
import csv, glob, sys, collections
d, lib, kernel, steps, envs = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4]), int(sys.argv[5])
acc = collections.defaultdict(list)
for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        if kernel in row['Kernel_Name']:
            acc[row['Counter_Name']].append(float(row['Counter_Value']))
m = {k: sum(v) / len(v) for k, v in acc.items()}
n = steps * envs
print(lib, 'launches', len(acc.get('SQ_WAVES', [])), ' per env-step: VALU %.1f SALU %.1f LDS %.1f' % (m['SQ_INSTS_VALU'] / n, m['SQ_INSTS_SALU'] / n, m['SQ_INSTS_LDS'] / n),
      ' wave cycles/step %.0f' % (4 * m['SQ_WAVE_CYCLES'] / n), ' VALU busy %.3f' % (4 * m['SQ_ACTIVE_INST_VALU'] / m['SQ_WAVE_CYCLES'] * 4 if False else m['SQ_ACTIVE_INST_VALU'] * 4 / m['SQ_WAVE_CYCLES'] * 4 / 4))
PY
  rm -rf $d
done
