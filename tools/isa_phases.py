#!/usr/bin/env python3
"""Static instruction mix per phase of a fused rollout kernel.

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -mllvm -disable-machine-licm \
          --cuda-device-only -DMATE_ISA_MARKS -S -o marked.s mate_amd/csrc/mate_engine.hip
    python tools/isa_phases.py marked.s [mangled-name-prefix]

-DMATE_ISA_MARKS turns the ROLL_STAMP phase boundaries of rollout_kernel into assembler comments; this script
cuts the kernel's text at them and counts VALU / SALU / LDS / VMEM instructions per segment (static counts: every
basic block once, rare paths included -- the dynamic mix comes from tools/pmc_mix.sh)."""
import collections
import re
import sys

path = sys.argv[1]
prefix = sys.argv[2] if len(sys.argv) > 2 else '_ZN4mate14rollout_kernelIfNS_10FixedShapeILi4ELi8ELi9ELb0EEELi1EEE'
names = {0: 'draws', 1: 'cameras', 2: 'targets', 3: 'visibility', 4: 'goals', 5: 'scratch', 6: 'pack', 7: 'loop head'}
seg, inside = 'prologue', False
counts = collections.OrderedDict()
ops = collections.defaultdict(collections.Counter)
for line in open(path):
    if line.startswith(prefix):
        inside = True
        continue
    if not inside:
        continue
    if line.startswith('.Lfunc_end'):
        break
    m = re.search(r'MATE_PHASE_END (\d+)', line)
    if m:
        seg = 'after ' + names[int(m.group(1))]
        continue
    t = line.strip().split()
    if not t or t[0].startswith(('.', ';')) or t[0].endswith(':'):
        continue
    op = t[0]
    kind = ('VALU' if op.startswith('v_') else 'SALU' if op.startswith('s_') else 'LDS' if op.startswith('ds_')
            else 'VMEM' if op.startswith(('global_', 'buffer_', 'flat_', 'scratch_')) else 'other')
    counts.setdefault(seg, collections.Counter())[kind] += 1
    ops[seg][op] += 1
for seg, c in counts.items():
    print(f'{seg:18s} VALU {c["VALU"]:5d}  SALU {c["SALU"]:5d}  LDS {c["LDS"]:4d}  VMEM {c["VMEM"]:4d}')
    if len(sys.argv) > 3:
        print('    ' + ', '.join(f'{o} {n}' for o, n in ops[seg].most_common(14)))
