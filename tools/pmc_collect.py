#!/usr/bin/env python3
"""PMC counters of every dominant kernel AT THE LAUNCH SHAPE it is benchmarked at (GPU box, through gpurun):

    python3 tools/pmc_collect.py gpurun_out/pmc_r03 [case ...]

One case = one bench.py command whose launches of the kernel of interest all have ONE length (--rollout R --steps 4R
--warmup R), profiled in separate `rocprofv3 --kernel-trace --pmc <group>` passes (never combined with other trace
domains; the program goes directly behind `--`).  Output: <out>/pmc_summary.json keyed "<kernel>@<steps per launch>" with
per-LAUNCH averages (FETCH_SIZE / WRITE_SIZE in KiB as rocprofv3 reports them -- on gfx950 FETCH_SIZE counts half of the
bytes read, tools/pmc_calibrate.py -- and SQ_* summed over the chip, SQ_*_CYCLES / SQ_ACTIVE_* / SQ_WAIT_* in quad-cycles),
`env_steps_per_launch`, and derived per-environment-step instruction counts.  profiles/latest_pmc.json is a copy;
bench.py's roofline.traffic reads the entry of the launch length it ran (no scaling between launch lengths)."""
import collections
import csv
import glob
import json
import os
import subprocess
import sys

GROUPS = [
    ['FETCH_SIZE'],
    ['WRITE_SIZE'],
    ['SQ_WAVES', 'SQ_WAVE_CYCLES', 'SQ_BUSY_CYCLES', 'SQ_INSTS_VALU', 'SQ_INSTS_SALU', 'SQ_INSTS_LDS', 'SQ_ACTIVE_INST_VALU', 'SQ_WAIT_INST_ANY'],
    ['SQ_WAIT_ANY', 'SQ_ACTIVE_INST_ANY', 'SQ_INSTS_VMEM_RD', 'SQ_INSTS_VMEM_WR', 'SQ_ACTIVE_INST_LDS', 'SQ_LDS_BANK_CONFLICT', 'SQ_WAIT_INST_LDS', 'GRBM_GUI_ACTIVE'],
    # round 5: what the LDS array itself does (SQ_LDS_IDX_ACTIVE = all LDS-array cycles, SQ_LDS_BANK_CONFLICT = the extra ones: their
    # quotient is the share of array cycles lost to conflicts; MI355X_MICROARCH.md, "LDS"), and how many lanes a vector instruction
    # has switched on (SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU, both in quad-cycles: 64 = every lane of every instruction)
    ['SQ_LDS_IDX_ACTIVE', 'SQ_LDS_BANK_CONFLICT', 'SQ_LDS_ADDR_CONFLICT', 'SQ_LDS_UNALIGNED_STALL', 'SQ_INSTS_LDS'],
    ['SQ_THREAD_CYCLES_VALU', 'SQ_ACTIVE_INST_VALU', 'SQ_INSTS_VALU', 'SQ_INST_CYCLES_VMEM_WR', 'SQ_INST_CYCLES_VMEM_RD'],
]
# name: (kernel substring, steps per launch, environments, bench arguments)
CASES = {
    'headline256': ('rollout_kernel', 256, 4096, ['--rollout', '256', '--steps', '1024', '--warmup', '256']),
    'headline20': ('rollout_kernel', 20, 4096, ['--rollout', '20', '--steps', '20', '--warmup', '20', '--rollout-reset-interval', '6']),
    'step': ('step_kernel', 1, 4096, ['--rollout', '0', '--steps', '512', '--warmup', '64']),
    'step16k': ('step_kernel', 1, 16384, ['--batch', '16384', '--rollout', '0', '--steps', '256', '--warmup', '64']),
    'c3': ('rollout_greedy_kernel', 48, 8192, ['--workload', 'MATE-8v8-9.yaml', '--batch', '8192', '--policy', 'greedy', '--rollout', '48', '--steps', '384', '--warmup', '48']),
    'c4shard': ('rollout_kernel', 256, 8192, ['--workload', 'MATE-4v8-0.yaml', '--batch', '8192', '--rollout', '256', '--steps', '1024', '--warmup', '256']),
    'c5shard': ('rollout_kernel', 256, 4096, ['--workload', 'MATE-Navigation.yaml', '--batch', '4096', '--rollout', '256', '--steps', '1024', '--warmup', '256']),
    # BASELINE configs 4 and 5 whole on ONE GPU (sixteen / eight generations of resident waves)
    # the learner-versus-greedy step (step_greedy_kernel): a target script of its own, not a bench.py command
    'versus': ('step_greedy_kernel', 1, 4096, ['4096', '300'], 'tools/versus_target.py'),
    'versus16k': ('step_greedy_kernel', 1, 16384, ['16384', '200'], 'tools/versus_target.py'),
    # round 6: four environments per wave against one (tools/subwave_target.py): the target trainers' MATE-2v4-0 FrameSkip(10) flow, BASELINE config 1's scenario fused
    'sub_target10_one': ('rollout_greedy_kernel', 10, 16384, ['target10', 'MATE-2v4-0.yaml', '16384', '40', '0'], 'tools/subwave_target.py'),
    'sub_target10_four': ('rollout_greedy_kernel', 10, 16384, ['target10', 'MATE-2v4-0.yaml', '16384', '40', '1'], 'tools/subwave_target.py'),
    'sub_4v2_one': ('rollout_kernel', 64, 16384, ['random64', 'MATE-4v2-9.yaml', '16384', '12', '0'], 'tools/subwave_target.py'),
    'sub_4v2_four': ('rollout_kernel', 64, 16384, ['random64', 'MATE-4v2-9.yaml', '16384', '12', '1'], 'tools/subwave_target.py'),
    'c4full': ('rollout_kernel', 64, 65536, ['--workload', 'MATE-4v8-0.yaml', '--batch', '65536', '--rollout', '64', '--steps', '256', '--warmup', '64']),
    'c5full': ('rollout_kernel', 64, 32768, ['--workload', 'MATE-Navigation.yaml', '--batch', '32768', '--rollout', '64', '--steps', '256', '--warmup', '64']),
}


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/pmc'
    names = sys.argv[2:] or ['headline256', 'headline20', 'step', 'c3']
    os.makedirs(out, exist_ok=True)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, TMPDIR='/tmp')
    summary = {}
    for name in names:
        kernel, steps, envs, args = CASES[name][:4]
        script = CASES[name][4] if len(CASES[name]) > 4 else None
        acc = collections.defaultdict(list)
        only = os.environ.get('PMC_GROUPS')       # e.g. PMC_GROUPS=4,5: only those counter groups (attribution runs with experiment builds)
        for gi, group in enumerate(GROUPS):
            if only and str(gi) not in only.split(','):
                continue
            d = os.path.join(out, f'raw_{name}_{gi}')
            target = (['python3', os.path.join(root, script)] + args) if script else (
                ['python3', os.path.join(root, 'bench.py'), '--reps', '1', '--rep-warmup', '1', '--no-cpu-baseline', '--no-extras', '--no-other-configs',
                 '--no-side-measurements'] + args)
            cmd = ['rocprofv3', '--kernel-trace', '--pmc'] + group + ['--output-format', 'csv', '-d', d, '-o', 'pmc', '--'] + target
            with open(os.path.join(out, f'{name}_{gi}.log'), 'w') as log:
                subprocess.call(cmd, env=env, stdout=log, stderr=subprocess.STDOUT, cwd=root)
            for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
                for row in csv.DictReader(open(f)):
                    if kernel in row['Kernel_Name'] and 'greedy' not in row['Kernel_Name'].replace(kernel, ''):
                        acc[row['Counter_Name']].append(float(row['Counter_Value']))
            subprocess.call(['rm', '-rf', d])
        if not acc:
            continue
        entry = {c: sum(v) / len(v) for c, v in acc.items()}
        entry['launches_sampled'] = len(next(iter(acc.values())))
        entry['env_steps_per_launch'] = envs * steps
        entry['bench_args'] = ' '.join(args)
        waves = entry.get('SQ_WAVES', 0.0)
        if waves:      # SQ_INSTS_* count wave-instructions; one wave = one environment in these kernels
            for c in ('SQ_INSTS_VALU', 'SQ_INSTS_SALU', 'SQ_INSTS_LDS'):
                entry[c + '_per_env_step'] = entry[c] / envs / steps
            if entry.get('SQ_WAVE_CYCLES'):
                entry['valu_active_fraction_of_wave_cycles'] = entry['SQ_ACTIVE_INST_VALU'] / entry['SQ_WAVE_CYCLES']
        if entry.get('SQ_LDS_IDX_ACTIVE'):
            entry['lds_conflict_share_of_array_cycles'] = entry['SQ_LDS_BANK_CONFLICT'] / entry['SQ_LDS_IDX_ACTIVE']
        if entry.get('SQ_THREAD_CYCLES_VALU') and entry.get('SQ_ACTIVE_INST_VALU'):
            entry['mean_active_lanes_per_valu_instruction'] = entry['SQ_THREAD_CYCLES_VALU'] / entry['SQ_ACTIVE_INST_VALU']
        entry['hbm_bytes_per_launch'] = (2.0 * entry.get('FETCH_SIZE', 0.0) + entry.get('WRITE_SIZE', 0.0)) * 1024.0
        summary[f'{kernel}@{steps}' + ('' if name in ('headline256', 'headline20', 'step', 'c3', 'versus') else ':' + name)] = entry
        print(name, json.dumps({k: (round(v, 1) if isinstance(v, float) else v) for k, v in entry.items()}, sort_keys=True), flush=True)
    json.dump(summary, open(os.path.join(out, 'pmc_summary.json'), 'w'), indent=1, sort_keys=True)


if __name__ == '__main__':
    main()
