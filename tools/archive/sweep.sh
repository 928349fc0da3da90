#!/bin/bash
# env-steps/s of the step path for the BASELINE.json scenarios (per-GPU shard sizes); one JSON line per case:
# the default flow (random policy: fused 32-step rollouts; greedy: one launch per step) and, for the random policy,
# the one-launch-per-step rate measured in the same run.
out=${1:-/dev/stdout}
{
for spec in "MATE-2v4-0.yaml 4096 random" "MATE-4v4-9.yaml 4096 random" "MATE-8v8-0.yaml 4096 random" "MATE-4v2-9.yaml 4096 random" "MATE-4v8-9.yaml 4096 random" "MATE-4v8-9.yaml 16384 random" "MATE-4v8-9.yaml 65536 random" "MATE-8v8-9.yaml 8192 random" "MATE-8v8-9.yaml 8192 greedy" "MATE-4v8-0.yaml 8192 random" "MATE-4v8-0.yaml 65536 random" "MATE-Navigation.yaml 4096 random" "MATE-Navigation.yaml 32768 random"; do
  set -- $spec
  python bench.py --workload $1 --batch $2 --policy $3 --steps 1024 --warmup 128 --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
r=d['roofline']; p=d.get('per_step_launch')
o={'workload':'$1','batch':$2,'policy':'$3','steps_per_launch':d['config']['steps_per_launch'],'env_steps_per_s':round(d['value']),'us_per_step':round(d['ms_per_step']*1e3,2),'kernel':r['kernel'].split('<')[0],'kernel_us_per_launch':round(r['kernel_avg_us'],2),'algorithmic_GBps':round(r['achieved'],1),'roofline_frac':round(r['frac'],3),'roofline_frac_resident':round(r['frac_resident'],3)}
if p: o['per_step_launch']={'env_steps_per_s':round(p['value']),'step_kernel_us':round(p['kernel_avg_us'],2),'roofline_frac':round(p['roofline_frac'],3)}
print(json.dumps(o))"
done
} > $out
