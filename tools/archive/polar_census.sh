#!/bin/bash
# The long census of round 4 (profiles/r04_soak.txt: 4096 x MATE-4v8-9 for 6000 steps, 4096 x MATE-4v2-9 for 3000 -- the two runs in which an
# environment diverged from the oracle) on the shipped build (over-long steps and truncated rays rescaled) and on the -DMATE_POLAR_CLAMP
# variant (both re-made from their polar form, utils.py:223-229, as the oracle does): python -m mate_amd.build --variant polar -DMATE_POLAR_CLAMP first.
out=gpurun_out/polar_census.txt
: > $out
for lib in libmate_engine.so libmate_engine_polar.so; do
  echo "== $lib" >> $out
  MATE_ENGINE_LIB=mate_amd/lib/$lib python tests/soak_vs_oracle.py MATE-4v8-9.yaml 4096 6000 24 2>/dev/null | tail -2 >> $out
  MATE_ENGINE_LIB=mate_amd/lib/$lib python tests/soak_vs_oracle.py MATE-4v2-9.yaml 4096 3000 32 2>/dev/null | tail -2 >> $out
done
# what the polar form costs: the fused rollout and the single step, same box, both builds
for lib in libmate_engine.so libmate_engine_polar.so libmate_engine.so libmate_engine_polar.so; do
  echo "== $lib" >> $out
  MATE_ENGINE_LIB=mate_amd/lib/$lib python bench.py --steps 1024 --no-cpu-baseline --no-other-configs --no-side-measurements --no-extras 2>/dev/null | python3 -c "
import json,sys
l=json.loads(sys.stdin.readline()); print('  rollout_kernel %.1f us per 256-step launch, %.3g env-steps/s' % (l['roofline']['kernel_avg_us'], l['value']))" >> $out
  MATE_ENGINE_LIB=mate_amd/lib/$lib python bench.py --rollout 0 --steps 1024 --no-cpu-baseline --no-other-configs --no-side-measurements --no-extras 2>/dev/null | python3 -c "
import json,sys
l=json.loads(sys.stdin.readline()); print('  step_kernel %.2f us, %.3g env-steps/s' % (l['roofline']['kernel_avg_us'], l['value']))" >> $out
done
cat $out
