#!/bin/bash
for lib in libmate_engine_base.so libmate_engine.so; do
  echo "== $lib"
  MATE_ENGINE_LIB=$PWD/mate_amd/lib/$lib timeout 120 python tools/launch_probe.py
done
echo "== new lib, generic flow"
MATE_FLOW_GENERIC=1 timeout 120 python tools/launch_probe.py
MATE_FLOW_GENERIC=1 timeout 120 python bench.py --batch 4096 --steps 2000 --warmup 200 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-160
echo "== new lib, no auto reset"
timeout 120 python - <<'PY'
import time, torch
from mate_amd.config import read_config
from mate_amd.engine import Engine
eng = Engine(read_config('MATE-4v8-9.yaml'), 4096, seed=0)
eng.reset()
for ar in (True, False):
    for _ in range(200): eng.step_random(auto_reset=ar)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3000): eng.step_random(auto_reset=ar)
    torch.cuda.synchronize(); print('auto_reset', ar, (time.perf_counter() - t0) / 3000 * 1e6, 'us/step')
PY
