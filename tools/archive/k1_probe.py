#!/usr/bin/env python3
"""One step per launch: step_kernel against the fused rollout kernels launched with a single step (dispatch-event kernel time
and end-to-end time per step).  python tools/k1_probe.py [workload] [batch]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mate_amd.config import read_config  # noqa: E402
from mate_amd.engine import Engine  # noqa: E402
workload = sys.argv[1] if len(sys.argv) > 1 else 'MATE-4v8-9.yaml'
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
eng = Engine(read_config(workload), batch, seed=0)
eng.reset()
for name, fn in (('step_random (step_kernel)', lambda: eng.step_random(auto_reset=8)), ('rollout_random(1)', lambda: eng.rollout_random(1, auto_reset=8)),
                 ('rollout_random(2)', lambda: eng.rollout_random(2, auto_reset=8)), ('rollout_random(4)', lambda: eng.rollout_random(4, auto_reset=8))):
    for _ in range(64):
        fn()
    torch.cuda.synchronize()
    eng.kernel_time(enable=1)
    t0 = time.perf_counter()
    for _ in range(512):
        fn()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ms, n = eng.kernel_time(enable=False)
    print(f'{workload} x {batch} {name}: {dt / 512 * 1e6:.1f} us per call end to end, kernel {ms * 1e3:.2f} us x {n}')
