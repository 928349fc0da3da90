#!/usr/bin/env python3
"""The per-step kernel as one wave per environment (step_kernel) and as two (step_split_kernel, MATE_STEP_SPLIT=1): dispatch-event
kernel time and end-to-end time per step, on-device random policy and caller-supplied f32 actions, one reset launch per 8 steps.
    python tools/split_probe.py [workload] [batch] [stagger digits ...]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mate_amd.config import read_config  # noqa: E402
from mate_amd.engine import Engine  # noqa: E402
workload = sys.argv[1] if len(sys.argv) > 1 else 'MATE-4v8-9.yaml'
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
staggers = sys.argv[3:] or ['default']
for split in os.environ.get('SPLITS', '0 1 2 0 1 2').split():
    for stagger in staggers:
        os.environ['MATE_STEP_SPLIT'] = split
        if stagger != 'default':
            os.environ['MATE_STAGGER'] = stagger
        eng = Engine(read_config(workload), batch, seed=0)
        os.environ.pop('MATE_STEP_SPLIT'); os.environ.pop('MATE_STAGGER', None)
        eng.reset()
        cam = torch.rand((batch, eng.num_cameras, 2), device='cuda') * 4 - 2
        tgt = torch.rand((batch, eng.num_targets, 2), device='cuda') * 30 - 15
        for name, fn in (('step_random', lambda: eng.step_random(auto_reset=8)), ('step(f32 actions)', lambda: eng.step(cam, tgt, auto_reset=8))):
            for _ in range(128):
                fn()
            torch.cuda.synchronize()
            eng.kernel_time(enable=1)
            t0 = time.perf_counter()
            for _ in range(1024):
                fn()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            ms, n = eng.kernel_time(enable=False)
            print(f'{workload} x {batch} split={split} stagger={stagger} {name}: {dt / 1024 * 1e6:.2f} us per step end to end, kernel {ms * 1e3:.2f} us x {n}', flush=True)
        del eng
