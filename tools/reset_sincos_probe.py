#!/usr/bin/env python3
"""What a sin / cos table for the 360 integer-degree rays of an occlusion table could save at most (review item: "take sin/cos of
the 360 integer-degree rays from a constant table"): the whole-batch reset and the restart of a tenth of the batch on MATE-8v8-9 x
8192, shipped build against a build in which those 360 evaluations are free (-DMATE_ABLATE_DEGREE_SINCOS: wrong tables, valid timing).
python tools/reset_sincos_probe.py lib_shipped.so lib_ablate.so"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mate_amd import _native  # noqa: E402
from mate_amd.config import read_config  # noqa: E402
from mate_amd.engine import Engine  # noqa: E402
cfg = read_config('MATE-8v8-9.yaml')
n = 8192
for path in sys.argv[1:3]:
    _native.lib, _native.LIB_PATH = None, os.path.abspath(path)
    eng = Engine(cfg, n, seed=0)
    eng.reset()
    mask = torch.zeros(n, dtype=torch.uint8, device='cuda')
    mask[::10] = 1
    out = []
    for what, fn in (('whole batch', lambda: eng.reset()), ('every 10th environment', lambda: eng.reset(env_mask=mask))):
        times = []
        for _ in range(9):
            torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); times.append(time.perf_counter() - t0)
        out.append(f'{what} {sorted(times[2:])[3] * 1e3:.3f} ms')
    print(os.path.basename(path), '; '.join(out), flush=True)
    eng.close(); del eng
