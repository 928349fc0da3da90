#!/usr/bin/env python3
"""What a fused rollout launch costs before its first step and after its last (profiling build, -DMATE_PHASE_CLOCKS):
MATE_ENGINE_LIB=mate_amd/lib/libmate_engine_prof.so python tools/rollout_prologue.py [workload] [batch] [R ...]"""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mate_amd.config import read_config  # noqa: E402
from mate_amd.engine import Engine  # noqa: E402
workload = sys.argv[1] if len(sys.argv) > 1 else 'MATE-4v8-9.yaml'
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
Rs = [int(x) for x in sys.argv[3:]] or [1, 20]
eng = Engine(read_config(workload), batch, seed=0)
eng.reset()
buf = torch.zeros((batch, 16), dtype=torch.int64, device='cuda')
eng.lib.mate_engine_debug_phase_clocks.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
eng.lib.mate_engine_debug_phase_clocks(eng._h, ctypes.c_void_p(buf.data_ptr()))
for R in Rs:
    for _ in range(4):
        eng.rollout_random(R, auto_reset=8)
    torch.cuda.synchronize()
    t = buf.cpu().numpy().astype(np.float64)
    pro = np.stack([t[:, 8], t[:, 9] - t[:, 8], t[:, 10] - t[:, 9], t[:, 11] - t[:, 10], t[:, 12]], axis=1)
    steps = t[:, :8].sum(axis=1)
    print(f'{workload} x {batch}, {R}-step launch: cycles per wave, mean / p50 / p90')
    for i, n in enumerate(['records + entity table', 'lane roles', 'row-image statics', 'screen seed, draw role, held state', 'epilogue (state back)']):
        print(f'  {n:36s} {pro[:, i].mean():8.0f} {np.percentile(pro[:, i], 50):8.0f} {np.percentile(pro[:, i], 90):8.0f}')
    print(f'  {"the steps":36s} {steps.mean():8.0f}   ({steps.mean() / R:.0f} per step)')
    print(f'  wave life {t[:, 14].mean():.0f} (p90 {np.percentile(t[:, 14], 90):.0f}, max {t[:, 14].max():.0f}); ticks per us {t[:, 14].sum() / (t[:, 15].sum() / 100.0):.0f}')
