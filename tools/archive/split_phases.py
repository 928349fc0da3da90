#!/usr/bin/env python3
"""Per-wave phase stamps of the two-wave step kernel (profiling build: -DMATE_PHASE_CLOCKS;
MATE_STEP_SPLIT=1 MATE_ENGINE_LIB=mate_amd/lib/libmate_engine_prof.so python tools/split_phases.py [workload] [batch] [flow])."""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mate_amd.config import read_config  # noqa: E402
from mate_amd.engine import Engine  # noqa: E402
workload = sys.argv[1] if len(sys.argv) > 1 else 'MATE-4v8-9.yaml'
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
flow = sys.argv[3] if len(sys.argv) > 3 else 'random'
os.environ.setdefault('MATE_STEP_SPLIT', '1')
eng = Engine(read_config(workload), batch, seed=0)
eng.reset()
cam = torch.rand((batch, eng.num_cameras, 2), device='cuda') * 4 - 2
tgt = torch.rand((batch, eng.num_targets, 2), device='cuda') * 30 - 15
step = (lambda: eng.step_random(auto_reset=8)) if flow == 'random' else (lambda: eng.step(cam, tgt, auto_reset=8))
for _ in range(50):
    step()
buf = torch.zeros((batch, 16), dtype=torch.int64, device='cuda')
eng.lib.mate_engine_debug_phase_clocks.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
eng.lib.mate_engine_debug_phase_clocks(eng._h, ctypes.c_void_p(buf.data_ptr()))
names = ['load+draws+commit', 'kinematics', 'wait barrier 1', 'view', 'goals / (B: -)', 'wait barrier 2', 'pack + store']
acc = {0: [], 1: []}
life = []
for _ in range(20):
    step()
    torch.cuda.synchronize()
    t = buf.cpu().numpy().astype(np.float64)
    for role in (0, 1):
        acc[role].append(np.diff(t[:, role * 8:role * 8 + 8], axis=1))
    life.append((t[:, [7, 15]].max(axis=1) - t[:, [0, 8]].min(axis=1)))
for role in (0, 1):
    d = np.concatenate(acc[role])
    print('wave', 'AB'[role], '(cycles: mean / p50 / p90 / max)')
    for i, n in enumerate(names):
        print(f'  {n:20s} {d[:, i].mean():8.0f} {np.percentile(d[:, i], 50):8.0f} {np.percentile(d[:, i], 90):8.0f} {d[:, i].max():8.0f}')
    print(f'  {"wave life":20s} {d.sum(axis=1).mean():8.0f} {np.percentile(d.sum(axis=1), 50):8.0f} {np.percentile(d.sum(axis=1), 90):8.0f} {d.sum(axis=1).max():8.0f}')
life = np.concatenate(life)
print(f'environment (first start .. last end of its two waves): mean {life.mean():.0f}  p50 {np.percentile(life, 50):.0f}  p90 {np.percentile(life, 90):.0f}  max {life.max():.0f}')
