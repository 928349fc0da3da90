#!/usr/bin/env python3
"""Per-phase cycles per step of the fused Greedy rollout (needs the -DMATE_PHASE_CLOCKS build:
MATE_ENGINE_LIB=mate_amd/lib/libmate_engine_prof.so python tools/greedy_phases.py [workload] [batch] [R])."""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mate_amd.config import read_config  # noqa: E402
from mate_amd.engine import Engine  # noqa: E402
workload = sys.argv[1] if len(sys.argv) > 1 else 'MATE-8v8-9.yaml'
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
R = int(sys.argv[3]) if len(sys.argv) > 3 else 32
eng = Engine(read_config(workload), batch, seed=0)
eng.enable_policies()
eng.reset()
for _ in range(6):
    eng.rollout_greedy(R)
buf = torch.zeros((batch, 16), dtype=torch.int64, device='cuda')
eng.lib.mate_engine_debug_phase_clocks.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
eng.lib.mate_engine_debug_phase_clocks(eng._h, ctypes.c_void_p(buf.data_ptr()))
eng.rollout_greedy(R)
torch.cuda.synchronize()
raw = buf.cpu().numpy().astype(np.float64)
t = raw[:, :13] / R
names = ['draws', 'cameras', 'targets', 'view', 'assign', 'scratch', 'pack', 'loop', 'observe', 'zoom', 'actions', 'communicate', 'choose']
print(f'{workload} batch {batch} R {R}: cycles per step per wave, mean / p50 / p99 over waves')
for i, n in enumerate(names):
    print(f'  {n:11s} {t[:, i].mean():8.0f} {np.percentile(t[:, i], 50):8.0f} {np.percentile(t[:, i], 99):8.0f}')
print(f'  total    {t.sum(axis=1).mean():8.0f}')
print('  s_memtime ticks per microsecond in this launch: %.0f' % (raw[:, 14].sum() / (raw[:, 15].sum() / 100.0)))
