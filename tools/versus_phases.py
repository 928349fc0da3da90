#!/usr/bin/env python3
"""Per-phase wave cycles of step_greedy_kernel (the one-launch learner-versus-greedy step); needs the profiling build:
python -m mate_amd.build --prof; MATE_ENGINE_LIB=mate_amd/lib/libmate_engine_prof.so python tools/versus_phases.py [workload] [batch]"""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mate_amd.config import read_config  # noqa: E402
from mate_amd.engine import Engine  # noqa: E402

workload = sys.argv[1] if len(sys.argv) > 1 else 'MATE-4v8-9.yaml'
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
eng = Engine(read_config(workload), batch, seed=0)
eng.enable_policies()
eng.reset()
mine = torch.zeros((batch, eng.num_cameras, 2), device='cuda')
for _ in range(50):
    eng.step_versus_greedy('camera', mine, auto_reset=32)
buf = torch.zeros((batch, 16), dtype=torch.int64, device='cuda')
eng.lib.mate_engine_debug_phase_clocks.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
eng.lib.mate_engine_debug_phase_clocks(eng._h, ctypes.c_void_p(buf.data_ptr()))
names = ['records+draws', 'entity table', 'agents', 'kinematics', 'view', 'goals', 'rows', 'state stores']
rows = []
for _ in range(20):
    eng.step_versus_greedy('camera', mine, auto_reset=32)
    torch.cuda.synchronize()
    assert eng.last_flow == 4, eng.last_flow
    rows.append(buf.cpu().numpy().astype(np.float64))
t = np.concatenate(rows)
t = t[t[:, 8] > t[:, 0]]
d = np.diff(t[:, :9], axis=1)
print(f'{workload} x {batch}, step_greedy_kernel, s_memtime ticks per wave: p50 / p90 / p99 / max')
for i, n in enumerate(names):
    print(f'  {n:14s}', np.percentile(d[:, i], [50, 90, 99, 100]).round(0))
for i, n in zip(range(9, 14), ['observe', 'zoom', 'actions', 'communicate', 'choose']):
    print(f'    agents/{n:12s}', np.percentile(t[:, i], [50, 90, 99, 100]).round(0))
life = t[:, 8] - t[:, 0]
print('  wave life     ', np.percentile(life, [50, 90, 99, 100]).round(0))
print('  s_memtime ticks per microsecond: %.0f' % (life.sum() / (t[:, 15].sum() / 100.0)))
