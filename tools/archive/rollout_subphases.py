#!/usr/bin/env python3
"""Where the SLOWEST waves of a fused launch spend their time (the launch lasts as long as they do).  Needs the sub-phase variant:
    python -m mate_amd.build --variant sub -DMATE_PHASE_CLOCKS -DMATE_SUB_CLOCKS
    MATE_ENGINE_LIB=$PWD/mate_amd/lib/libmate_engine_sub.so python3 tools/rollout_subphases.py [workload] [batch] [R]
Per wave: the eight phases of rollout_phases.py, the visibility phase split into sector geometry + fetch issue / range tests /
record wait + interpolation (+ overflow trips) / mask words / tracked bits, and per launch how many steps had an occlusion lookup and
an overflowing degree."""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mate_amd.config import read_config  # noqa: E402
from mate_amd.engine import Engine  # noqa: E402
workload = sys.argv[1] if len(sys.argv) > 1 else 'MATE-4v8-9.yaml'
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
R = int(sys.argv[3]) if len(sys.argv) > 3 else 256
eng = Engine(read_config(workload), batch, seed=0)
eng.reset()
for _ in range(3):
    eng.rollout_random(R)
buf = torch.zeros((batch, 32), dtype=torch.int64, device='cuda')
eng.lib.mate_engine_debug_phase_clocks.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
eng.lib.mate_engine_debug_phase_clocks(eng._h, ctypes.c_void_p(buf.data_ptr()))
eng.rollout_random(R)
torch.cuda.synchronize()
raw = buf.cpu().numpy().astype(np.float64)
t = raw[:, :8] / R
sub = raw[:, 16:24] / R
cnt = raw[:, 24:28] / R
tot = t.sum(axis=1)
names = ['draws', 'cameras', 'targets', 'view', 'assign', 'scratch', 'pack', 'loop']
subn = ['view: sector geometry + fetch issue', 'view: range tests', 'view: record wait + interpolation', 'view: mask words', 'view: tracked bits',
        'targets: step vector', 'targets: candidate circles', 'targets: clip + table']
cntn = ['steps with a lookup', 'steps with an overflowing degree', 'steps with a collision candidate', 'steps with a deflected target']
order = np.argsort(tot)
groups = [('all waves', order), ('fastest 10 %', order[:batch // 10]), ('middle 10 %', order[batch * 45 // 100: batch * 55 // 100]),
          ('slowest 10 %', order[-(batch // 10):]), ('slowest 1 %', order[-max(4, batch // 100):]), ('slowest 8', order[-8:])]
print(f'{workload} x {batch}, {R}-step launch (sub-phase variant): cycles per step per wave, means over groups of waves by total')
print('%-32s' % '' + ''.join('%14s' % g[0] for g in groups))
print('%-32s' % 'total' + ''.join('%14.0f' % tot[g[1]].mean() for g in groups))
for i, n in enumerate(names):
    print('%-32s' % n + ''.join('%14.0f' % t[g[1], i].mean() for g in groups))
for i, n in enumerate(subn):
    print('%-34s' % ('  ' + n)[:34] + ''.join('%14.0f' % sub[g[1], i].mean() for g in groups))
for i, n in enumerate(cntn):
    print('%-32s' % n[:32] + ''.join('%14.2f' % cnt[g[1], i].mean() for g in groups))
wv = (buf.cpu().numpy()[:, 13] & 0xf)
print('%-32s' % 'hardware wave slot (mean)' + ''.join('%14.2f' % wv[g[1]].mean() for g in groups))
