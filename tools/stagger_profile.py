#!/usr/bin/env python3
"""When do waves reach each phase, relative to the first wave's start?  (prof build:
MATE_ENGINE_LIB=mate_amd/lib/libmate_engine_prof.so MATE_STAGGER=k python tools/stagger_profile.py)"""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mate_amd.config import read_config  # noqa: E402
from mate_amd.engine import Engine  # noqa: E402

workload = sys.argv[1] if len(sys.argv) > 1 else 'MATE-4v8-9.yaml'
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
eng = Engine(read_config(workload), batch, seed=0)
eng.reset()
for _ in range(50):
    eng.step_random(auto_reset=True)
buf = torch.zeros((batch, 16), dtype=torch.int64, device='cuda')
eng.lib.mate_engine_debug_phase_clocks.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
eng.lib.mate_engine_debug_phase_clocks(eng._h, ctypes.c_void_p(buf.data_ptr()))
names = ['start', 'load', 'sim', 'view', 'assign', 'scratch', 'pack', 'store', 'end']
for rep in range(3):
    eng.step_random(auto_reset=True)
    torch.cuda.synchronize()
t = buf.cpu().numpy().astype(np.float64)
# s_memtime counters are per XCD: normalise every wave to the first start of its own XCD (clusters of the time base)
order = np.argsort(t[:, 0])
base = np.zeros(batch)
cl = np.zeros(batch, dtype=np.int64)
start = 0
sorted_t = t[order, 0]
k = 0
for i in range(1, batch + 1):
    if i == batch or sorted_t[i] - sorted_t[i - 1] > 1e6:
        base[order[start:i]] = sorted_t[start]
        cl[order[start:i]] = k
        k += 1
        start = i
print('clusters (XCDs):', k, np.bincount(cl))
rel = (t[:, :9] - base[:, None]) / 1000.0
print('stagger', os.environ.get('MATE_STAGGER', '0'), ': kcycles after the first wave start on the same XCD: p0 p25 p50 p75 p100')
for i, n in enumerate(names):
    print(f'  {n:8s}', np.percentile(rel[:, i], [0, 25, 50, 75, 100]).round(2))
hw = t[:, 15].astype(np.int64)
slot = hw & 15
print('  wave slots used', np.bincount(slot))
for kk in range(4):
    sel = (slot & 3) == kk
    if sel.any():
        print(f'  slot&3={kk}: n={sel.sum()} start p50 {np.median(rel[sel, 0]):.2f} pack-start p50 {np.median(rel[sel, 6]):.2f} end p50 {np.median(rel[sel, 8]):.2f} end max {rel[sel, 8].max():.2f}')
np.savez(f'gpurun_out/stagger_{os.environ.get("MATE_STAGGER", "0")}.npz', t=t)
