#!/usr/bin/env python3
"""Per-phase wave latency of the step kernel (needs the -DMATE_PHASE_CLOCKS build:
MATE_ENGINE_LIB=mate_amd/lib/libmate_engine_prof.so python tools/phase_profile.py)."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mate_amd.config import read_config  # noqa: E402
from mate_amd.engine import Engine  # noqa: E402

workload = sys.argv[1] if len(sys.argv) > 1 else 'MATE-4v8-9.yaml'
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
skip = int(sys.argv[3]) if len(sys.argv) > 3 else 0      # runtime phase mask (mate_engine_debug_skip; 255: launch + records + state store only)
eng = Engine(read_config(workload), batch, seed=0)
eng.lib.mate_engine_debug_skip.argtypes = [ctypes.c_void_p, ctypes.c_int32]
eng.reset()
for _ in range(50):
    eng.step_random(auto_reset=True)
buf = torch.zeros((batch, 16), dtype=torch.int64, device='cuda')
eng.lib.mate_engine_debug_phase_clocks.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
eng.lib.mate_engine_debug_phase_clocks(eng._h, ctypes.c_void_p(buf.data_ptr()))
eng.lib.mate_engine_debug_skip(eng._h, skip)
names = ['table+sync', 'load', 'simulate', 'view', 'assign', 'scratch', 'pack', 'store']
acc = np.zeros(8)
spans = []
reps = 20
for _ in range(reps):
    eng.step_random(auto_reset=True)
    torch.cuda.synchronize()
    t = buf.cpu().numpy().astype(np.float64)
    acc += np.diff(t[:, :9], axis=1).mean(axis=0)
    spans.append((t[:, 8].max() - t[:, 0].min(), (t[:, 8] - t[:, 0]).mean()))
    starts = t[:, 0] - t[:, 0].min()
    ends = t[:, 8].max() - t[:, 8]
acc /= reps
print('s_memtime ticks')
for n, v in zip(names, acc):
    print(f'  {n:12s} {v:9.1f}')
print('  per-wave total %.0f cycles; kernel span first-start..last-end %.0f cycles' % (np.mean([s[1] for s in spans]), np.mean([s[0] for s in spans])))
# every XCD counts its own clock: compare start / end times only inside one XCD (clusters of start stamps)
order = np.argsort(t[:, 0])
gaps = np.where(np.diff(t[order, 0]) > 1e6)[0]
groups = np.split(order, gaps + 1)
print('  %d clock domains (XCDs)' % len(groups))
for grp in groups:
    st, en = t[grp, 0] - t[grp, 0].min(), t[grp, 8] - t[grp, 0].min()
    print('   waves %4d  start p50/p90/max %s  end p50/p90/max %s' % (len(grp), np.percentile(st, [50, 90, 100]).round(0), np.percentile(en, [50, 90, 100]).round(0)))
kt = eng.kernel_time(0)



t = buf.cpu().numpy().astype(np.float64)
d = np.diff(t[:, :9], axis=1)
print('per-phase cycles: p50 / p90 / p99 / max')
for i, n in enumerate(names):
    print(f'  {n:12s}', np.percentile(d[:, i], [50, 90, 99, 100]).round(0))
life = t[:, 8] - t[:, 0]
print('  wave life   ', np.percentile(life, [50, 90, 99, 100]).round(0))
print('  s_memtime ticks per microsecond (against the 100 MHz s_memrealtime over every wave): %.0f' % (life.sum() / (t[:, 15].sum() / 100.0)))
slow = np.argsort(life)[-5:]
for e in slow:
    print('  slow env', e, d[e].round(0))

# inside simulate: [2]=start, 9=after draws, 12=after cameras, 10=after target prep, 11=after near screen, [3]=end
sub = np.stack([t[:, 9] - t[:, 2], t[:, 12] - t[:, 9], t[:, 10] - t[:, 12], t[:, 11] - t[:, 10], t[:, 3] - t[:, 11]], axis=1)
for i, n in enumerate(['draws', 'cameras', 'tgt prep', 'near screen', 'walk+finish']):
    print(f'  sim/{n:12s}', np.percentile(sub[:, i], [50, 90, 99, 100]).round(0))

sub = np.stack([t[:, 13] - t[:, 3], t[:, 14] - t[:, 13], t[:, 4] - t[:, 14]], axis=1)
for i, n in enumerate(['sector', 'range', 'tracked/inside']):
    print(f'  view/{n:12s}', np.percentile(sub[:, i], [50, 90, 99, 100]).round(0))
