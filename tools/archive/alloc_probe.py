#!/usr/bin/env python3
"""Does the speed of the fused rollout depend on WHERE its output buffers lie?  Several engines of the same build in one
process (each with its own 6.5 GB of observation blocks), launches interleaved and timed one by one; then the same with
the observation blocks of the engines swapped.   python tools/alloc_probe.py [n_engines] [R]"""
import os, statistics, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mate_amd.config import read_config  # noqa: E402
from mate_amd.engine import Engine  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
R = int(sys.argv[2]) if len(sys.argv) > 2 else 256
pre = float(os.environ.get('PRE_GB', '0'))
if pre:
    dummy = torch.empty(int(pre * 2**30), dtype=torch.uint8, device='cuda')
engines = []
for i in range(n):
    eng = Engine(read_config('MATE-4v8-9.yaml'), 4096, seed=0)
    eng.reset()
    engines.append(eng)
for eng in engines:
    for _ in range(2):
        out = eng.rollout_random(R, auto_reset=True)
    print('engine', len([e for e in engines if e is eng]), 'camera block at 0x%x' % out[0].data_ptr() if isinstance(out, (tuple, list)) else type(out))
torch.cuda.synchronize()
times = [[] for _ in engines]
for _ in range(10):
    for i, eng in enumerate(engines):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); eng.rollout_random(R, auto_reset=True); e1.record(); e1.synchronize()
        times[i].append(e0.elapsed_time(e1) * 1e3)
for i, t in enumerate(times):
    print('engine created #%d: median %.1f us  min %.1f' % (i, statistics.median(t), min(t)))

# ---- does the speed follow the output buffers or the engine?  swap the blocks of engine 0 and the last engine
a, b = engines[0], engines[-1]
a._rollout, b._rollout = b._rollout, a._rollout
for e in (a, b):
    e._rollout.pop('_calls', None)
times = [[] for _ in engines]
for _ in range(10):
    for i, eng in enumerate(engines):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); eng.rollout_random(R, auto_reset=True); e1.record(); e1.synchronize()
        times[i].append(e0.elapsed_time(e1) * 1e3)
print('after swapping the output blocks of the first and the last engine:')
for i, t in enumerate(times):
    r = engines[i]._rollout
    print('engine created #%d: median %.1f us  min %.1f   cam 0x%x tgt 0x%x scal 0x%x' % (i, statistics.median(t), min(t), r['camera_obs'].data_ptr(), r['target_obs'].data_ptr(), r['scalars'].data_ptr()))
