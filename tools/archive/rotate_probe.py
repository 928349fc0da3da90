#!/usr/bin/env python3
"""Wave-priority rotation variants of the fused rollout on the SAME observation blocks (MATE_ROLLOUT_ROTATE is read at engine
creation): launch times interleaved.   python tools/rotate_probe.py 1 13 14 15 0"""
import os, statistics, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mate_amd.config import read_config  # noqa: E402
from mate_amd.engine import Engine  # noqa: E402
modes = sys.argv[1:] or ['1', '13', '14', '0']
R = 256
engines = []
for m in modes:
    os.environ['MATE_ROLLOUT_ROTATE'] = m
    eng = Engine(read_config('MATE-4v8-9.yaml'), 4096, seed=0)
    eng.reset()
    engines.append(eng)
engines[0].reserve_rollout(R)
for eng in engines[1:]:
    eng._rollout = engines[0]._rollout
for eng in engines:
    eng.rollout_random(R, auto_reset=True); eng.rollout_random(R, auto_reset=True)
torch.cuda.synchronize()
times = [[] for _ in engines]
for _ in range(10):
    for i, eng in enumerate(engines):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); eng.rollout_random(R, auto_reset=True); e1.record(); e1.synchronize()
        times[i].append(e0.elapsed_time(e1) * 1e3)
for m, t in zip(modes, times):
    print('MATE_ROLLOUT_ROTATE=%s: median %.1f us  min %.1f' % (m, statistics.median(t), min(t)))
