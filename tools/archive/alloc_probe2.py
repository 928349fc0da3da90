#!/usr/bin/env python3
"""Store rate of a plain fill on each engine's observation blocks next to the time of the fused rollout that writes them."""
import os, statistics, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mate_amd.config import read_config  # noqa: E402
from mate_amd.engine import Engine  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
R = 256


def fill_rate(t):
    best = 1e9
    for rep in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); t.zero_(); e1.record(); e1.synchronize()
        if rep:
            best = min(best, e0.elapsed_time(e1))
    return t.numel() * t.element_size() / best / 1e6


engines = []
for i in range(n):
    eng = Engine(read_config('MATE-4v8-9.yaml'), 4096, seed=0)
    eng.reset()
    eng.rollout_random(R, auto_reset=True); eng.rollout_random(R, auto_reset=True)
    engines.append(eng)
torch.cuda.synchronize()
times = [[] for _ in engines]
for _ in range(8):
    for i, eng in enumerate(engines):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); eng.rollout_random(R, auto_reset=True); e1.record(); e1.synchronize()
        times[i].append(e0.elapsed_time(e1) * 1e3)
for i, eng in enumerate(engines):
    r = eng._rollout
    print('engine #%d: rollout median %.1f us | fill GB/s: camera block %.0f target block %.0f' % (i, statistics.median(times[i]), fill_rate(r['camera_obs']), fill_rate(r['target_obs'])),
          '| candidates probed [GB/s]:', [[round(x) for x in rr] for rr in getattr(eng, 'block_rates', [])])
