#!/usr/bin/env python3
"""What a 20-step timed region costs beyond its kernel: synchronise, one 20-step rollout launch, synchronise -- plain, with the
dispatch events of Engine.kernel_time armed, and with an event record + a side-stream copy around it (the bench's gather).
    python tools/region_probe.py"""
import os, statistics, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mate_amd.config import read_config  # noqa: E402
from mate_amd.engine import Engine  # noqa: E402
eng = Engine(read_config('MATE-4v8-9.yaml'), 4096, seed=0)
eng.reset()
eng.reserve_rollout(20)
for _ in range(200):
    eng.rollout_random(20, auto_reset=6)
torch.cuda.synchronize()
side = torch.cuda.Stream()
ev = torch.cuda.Event()
slot = torch.zeros(5, dtype=torch.float64, device='cuda')


def region(kind):
    if kind == 'premark':
        ev.record()              # (the marker of what the gather may read: recorded behind the previous region's last launch)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if kind == 'gather':
        ev.record()
    eng.rollout_random(20, auto_reset=6)
    if kind in ('gather', 'premark'):
        with torch.cuda.stream(side):
            side.wait_event(ev)
            slot.copy_(eng.episode_stats, non_blocking=True)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e6


for kind, timing in (('plain', 0), ('plain', 1), ('gather', 1), ('premark', 1), ('gather', 1), ('premark', 1), ('plain', 0)):
    eng.kernel_time(enable=timing)
    ts = sorted(region(kind) for _ in range(300))
    avg, n = eng.kernel_time(enable=0)
    print('%-6s dispatch events %s: region median %.1f us p10 %.1f' % (kind, 'armed' if timing else 'off  ', ts[150], ts[30]), '(kernel %.1f us over %d launches)' % (avg * 1e3, n) if n else '')
