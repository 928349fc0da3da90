#!/usr/bin/env python3
"""Bare A/B of engine builds on the fused rollout, outside bench.py: one engine per build in ONE process, the builds' launches
interleaved (A B A B ...), every launch timed with its own event pair; no statistics gather, no second stream.

    python tools/rollout_ab.py libA.so libB.so [workload] [batch] [R] [rounds]
prints per build: median / min launch time, and the same after a fresh reset (episodes young: fewer pairs in range)."""
import ctypes, os, statistics, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mate_amd.config import read_config  # noqa: E402
import mate_amd._native as native  # noqa: E402
import mate_amd.engine as engine_mod  # noqa: E402

libs = [os.path.abspath(p) for p in sys.argv[1:] if p.endswith('.so')]      # two or more builds
rest = [p for p in sys.argv[1:] if not p.endswith('.so')]
workload = rest[0] if len(rest) > 0 else 'MATE-4v8-9.yaml'
batch = int(rest[1]) if len(rest) > 1 else 4096
R = int(rest[2]) if len(rest) > 2 else 256
rounds = int(rest[3]) if len(rest) > 3 else 12
# where the observation blocks lie decides 5-25 % of a launch (tools/store_roof.hip): BOTH builds write the SAME blocks here, and
# BALLAST_GB (default 64) of device memory is taken first -- the blocks then lie in the part of the memory where every
# allocation measured fast
ballast = torch.empty(int(float(os.environ.get('BALLAST_GB', '64')) * 2**30), dtype=torch.uint8, device='cuda')
engines = []
for lib in libs:
    native.lib, native.LIB_PATH = None, lib      # every Engine keeps the handle it was created with
    eng = engine_mod.Engine(read_config(workload), batch, seed=0)
    eng.reset()
    engines.append(eng)
engines[0].reserve_rollout(R)
for eng in engines[1:]:
    eng._rollout = engines[0]._rollout
for eng in engines:
    for _ in range(2):
        eng.rollout_random(R, auto_reset=True)
torch.cuda.synchronize()
times = [[] for _ in engines]
for _ in range(rounds):
    for i, eng in enumerate(engines):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); eng.rollout_random(R, auto_reset=True); e1.record(); e1.synchronize()
        times[i].append(e0.elapsed_time(e1) * 1e3)
for lib, t in zip(libs, times):
    print(os.path.basename(lib), 'us per %d-step call (launch + resets): median %.1f min %.1f' % (R, statistics.median(t), min(t)), ' %.3g env-steps/s' % (batch * R / statistics.median(t) * 1e6))
