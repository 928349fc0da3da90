#!/usr/bin/env python3
"""A/B of two builds on the fused Greedy-vs-Greedy rollout (BASELINE config 3), interleaved in one process:
python tools/c3_ab.py libA.so libB.so [workload] [batch] [R]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mate_amd import _native  # noqa: E402
from mate_amd.config import read_config  # noqa: E402
from mate_amd.engine import Engine  # noqa: E402
libs = sys.argv[1:3]
workload = sys.argv[3] if len(sys.argv) > 3 else 'MATE-8v8-9.yaml'
batch = int(sys.argv[4]) if len(sys.argv) > 4 else 8192
R = int(sys.argv[5]) if len(sys.argv) > 5 else 48
cfg = read_config(workload)
engines = []
for path in libs:
    _native.lib, _native.LIB_PATH = None, os.path.abspath(path)
    e = Engine(cfg, batch, seed=0)
    e.enable_policies()
    e.reset()
    engines.append((path, e))
engines[0][1].reserve_rollout(R, search='none')
for _, e in engines[1:]:
    e._rollout = engines[0][1]._rollout
for _, e in engines:
    for _ in range(30):                      # into the steady state: episodes of ~1.2 k steps
        e.rollout_greedy(R, auto_reset=2)
torch.cuda.synchronize()
for rnd in range(4):
    for path, e in engines:
        e.kernel_time(enable=1)
        for _ in range(12):
            e.rollout_greedy(R, auto_reset=2)
        torch.cuda.synchronize()
        ms, n = e.kernel_time(enable=False)
        print(f'round {rnd} {os.path.basename(path):28s} rollout_greedy_kernel {ms * 1e3:8.1f} us per {R}-step launch x {n}', flush=True)
