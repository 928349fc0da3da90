#!/usr/bin/env python3
"""Time of a full reset() (placement, occlusion tables, first view) of a batch: python tools/reset_probe.py [workload] [batch]."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mate_amd.config import read_config  # noqa: E402
from mate_amd.engine import Engine  # noqa: E402
workload = sys.argv[1] if len(sys.argv) > 1 else 'MATE-8v8-9.yaml'
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
eng = Engine(read_config(workload), batch, seed=0)
eng.reset()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
times = []
for _ in range(5):
    a.record(); eng.reset(); b.record(); torch.cuda.synchronize()
    times.append(a.elapsed_time(b))
print(os.environ.get('MATE_ENGINE_LIB', 'default').split('/')[-1], workload, batch, 'full reset ms', ' '.join('%.2f' % t for t in times))
