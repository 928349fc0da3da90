#!/usr/bin/env python3
"""What the small tier of the occlusion-table launches should hold (MATE_LUT_SMALL_CAP): a masked reset of ~8 % of the batch (what a restart
group of the Greedy-vs-Greedy flow restarts) and a whole-batch reset, timed per call; and how many tables the small tier deferred.
python tools/reset_cap_probe.py [workload] [batch]     (run once per MATE_LUT_SMALL_CAP value)"""
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
gen = torch.Generator(device='cuda'); gen.manual_seed(1)
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
part, full = [], []
for _ in range(6):
    mask = (torch.rand(batch, device='cuda', generator=gen) < 0.08).to(torch.uint8)
    torch.cuda.synchronize()
    a.record(); eng.reset(mask); b.record(); torch.cuda.synchronize()
    part.append(a.elapsed_time(b) * 1e3)
for _ in range(4):
    a.record(); eng.reset(); b.record(); torch.cuda.synchronize()
    full.append(a.elapsed_time(b) * 1e3)
print(f'MATE_LUT_SMALL_CAP={os.environ.get("MATE_LUT_SMALL_CAP", "default")} {workload} x {batch}: masked reset of ~8 % of the batch {sorted(part)[len(part) // 2]:.0f} us (median of 6: '
      + ' '.join('%.0f' % t for t in part) + f'); whole-batch reset {sorted(full)[len(full) // 2]:.0f} us', flush=True)
