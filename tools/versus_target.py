#!/usr/bin/env python3
"""A profiling target: `launches` learner-versus-greedy steps (Engine.step_versus_greedy, the learner plays the cameras) of `batch`
environments of MATE-4v8-9, direct launches.  python tools/versus_target.py [batch] [launches]   (tools/pmc_collect.py cases `versus*`)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mate_amd.config import read_config  # noqa: E402
from mate_amd.engine import Engine  # noqa: E402
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
launches = int(sys.argv[2]) if len(sys.argv) > 2 else 300
eng = Engine(read_config('MATE-4v8-9.yaml'), batch, seed=0)
eng.enable_policies()
eng.reset()
mine = (torch.rand((batch, 4, 2), device='cuda') * 2 - 1) * torch.tensor([5.0, 2.5], device='cuda')
for _ in range(launches):
    mine.mul_(-1.0)
    eng.step_versus_greedy('camera', mine, auto_reset=64)
torch.cuda.synchronize()
print('done', eng.last_flow)
