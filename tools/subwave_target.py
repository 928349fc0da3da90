#!/usr/bin/env python3
"""A profiling target: `launches` fused launches of a small scenario with one or four environments per wave (mate_engine_set_sub_wave).
python tools/subwave_target.py <flow: target10 | random64> <scenario> <batch> <launches> <0 | 1: environments per wave forced to one / the shape's number>
(tools/pmc_collect.py cases `sub_*`; target10 = the target trainers' flow: MultiTarget(GreedyCameraAgent) + FrameSkip(10))"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mate_amd.config import read_config  # noqa: E402
from mate_amd.engine import Engine  # noqa: E402
flow, scenario, batch, launches, on = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), bool(int(sys.argv[5]))
eng = Engine(read_config(scenario), batch, seed=0)
per_wave = eng.set_sub_wave(on)
eng.enable_policies()
eng.reset()
if flow == 'target10':
    eng.reserve_rollout(10, search='none')
    mine = (torch.rand((batch, eng.num_targets, 2), device='cuda') * 2 - 1) * 10.0
    for _ in range(launches):
        mine.mul_(-1.0)
        eng.rollout_versus_greedy('target', mine, 10, auto_reset=4)
else:
    eng.reserve_rollout(64, search='none')
    for _ in range(launches):
        eng.rollout_random(64, auto_reset=2)
torch.cuda.synchronize()
print('done', flow, scenario, batch, 'environments per wave:', per_wave)
