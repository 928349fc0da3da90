#!/usr/bin/env python3
"""How much the per-step kernels at a large batch depend on WHERE an engine's occlusion records lie: several engines of one build alive in
one process, the learner-versus-greedy step and step_random timed on each (dispatch events).  MATE_LUT_BLOCKS=1 (an experiment build
switch): the records from mate_engine_block_alloc's shuffled 2 MiB chunks instead of hipMalloc.
python tools/lut_place_probe.py [batch] [engines] [workload]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mate_amd.config import read_config  # noqa: E402
from mate_amd.engine import Engine  # noqa: E402
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
count = int(sys.argv[2]) if len(sys.argv) > 2 else 5
cfg = read_config(sys.argv[3] if len(sys.argv) > 3 else 'MATE-4v8-9.yaml')
mine = torch.zeros((batch, 4, 2), device='cuda')
engines = []
for k in range(count):
    e = Engine(cfg, batch, seed=0)
    e.enable_policies()
    e.reset()
    engines.append(e)
for rnd in range(2):
    out = []
    for k, e in enumerate(engines):
        for _ in range(40):
            e.step_versus_greedy('camera', mine, auto_reset=64)
        torch.cuda.synchronize(); e.kernel_time(enable=1)
        for _ in range(200):
            e.step_versus_greedy('camera', mine, auto_reset=64)
        torch.cuda.synchronize(); ms, _ = e.kernel_time(enable=False)
        out.append(ms * 1e3)
    print(f'MATE_LUT_BLOCKS={os.environ.get("MATE_LUT_BLOCKS", "0")} batch {batch} round {rnd}: step_greedy_kernel us per engine: ' + ' '.join(f'{v:6.2f}' for v in out), flush=True)
