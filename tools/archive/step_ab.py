#!/usr/bin/env python3
"""A/B of two builds of the library on the per-step kernels, interleaved in one process: python tools/step_ab.py libA.so libB.so [batch] [workload]
(step_random and the learner-versus-greedy step; kernel time from the dispatch events, 400 launches per round, 3 rounds each)."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mate_amd import _native  # noqa: E402
from mate_amd.config import read_config  # noqa: E402
from mate_amd.engine import Engine  # noqa: E402

libs = sys.argv[1:3]
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
cfg = read_config(sys.argv[4] if len(sys.argv) > 4 else 'MATE-4v8-9.yaml')
engines = []
for path in libs:
    _native.lib, _native.LIB_PATH = None, os.path.abspath(path)      # every Engine keeps the handle it was created with
    a = Engine(cfg, batch, seed=0); a.reset()
    b = Engine(cfg, batch, seed=0); b.enable_policies(); b.reset()
    engines.append((path, a, b))
mine = torch.zeros((batch, 4, 2), device='cuda')
for rnd in range(3):
    for path, a, b in engines:
        for _ in range(50):
            a.step_random(auto_reset=32)
        torch.cuda.synchronize(); a.kernel_time(enable=1)
        for _ in range(400):
            a.step_random(auto_reset=32)
        torch.cuda.synchronize(); ms_a, _ = a.kernel_time(enable=False)
        for _ in range(50):
            b.step_versus_greedy('camera', mine, auto_reset=64)
        torch.cuda.synchronize(); b.kernel_time(enable=1)
        for _ in range(400):
            b.step_versus_greedy('camera', mine, auto_reset=64)
        torch.cuda.synchronize(); ms_b, _ = b.kernel_time(enable=False)
        print(f'round {rnd} {os.path.basename(path):32s} step_kernel {ms_a * 1e3:6.2f} us   step_greedy_kernel {ms_b * 1e3:6.2f} us', flush=True)
