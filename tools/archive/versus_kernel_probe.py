#!/usr/bin/env python3
"""Per-launch kernel time of the learner-versus-greedy step (Engine.step_versus_greedy), fused one-launch form against the
two-launch form (MATE_POLICY_SPLIT=1), from the dispatch events.  python tools/versus_kernel_probe.py [workload] [batch]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mate_amd.config import read_config  # noqa: E402
from mate_amd.engine import Engine  # noqa: E402
workload = sys.argv[1] if len(sys.argv) > 1 else 'MATE-4v8-9.yaml'
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
eng = Engine(read_config(workload), batch, seed=0)
eng.enable_policies()
eng.reset()
mine = torch.zeros((batch, eng.num_cameras, 2), device='cuda')
for k in (8, 1):
    for _ in range(64):
        eng.step_versus_greedy('camera', mine, auto_reset=k)
    torch.cuda.synchronize()
    eng.kernel_time(enable=1)
    t0 = time.perf_counter()
    for _ in range(512):
        eng.step_versus_greedy('camera', mine, auto_reset=k)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ms, n = eng.kernel_time(enable=False)
    print(f'{workload} x {batch} auto_reset={k} split={os.environ.get("MATE_POLICY_SPLIT", "0")}: flow {eng.last_flow}, {dt / 512 * 1e6:.1f} us per step end to end, '
          f'timed kernel {ms * 1e3:.2f} us x {n} launches')
