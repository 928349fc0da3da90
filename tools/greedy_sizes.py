import os, sys, time
sys.path.insert(0, '/root/repo')
import torch
from mate_amd.config import read_config
from mate_amd.engine import Engine
for n in (4096, 4100, 5120, 6144, 8192):
    eng = Engine(read_config('MATE-8v8-9.yaml'), n, seed=0)
    eng.enable_policies(); eng.reset()
    R, reps = 32, 8
    for _ in range(2): eng.rollout_greedy(R)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): eng.rollout_greedy(R)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f'batch {n}: {dt / reps * 1e6:.0f} us per 32-step launch, {dt / (reps * R) * 1e6:.1f} us/step')
    del eng
