"""Host-side cost of one Engine.step_random call (two kernel launches through ctypes) vs the GPU time per step."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mate_amd.config import read_config
from mate_amd.engine import Engine
for n in (64, 4096):
    eng = Engine(read_config('MATE-4v8-9.yaml'), n, seed=0)
    eng.reset()
    for _ in range(200):
        eng.step_random(auto_reset=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3000):
        eng.step_random(auto_reset=True)
    t_enqueue = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_total = time.perf_counter() - t0
    print(f'batch {n}: enqueue {t_enqueue / 3000 * 1e6:.2f} us/step on the host, {t_total / 3000 * 1e6:.2f} us/step end to end')
