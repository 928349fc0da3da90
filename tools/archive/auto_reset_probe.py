import sys, time, torch
sys.path.insert(0, '.')
from mate_amd.config import read_config
from mate_amd.engine import Engine
eng = Engine(read_config('MATE-4v8-9.yaml'), 4096, seed=0)
eng.reset()
for ar in (1, 8, 32, 0):
    for _ in range(200): eng.step_random(auto_reset=ar)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(1000): eng.step_random(auto_reset=ar)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    print('auto_reset', ar, 'flow', eng.last_flow, round(best * 1e3, 2), 'us/step')
