import sys, time, torch
sys.path.insert(0, '/root/repo')
from mate_amd.config import read_config
from mate_amd.engine import Engine
for n in (4096, 8192):
    eng = Engine(read_config('MATE-4v8-9.yaml'), n, seed=0)
    eng.reset()
    for R in (8, 32, 64, 128):
        for _ in range(5): eng.rollout_random(R, auto_reset=True)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        reps = max(8, 2048 // R)
        for _ in range(reps): eng.rollout_random(R, auto_reset=True)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f'batch {n} rollout R={R}: {dt / (reps * R) * 1e6:.2f} us/step  {n * reps * R / dt / 1e6:.1f} M env-steps/s')
    del eng
