import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mate_amd.config import read_config
from mate_amd.engine import Engine
R = int(sys.argv[1]) if len(sys.argv) > 1 else 50
eng = Engine(read_config('MATE-4v8-9.yaml'), 4096, seed=0)
eng.reset()
eng.rollout_random(R)
torch.cuda.synchronize()
for chunk in range(4):
    n = 40
    eng.kernel_time(enable=1)
    t0 = time.perf_counter()
    for _ in range(n):
        eng.rollout_random(R)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    k_ms, launches = eng.kernel_time(enable=0)
    print(f'R={R}: {dt / (n * R) * 1e6:.2f} us/step  {4096 * n * R / dt / 1e6:.1f} M env-steps/s; rollout kernel {k_ms * 1e3 / R:.2f} us/step', flush=True)
