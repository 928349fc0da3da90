import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mate_amd.config import read_config
from mate_amd.engine import Engine
eng = Engine(read_config('MATE-4v8-9.yaml'), 4096, seed=0)
eng.reset()
torch.cuda.synchronize()
for chunk in range(12):
    n = 2000
    t0 = time.perf_counter()
    for _ in range(n):
        eng.step_random(auto_reset=True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f'chunk {chunk}: {dt / n * 1e6:.2f} us/step  {4096 * n / dt / 1e6:.1f} M env-steps/s', flush=True)
os.system('rocm-smi --showclocks 2>/dev/null | head -20')
