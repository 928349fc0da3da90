"""Reset latency / throughput: full-batch reset and the batched auto-reset of the Greedy workload."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mate_amd.config import read_config
from mate_amd.engine import Engine
for wl, n in (('MATE-4v8-9.yaml', 4096), ('MATE-8v8-9.yaml', 8192)):
    eng = Engine(read_config(wl), n, seed=0)
    eng.reset(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        eng.reset()
    torch.cuda.synchronize()
    print(wl, n, 'full reset %.2f ms' % ((time.perf_counter() - t0) / 5 * 1e3), 'monolithic' if os.environ.get('MATE_RESET_MONOLITHIC') else 'split')
