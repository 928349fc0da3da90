"""Phase ablation of step_kernel (prof build): kernel time with one phase skipped."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mate_amd.config import read_config
from mate_amd.engine import Engine
eng = Engine(read_config('MATE-4v8-9.yaml'), 4096, seed=0)
eng.lib.mate_engine_debug_skip.argtypes = [ctypes.c_void_p, ctypes.c_int32]
eng.reset()
for _ in range(200):
    eng.step_random(auto_reset=False)
names = {0: 'nothing', 1: 'draws', 2: 'cameras', 4: 'targets', 8: 'view', 32: 'assign', 64: 'scratch', 128: 'pack', 255 - 128: 'all but pack', 255: 'everything'}
base = None
for mask, name in names.items():
    eng.lib.mate_engine_debug_skip(eng._h, mask)
    for _ in range(20):
        eng.step_random(auto_reset=False)
    eng.kernel_time(enable=1)
    for _ in range(300):
        eng.step_random(auto_reset=False)
    ms, n = eng.kernel_time(enable=0)
    base = base or ms
    print(f'skip {name:14s}: kernel {ms * 1e3:6.2f} us   delta {1e3 * (base - ms):6.2f} us')
