#!/usr/bin/env python3
"""What a single step_kernel launch is made of, in TIME (tools/ablate_pmc.py: the same in instruction counts): the profiling build's
runtime phase mask (mate_engine_debug_skip; bits 1 draws, 2 cameras, 4 targets, 8 view, 32 goals, 64 scratch, 128 pack + stores)
switches phases off one at a time and all together; every launch timed with its own dispatch events.

    MATE_ENGINE_LIB=$PWD/mate_amd/lib/libmate_engine_prof.so python3 tools/step_ablate.py [workload] [batch]

A kernel without a phase is not a simulation (auto_reset is off; the state drifts wherever it likes): the numbers say what the
launch costs when a phase's work is absent -- `everything` is launch + records + state store, `pack` is the step without its
observation stores."""
import ctypes, os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mate_amd.config import read_config
from mate_amd.engine import Engine
workload = sys.argv[1] if len(sys.argv) > 1 else 'MATE-4v8-9.yaml'
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
MASKS = [(0, 'nothing'), (1, 'draws'), (2, 'cameras'), (4, 'targets'), (8, 'view'), (32, 'goals'), (64, 'scratch'), (128, 'pack + stores'),
         (64 | 128, 'scratch + pack + stores'), (255 & ~128, 'everything but pack + stores'), (255, 'everything'), (0, 'nothing (again)')]
eng = Engine(read_config(workload), batch, seed=0)
eng.lib.mate_engine_debug_skip.argtypes = [ctypes.c_void_p, ctypes.c_int32]
eng.reset()
for _ in range(64):
    eng.step_random(auto_reset=False)
torch.cuda.synchronize()
print(f'{workload} x {batch}, step_random, one launch per step: median / min kernel time [us] over 200 launches with a phase switched off')
for mask, name in MASKS:
    eng.lib.mate_engine_debug_skip(eng._h, mask)
    for _ in range(8):
        eng.step_random(auto_reset=False)
    eng.kernel_time(enable=1)
    for _ in range(200):
        eng.step_random(auto_reset=False)
    torch.cuda.synchronize()
    ms, n = eng.kernel_time(enable=False)
    print(f'  without {name:30s} {ms * 1e3:7.2f} us average over {n} launches')
