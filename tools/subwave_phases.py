#!/usr/bin/env python3
"""Per-phase cycles per step of the fused rollout of a learner's team against the greedy opponents, one environment per wave against
four (needs the -DMATE_PHASE_CLOCKS build: MATE_ENGINE_LIB=mate_amd/lib/libmate_engine_prof.so python tools/subwave_phases.py
[workload] [batch] [frames] [team])."""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mate_amd.config import read_config  # noqa: E402
from mate_amd.engine import Engine  # noqa: E402
workload = sys.argv[1] if len(sys.argv) > 1 else 'MATE-2v4-0.yaml'
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
R = int(sys.argv[3]) if len(sys.argv) > 3 else 10
team = sys.argv[4] if len(sys.argv) > 4 else 'target'
names = ['draws', 'cameras', 'targets', 'view', 'assign', 'scratch', 'pack', 'loop', 'observe', 'zoom', 'actions', 'communicate', 'choose']
for on in (False, True):
    eng = Engine(read_config(workload), batch, seed=0)
    per = eng.set_sub_wave(on)
    eng.enable_policies()
    eng.reset()
    k = eng.num_targets if team == 'target' else eng.num_cameras
    mine = (torch.rand((batch, k, 2), device='cuda') * 2 - 1) * (10.0 if team == 'target' else 2.5)
    launch = (lambda: eng.rollout_versus_greedy(team, mine, R, auto_reset=4)) if team in ('target', 'camera') else (lambda: eng.rollout_greedy(R, auto_reset=2))
    for _ in range(6):
        launch()
    buf = torch.zeros((batch, 16), dtype=torch.int64, device='cuda')
    eng.lib.mate_engine_debug_phase_clocks.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    eng.lib.mate_engine_debug_phase_clocks(eng._h, ctypes.c_void_p(buf.data_ptr()))
    launch()
    torch.cuda.synchronize()
    raw = buf.cpu().numpy().astype(np.float64)
    t = raw[:, :13] / R
    print(f'{workload} batch {batch} R {R} team {team}, {per} environment(s) per wave: cycles per step per wave, mean / p50 / p99 over environments')
    for i, n in enumerate(names):
        print(f'  {n:11s} {t[:, i].mean():8.0f} {np.percentile(t[:, i], 50):8.0f} {np.percentile(t[:, i], 99):8.0f}')
    print(f'  total    {t.sum(axis=1).mean():8.0f}   wave lifetime {raw[:, 14].mean():10.0f} ticks')
    print('  s_memtime ticks per microsecond in this launch: %.0f' % (raw[:, 14].sum() / (raw[:, 15].sum() / 100.0)))
    eng.close()
    del eng
