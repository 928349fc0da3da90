#!/usr/bin/env python3
"""What two environments per wave could buy AT MOST (round 4 review, item 3): a kernel that packs the environments of a batch of N into
N / 2 waves -- every vector instruction serving both, at no extra instruction -- would run like today's kernel on N / 2 environments.
So T(N) / T(N / 2) of today's fused rollout bounds the gain of a perfect packing at N environments from above (the real one pays for
every wave-uniform scalar that becomes a per-half vector value).  python tools/packing_bound_probe.py [steps per launch]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mate_amd.config import read_config  # noqa: E402
from mate_amd.engine import Engine  # noqa: E402

R = int(sys.argv[1]) if len(sys.argv) > 1 else 64
B_ALG = {'MATE-4v2-9.yaml': 3166, 'MATE-2v4-0.yaml': None, 'MATE-4v8-9.yaml': 7504, 'MATE-4v4-9.yaml': None}


def b_alg(Nc, Nt, No):
    Dc = 13 + 9 + 5 * Nt + 4 * No + 7 * Nc
    Dt = 13 + 14 + 7 * Nc + 4 * No + 5 * Nt
    return 4 * (Nc * Dc + Nt * Dt) + 8 * (Nc + Nt) + 2 * (16 * Nc + 35 * Nt + 72) + (24 * Nc + 24 * No + Nt) + 48


for workload in ('MATE-4v2-9.yaml', 'MATE-2v4-0.yaml', 'MATE-4v4-9.yaml', 'MATE-4v8-9.yaml'):
    cfg = read_config(workload)
    times = {}
    for batch in (1024, 2048, 4096, 8192, 16384):
        eng = Engine(cfg, batch, seed=0)
        eng.reset()
        eng.reserve_rollout(R, search='none')
        for _ in range(4):
            eng.rollout_random(R, auto_reset=4)
        torch.cuda.synchronize()
        eng.kernel_time(enable=1)
        for _ in range(16):
            eng.rollout_random(R, auto_reset=4)
        torch.cuda.synchronize()
        ms, n = eng.kernel_time(enable=False)
        times[batch] = ms * 1e3 / R
        bytes_ = b_alg(eng.num_cameras, eng.num_targets, eng.num_obstacles)
        print(f'{workload} x {batch:6d}: {times[batch]:7.3f} us per step ({R}-step launches), {batch / times[batch] * 1e6:.3g} env-steps/s, '
              f'{bytes_ * batch / times[batch] * 1e6 / 8e12:.3f} of the HBM peak'
              + (f'; T(N) / T(N/2) = {times[batch] / times[batch // 2]:.2f} (a perfect two-per-wave packing at {batch}: at most x{times[batch] / times[batch // 2]:.2f})' if batch // 2 in times else ''), flush=True)
        eng.close()
        del eng
        torch.cuda.empty_cache()
