"""Long identity soak (not collected by pytest; run by hand on an MI355X): four environments per wave against one, every fused flow,
thousands of steps with episodes ending and restarting all the way -- every row, scalar and mask of every launch and the state at the end
must be the same bits.

    python tests/soak_subwave.py [environments, default 4096] [steps per flow, default 3000]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mate_amd.config import read_config  # noqa: E402
from mate_amd.engine import Engine  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3000


def same(a, b):
    return torch.equal(a.contiguous().view(torch.uint8), b.contiguous().view(torch.uint8))


for name in ('MATE-2v4-0', 'MATE-4v2-9', 'MATE-2v4-9', 'MATE-4v4-9', 'MATE-1v1-9', 'MATE-2v2-0', 'MATE-4v8-0'):
    cfg = read_config(name + '.yaml', max_episode_steps=700)
    engines = []
    for on in (True, False):
        eng = Engine(cfg, n, seed=17, first_env_index=1000)
        eng.set_sub_wave(on)
        eng.enable_policies()
        eng.reset()
        engines.append(eng)
    t0, bad, launches = time.time(), 0, 0
    gen = torch.Generator(device='cuda').manual_seed(5)
    for flow, K in (('random', 50), ('greedy', 25), ('target', 10), ('camera', 5)):
        if flow == 'random' and name == 'MATE-4v8-0':
            continue                      # (two per wave is a Greedy-flow kernel there)
        for it in range(steps // K):
            k = engines[0].num_targets if flow == 'target' else engines[0].num_cameras
            act = (torch.rand((n, k, 2), device='cuda', generator=gen) * 2 - 1) * (25 if flow == 'target' else 6)
            out = []
            for eng in engines:
                if flow == 'random':
                    rows = eng.rollout_random(K, auto_reset=2, want_masks=True)
                elif flow == 'greedy':
                    rows = eng.rollout_greedy(K, auto_reset=('pipelined' if it % 7 == 3 else 2), want_masks=True)
                else:
                    rows = eng.rollout_versus_greedy(flow, act, K, auto_reset=3, want_masks=True)
                out.append(list(rows) + [eng._rollout['masks'][:K]])
            torch.cuda.synchronize()
            bad += 0 if all(same(x, y) for x, y in zip(*out)) else 1
            launches += 1
    for eng in engines:
        eng.rollout_greedy(1, auto_reset=True)
    final = same(engines[0].export_state(), engines[1].export_state()) and same(engines[0].episode_stats, engines[1].episode_stats)
    print(f'{name}: {n} envs, {launches} launches over four flows, {float(engines[0].episode_stats[0]):.0f} episodes finished; launches that differed: {bad}; '
          f'final state and statistics identical: {final} ({time.time() - t0:.0f} s)', flush=True)
    del engines
    torch.cuda.empty_cache()
