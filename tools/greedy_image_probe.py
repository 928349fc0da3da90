#!/usr/bin/env python3
"""The fused Greedy rollout of MATE-4v8-9 on the row image (MATE_GREEDY_IMAGE=1) against the descriptor packer (=0): first the same bits
(Greedy vs Greedy, a camera learner and a target learner against the greedy opponents, across episode ends), then kernel time and
roofline fraction per flow and batch.  python tools/greedy_image_probe.py [batches]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mate_amd.config import read_config  # noqa: E402
from mate_amd.engine import Engine  # noqa: E402
from bench import algorithmic_bytes  # noqa: E402


def make(cfg, n, image, seed=3):
    os.environ['MATE_GREEDY_IMAGE'] = '1' if image else '0'
    try:
        eng = Engine(cfg, n, seed=seed)
    finally:
        os.environ.pop('MATE_GREEDY_IMAGE', None)
    eng.enable_policies()
    eng.reset()
    return eng


def same(a, b):
    return torch.equal(a.contiguous().view(torch.uint8), b.contiguous().view(torch.uint8))


cfg = read_config('MATE-4v8-9.yaml', max_episode_steps=19)
n = 150
a, b = make(cfg, n, True), make(cfg, n, False)
gen = torch.Generator(device='cuda').manual_seed(1)
ok = True
for it in range(6):
    cam = (torch.rand((n, 4, 2), device='cuda', generator=gen) * 2 - 1) * 6
    tgt = (torch.rand((n, 8, 2), device='cuda', generator=gen) * 2 - 1) * 25
    for flow in ('greedy', 'camera', 'target'):
        out = []
        for e in (a, b):
            if flow == 'greedy':
                rows = e.rollout_greedy(7, auto_reset=True, want_masks=True)
            else:
                rows = e.rollout_versus_greedy(flow, cam if flow == 'camera' else tgt, 5, auto_reset=2, want_masks=True)
            out.append([r.clone() for r in rows] + [e._rollout['masks'][:rows[2].shape[0]].clone(), e.export_state().clone(), e.policy_actions()[0].clone(), e.policy_actions()[1].clone()])
        ok &= all(same(x, y) for x, y in zip(*out))
print('row image == descriptor packer, bit for bit:', ok, '; episodes finished:', float(a.episode_stats[0]), flush=True)
assert ok
full = read_config('MATE-4v8-9.yaml')
for batch in [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else '8192,16384,65536').split(',')]:
    for flow, K in (('camera', 5), ('greedy', 32)):
        line = f'{flow:7s} K={K:2d} N={batch:6d}'
        for image in (False, True):
            e = make(full, batch, image, seed=0)
            e.reserve_rollout(K, search='none')
            mine = (torch.rand((batch, 4, 2), device='cuda') * 2 - 1) * 2.5
            launch = (lambda: e.rollout_versus_greedy('camera', mine, K, auto_reset=4)) if flow == 'camera' else (lambda: e.rollout_greedy(K, auto_reset=2))
            for _ in range(8):
                launch()
            torch.cuda.synchronize()
            e.kernel_time(enable=1)
            idle0, t0, k = e.idle_steps(), time.perf_counter(), 0
            while time.perf_counter() - t0 < 0.5:
                for _ in range(4):
                    launch()
                k += 4
                torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            ex = batch * K * k - (e.idle_steps() - idle0)
            km, _ = e.kernel_time(enable=False)
            bb = algorithmic_bytes(4, 8, 9)
            line += f"   {'image' if image else 'packer'} {km * 1e3:9.2f} us frac {bb * batch * K / (km * 1e-3) / 8e12:.3f} e2e {bb * ex / dt / 8e12:.3f}"
            e.close(); del e; torch.cuda.empty_cache()
        print(line, flush=True)
