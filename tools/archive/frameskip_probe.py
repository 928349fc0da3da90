#!/usr/bin/env python3
"""FrameSkip(K) over MultiCamera(GreedyTargetAgent) -- what every example trainer's make_env builds (examples/ippo/camera/config.py:
frame_skip = 5) -- as ONE launch per learner action (Engine.rollout_versus_greedy) against K per-step launches:
python tools/frameskip_probe.py [K] [batches]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mate_amd.config import read_config  # noqa: E402
from mate_amd.engine import Engine  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 5
batches = [int(b) for b in (sys.argv[2] if len(sys.argv) > 2 else '4096,16384,65536').split(',')]
cfg = read_config('MATE-4v8-9.yaml')
for batch in batches:
    for resets in (13, 26):          # one restart of the finished environments per `resets` launches (65 / 130 frames)
        eng = Engine(cfg, batch, seed=0)
        eng.enable_policies()
        eng.reset()
        eng.reserve_rollout(K, search='none')
        mine = (torch.rand((batch, 4, 2), device='cuda') * 2 - 1) * torch.tensor([5.0, 2.5], device='cuda')
        n = 2 * resets * 8

        def run(launches):
            for _ in range(launches):
                mine.mul_(-1.0)                                   # the learner's stand-in: one kernel per ACTION (per K frames)
                eng.rollout_versus_greedy('camera', mine, K, auto_reset=resets)

        run(2 * resets)
        torch.cuda.synchronize()
        best = None
        for _ in range(3):
            i0, t0 = eng.idle_steps(), time.perf_counter()
            run(n)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            ex = batch * n * K - (eng.idle_steps() - i0)
            if best is None or dt < best[0]:
                best = (dt, ex)
        dt, ex = best
        eng.kernel_time(enable=1)
        run(resets)
        torch.cuda.synchronize()
        km, nl = eng.kernel_time(enable=False)
        print(f'MATE-4v8-9 x {batch}, FrameSkip({K}) versus greedy targets, restart per {resets} launches: {dt / n * 1e6:.1f} us per launch = {dt / n / K * 1e6:.2f} us per frame, '
              f'{ex / dt:.3g} executed env-steps/s, end_to_end_frac {7504 * ex / dt / 8e12:.3f}; rollout_greedy_kernel {km * 1e3:.1f} us per launch', flush=True)
        eng.close()
        del eng
        torch.cuda.empty_cache()
