#!/usr/bin/env python3
"""FrameSkip(K) over MultiCamera(GreedyTargetAgent) -- one policy kernel + one K-frame launch per learner action -- with the restarts
(a) by flag on the caller's stream behind every m-th launch, direct launches; (b) the same from a HIP graph (Stepper(frame_skip=K));
(c) pipelined: on the engine's side stream behind every m-th launch, under the next m launches (auto_reset = ('pipelined', m)).
python tools/frameskip_pipelined_probe.py [workload] [batches] [K] [frames per restart interval]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mate_amd.config import read_config  # noqa: E402
from mate_amd.engine import Engine  # noqa: E402

workload = sys.argv[1] if len(sys.argv) > 1 else 'MATE-4v8-9.yaml'
batches = [int(b) for b in (sys.argv[2] if len(sys.argv) > 2 else '4096,16384').split(',')]
K = int(sys.argv[3]) if len(sys.argv) > 3 else 5
frames = int(sys.argv[4]) if len(sys.argv) > 4 else 64
cfg = read_config(workload)
B_ALG = 7504
m = max(1, frames // K)

for batch in batches:
    launches = (1024 if batch <= 16384 else 256) // K // m * m
    for form in ('flagged', 'graph', 'pipelined'):
        eng = Engine(cfg, batch, seed=0)
        eng.enable_policies()
        eng.reset()
        eng.reserve_rollout(K, search='none')
        mine = (torch.rand((batch, eng.num_cameras, 2), device='cuda') * 2 - 1) * torch.tensor([5.0, 2.5], device='cuda')
        st = None
        if form == 'graph':
            st = eng.make_stepper(mine, None, auto_reset=m, graph_steps=m, between=lambda: mine.mul_(-1.0), versus='camera', frame_skip=K)
            run = st.run
        else:
            mode = m if form == 'flagged' else ('pipelined', m)

            def run(n):
                for _ in range(n):
                    mine.mul_(-1.0)
                    eng.rollout_versus_greedy('camera', mine, K, auto_reset=mode)
        run(4 * m)
        torch.cuda.synchronize()
        best = None
        for _ in range(3):
            i0, t0 = eng.idle_steps(), time.perf_counter()
            run(launches)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            ex = batch * launches * K - (eng.idle_steps() - i0)
            if best is None or dt < best[0]:
                best = (dt, ex)
        dt, ex = best
        print(f'{workload} x {batch} FrameSkip({K}) restarts per {m} launches, {form:9s}: {dt / launches * 1e6:7.2f} us per launch, {ex / dt:.3g} executed env-steps/s, '
              f'end_to_end_frac {B_ALG * ex / dt / 8e12:.3f}, idle share {1 - ex / (batch * launches * K):.3f}', flush=True)
        if st is not None:
            st.close()
        del st, eng
        torch.cuda.empty_cache()
