#!/usr/bin/env python3
"""BASELINE config 3 (MATE-8v8-9 x 8192, Greedy vs Greedy): executed env-steps/s of the fused rollouts with batched restarts
(auto_reset = 2, bench.py's default so far) and with pipelined restarts (resets on the side stream under the next launch), by
launch length.  python tools/c3_pipelined_probe.py [seconds per point]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mate_amd.config import read_config  # noqa: E402
from mate_amd.engine import Engine  # noqa: E402
seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
batch = 8192
for mode, R in ((2, 48), ('pipelined', 48), ('pipelined', 32), ('pipelined', 24), (2, 48), ('pipelined', 48)):
    eng = Engine(read_config('MATE-8v8-9.yaml'), batch, seed=0)
    eng.enable_policies()
    eng.reset()
    eng.reserve_rollout(R)
    for _ in range(int(1400 / R)):                     # into the steady state: episodes last ~1.2 k steps
        eng.rollout_greedy(R, auto_reset=mode)
    torch.cuda.synchronize()
    eng.kernel_time(enable=1)
    idle0, t0, n = eng.idle_steps(), time.perf_counter(), 0
    while time.perf_counter() - t0 < seconds:
        for _ in range(8):
            eng.rollout_greedy(R, auto_reset=mode)
        n += 8
        torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    executed = batch * R * n - (eng.idle_steps() - idle0)
    ms, timed = eng.kernel_time(enable=False)
    print(f'auto_reset={mode!s:10s} R={R:3d}: {executed / dt:.4g} executed env-steps/s, idle slots {100 * (1 - executed / (batch * R * n)):.1f} %, '
          f'rollout_greedy_kernel {ms * 1e3:.0f} us x {timed}, {dt / n * 1e3:.3f} ms per launch end to end', flush=True)
    eng.close()
    del eng
