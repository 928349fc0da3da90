"""Greedy vs Greedy (MATE-8v8-9): fused rollout launches against one policy + one step launch per step."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mate_amd.config import read_config
from mate_amd.engine import Engine
for n in (1024, 2048, 3072, 4096, 8192):
    eng = Engine(read_config('MATE-8v8-9.yaml'), n, seed=0)
    eng.enable_policies(); eng.reset()
    R, reps = 32, 16
    for _ in range(2): eng.rollout_greedy(R)
    torch.cuda.synchronize(); t0 = time.perf_counter(); i0 = eng.idle_steps()
    for _ in range(reps): eng.rollout_greedy(R)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    fused = (n * reps * R - (eng.idle_steps() - i0)) / dt / 1e6
    for _ in range(32): eng.step_greedy(auto_reset=32)
    torch.cuda.synchronize(); t0 = time.perf_counter(); i0 = eng.idle_steps()
    for _ in range(reps * R): eng.step_greedy(auto_reset=32)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    single = (n * reps * R - (eng.idle_steps() - i0)) / dt / 1e6
    print(f'batch {n}: fused rollout {fused:.1f} M executed env-steps/s, policy + step launches {single:.1f} M')
    del eng
