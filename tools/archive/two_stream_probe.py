"""Experiment: one 4096-environment rollout launch vs two 2048-environment engines on two streams."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mate_amd.config import read_config
from mate_amd.engine import Engine
cfg = read_config('MATE-4v8-9.yaml')
R, reps = 32, 64
one = Engine(cfg, 4096, seed=0)
one.reset()
for _ in range(4): one.rollout_random(R)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(reps): one.rollout_random(R)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f'one engine x 4096: {4096 * R * reps / dt / 1e6:.1f} M env-steps/s')
del one
for parts in (2, 4):
    n = 4096 // parts
    engs = [Engine(cfg, n, seed=0, first_env_index=i * n) for i in range(parts)]
    streams = [torch.cuda.Stream() for _ in range(parts)]
    for e, s in zip(engs, streams):
        with torch.cuda.stream(s):
            e.reset()
            for _ in range(4): e.rollout_random(R)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        for e, s in zip(engs, streams):
            with torch.cuda.stream(s):
                e.rollout_random(R)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f'{parts} engines x {n} on {parts} streams: {4096 * R * reps / dt / 1e6:.1f} M env-steps/s')
    del engs

print('one launch per step:')
one = Engine(cfg, 4096, seed=0)
one.reset()
for _ in range(100): one.step_random(auto_reset=True)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(2000): one.step_random(auto_reset=True)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f'one engine x 4096: {4096 * 2000 / dt / 1e6:.1f} M env-steps/s')
del one
for parts in (2,):
    n = 4096 // parts
    engs = [Engine(cfg, n, seed=0, first_env_index=i * n) for i in range(parts)]
    streams = [torch.cuda.Stream() for _ in range(parts)]
    for e, s in zip(engs, streams):
        with torch.cuda.stream(s):
            e.reset()
            for _ in range(100): e.step_random(auto_reset=True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(2000):
        for e, s in zip(engs, streams):
            with torch.cuda.stream(s):
                e.step_random(auto_reset=True)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f'{parts} engines x {n} on {parts} streams: {4096 * 2000 / dt / 1e6:.1f} M env-steps/s')
