#!/usr/bin/env python3
"""BASELINE config 3 (MATE-8v8-9 x 8192, the reference's Greedy agents on both sides, fused 48-step launches, restarts after every
second launch) as ONE engine and as G groups of environments on G streams (mate_amd.engine.EngineGroups): does the tail of one
group's launch fill with the other group's waves?
python tools/c3_groups_probe.py [workload] [batch] [R] [groups comma-separated]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mate_amd.config import read_config  # noqa: E402
from mate_amd.engine import EngineGroups  # noqa: E402

workload = sys.argv[1] if len(sys.argv) > 1 else 'MATE-8v8-9.yaml'
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
R = int(sys.argv[3]) if len(sys.argv) > 3 else 48
group_counts = [int(g) for g in (sys.argv[4] if len(sys.argv) > 4 else '1,2,3,4').split(',')]
cfg = read_config(workload)
B_ALG = {'MATE-8v8-9.yaml': 11568, 'MATE-4v8-9.yaml': 7504}.get(workload, 7504)
LAUNCHES = 24

for G in group_counts:
    n = batch // G * G
    eg = EngineGroups(cfg, n, groups=G, seed=0, policies=True)
    eg.reset()
    eg.each(lambda g, e: e.reserve_rollout(R, search='none'))
    body = lambda g, e: e.rollout_greedy(R, auto_reset=2)  # noqa: E731
    if G > 1:
        eg.pick_streams(body)
    for _ in range(40):                      # into the steady state: episodes of ~1.2 k steps
        eg.each(body)
    torch.cuda.synchronize()
    best = None
    for _ in range(3):
        i0 = eg.idle_steps()
        t0 = time.perf_counter()
        for _ in range(LAUNCHES):
            eg.each(body)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        ex = n * R * LAUNCHES - (eg.idle_steps() - i0)
        if best is None or dt < best[0]:
            best = (dt, ex)
    dt, ex = best
    print(f'{workload} x {n} greedy-vs-greedy as {G} group(s): {dt / LAUNCHES * 1e3:.3f} ms per {R}-step launch of the whole batch, '
          f'{ex / dt:.3g} executed env-steps/s, end_to_end_frac {B_ALG * ex / dt / 8e12:.3f}', flush=True)
    eg.close()
    del eg
    torch.cuda.empty_cache()
