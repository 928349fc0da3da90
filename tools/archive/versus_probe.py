#!/usr/bin/env python3
"""Learner versus greedy opponents (MultiCamera / MultiTarget on the device): steps per second of the per-step flow, launched
directly and replayed from a HIP graph.  python tools/versus_probe.py [workload] [batch] [team]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mate_amd.config import read_config  # noqa: E402
from mate_amd.engine import Engine  # noqa: E402
workload = sys.argv[1] if len(sys.argv) > 1 else 'MATE-4v8-9.yaml'
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
team = sys.argv[3] if len(sys.argv) > 3 else 'camera'
for graph_steps in (0, 64):
    eng = Engine(read_config(workload), batch, seed=0)
    eng.enable_policies()
    eng.reset()
    agents = eng.num_cameras if team == 'camera' else eng.num_targets
    mine = torch.zeros((batch, agents, 2), device='cuda')

    def policy():
        mine.mul_(-1.0).add_(0.5)            # the learner's stand-in: one elementwise kernel per step

    st = eng.make_stepper(mine if team == 'camera' else None, mine if team == 'target' else None, auto_reset=8, graph_steps=graph_steps,
                          between=policy, versus=team)
    st.run(512)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    st.run(2048)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f'{workload} x {batch}, learner plays the {team}s, greedy opponents on the device, ' + ('HIP graph of 64 steps' if graph_steps else 'direct launches') +
          f': {dt / 2048 * 1e6:.1f} us per step, {batch * 2048 / dt:.3g} env-steps/s')
    st.close()
    del st, eng
