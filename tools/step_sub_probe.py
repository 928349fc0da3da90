#!/usr/bin/env python3
"""The per-step Greedy flows of the small scenarios (step_versus_greedy / step_greedy, replayed from HIP graphs) on the one-step form of the
sub-wave rollout kernel against step_greedy_kernel: same bits first, then microseconds per step.  python tools/step_sub_probe.py"""
import os, sys, time, torch
sys.path.insert(0, '/root/repo')
from mate_amd.config import read_config
from mate_amd.engine import Engine
sys.path.insert(0, '/root/repo'); from bench import algorithmic_bytes
def same(a, b): return torch.equal(a.contiguous().view(torch.uint8), b.contiguous().view(torch.uint8))
def make(name, n, sub, seed=0, **kw):
    os.environ['MATE_STEP_SUBWAVE'] = '1' if sub else '0'      # (1 is the default since it measured faster)
    if sub: os.environ['MATE_SUBWAVE'] = '1'
    try: e = Engine(read_config(name + '.yaml', **kw), n, seed=seed)
    finally:
        os.environ.pop('MATE_STEP_SUBWAVE', None); os.environ.pop('MATE_SUBWAVE', None)
    e.enable_policies(); e.reset(); return e
# identity: per-step flows, across episode ends, batched restarts
for name, team in (('MATE-2v4-0', 'target'), ('MATE-4v2-9', 'camera'), ('MATE-2v2-9', 'target')):
    a, b = make(name, 70, True, seed=4, max_episode_steps=9), make(name, 70, False, seed=4, max_episode_steps=9)
    gen = torch.Generator(device='cuda').manual_seed(2)
    k = a.num_targets if team == 'target' else a.num_cameras
    ok = True
    for it in range(30):
        act = (torch.rand((70, k, 2), device='cuda', generator=gen) * 2 - 1) * (25 if team == 'target' else 6)
        out = []
        for e in (a, b):
            if it % 3 == 2: e.step_greedy(auto_reset=4 if it >= 15 else True)
            else: e.step_versus_greedy(team, act, auto_reset=4 if it >= 15 else True)
            out.append([e.camera_obs.clone(), e.target_obs.clone(), e.scalars.clone(), e.export_state().clone()])
        ok &= all(same(x, y) for x, y in zip(*out))
    print(name, 'per-step flows on the sub-wave kernel identical:', ok, 'flows', a.last_flow, b.last_flow, 'episodes', float(a.episode_stats[0]), flush=True)
for name, team in (('MATE-2v4-0', 'target'), ('MATE-4v2-9', 'camera')):
    for n in (8192, 16384, 65536):
        line = f'{name} versus {team} N={n}'
        for sub in (False, True):
            e = make(name, n, sub)
            k = e.num_targets if team == 'target' else e.num_cameras
            mine = (torch.rand((n, k, 2), device='cuda') * 2 - 1) * 5
            st = e.make_stepper(mine if team == 'camera' else None, mine if team == 'target' else None, auto_reset=32, graph_steps=64, between=lambda m=mine: m.mul_(-1.0), versus=team)
            st.run(128); torch.cuda.synchronize()
            i0, t0 = e.idle_steps(), time.perf_counter(); st.run(512); torch.cuda.synchronize(); dt = time.perf_counter() - t0
            ex = n * 512 - (e.idle_steps() - i0)
            bb = algorithmic_bytes(e.num_cameras, e.num_targets, e.num_obstacles)
            line += f"   {'sub-wave' if sub else 'step_greedy'} {dt / 512 * 1e6:8.2f} us/step e2e {bb * ex / dt / 8e12:.3f}"
            st.close(); e.close(); del st, e; torch.cuda.empty_cache()
        print(line, flush=True)
# ... and step(actions): the joint actions of both teams in caller-owned buffers that a stand-in policy kernel rewrites before every step
for name in ('MATE-2v4-0', 'MATE-4v2-9', 'MATE-1v1-9'):
    for n in (8192, 16384, 65536):
        line = f'{name} step(actions) N={n}'
        for sub in (False, True):
            e = make(name, n, sub)
            flat = (torch.rand(n * (e.num_cameras + e.num_targets) * 2, device='cuda') * 2 - 1) * 5
            cam, tgt = flat[:n * e.num_cameras * 2].view(n, e.num_cameras, 2), flat[n * e.num_cameras * 2:].view(n, e.num_targets, 2)
            st = e.make_stepper(cam, tgt, auto_reset=32, graph_steps=64, between=lambda f=flat: f.mul_(-1.0))
            st.run(128); torch.cuda.synchronize()
            i0, t0 = e.idle_steps(), time.perf_counter(); st.run(512); torch.cuda.synchronize(); dt = time.perf_counter() - t0
            ex = n * 512 - (e.idle_steps() - i0)
            bb = algorithmic_bytes(e.num_cameras, e.num_targets, e.num_obstacles)
            line += f"   {'sub-wave' if sub else 'step_kernel'} {dt / 512 * 1e6:8.2f} us/step e2e {bb * ex / dt / 8e12:.3f}"
            st.close(); e.close(); del st, e; torch.cuda.empty_cache()
        print(line, flush=True)
