#!/usr/bin/env python3
"""The learner-facing per-step flows at several batch sizes: step(actions) replayed from a HIP graph (`external_actions`),
learner versus the on-device greedy opponents (`versus_greedy`), and the same batch stepped as TWO half-batch engines on two
streams (a learner that alternates between two groups of environments: one group's step runs under the other's policy).
python tools/learner_flows_probe.py [workload] [batches comma-separated] [reset_interval]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mate_amd.config import read_config  # noqa: E402
from mate_amd.engine import Engine  # noqa: E402

workload = sys.argv[1] if len(sys.argv) > 1 else 'MATE-4v8-9.yaml'
batches = [int(b) for b in (sys.argv[2] if len(sys.argv) > 2 else '4096,16384,65536').split(',')]
interval = int(sys.argv[3]) if len(sys.argv) > 3 else 32
flows = (sys.argv[4] if len(sys.argv) > 4 else 'external,versus,groups').split(',')
GRAPH = 64
cfg = read_config(workload)
B_ALG = {'MATE-4v8-9.yaml': 7504}.get(workload, 7504)


def timed(run, steps, n_envs, idle):
    run(GRAPH * 2)
    torch.cuda.synchronize()
    best = None
    for _ in range(3):
        i0 = idle()
        t0 = time.perf_counter()
        run(steps)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        ex = n_envs * steps - (idle() - i0)
        if best is None or dt < best[0]:
            best = (dt, ex)
    dt, ex = best
    return dt / steps * 1e6, ex / dt, B_ALG * ex / dt / 8e12


for batch in batches:
    steps = 1024 if batch <= 16384 else 256
    if 'external' in flows:      # (1) step(actions), one engine
        eng = Engine(cfg, batch, seed=0)
        eng.reset()
        flat = torch.rand(batch * 12 * 2, device='cuda') * 2 - 1
        cam, tgt = flat[:batch * 8].view(batch, 4, 2), flat[batch * 8:].view(batch, 8, 2)
        st = eng.make_stepper(cam, tgt, auto_reset=interval, graph_steps=GRAPH, between=lambda: flat.mul_(-1.0))
        us, rate, frac = timed(st.run, steps, batch, eng.idle_steps)
        print(f'{workload} x {batch} external_actions (graph {GRAPH}, reset/{interval}): {us:.2f} us/step, {rate:.3g} env-steps/s, end_to_end_frac {frac:.3f}', flush=True)
        st.close()
        del st, eng
        torch.cuda.empty_cache()
    if 'versus' in flows:        # (2) learner versus greedy
        eng = Engine(cfg, batch, seed=0)
        eng.enable_policies()
        eng.reset()
        mine = torch.zeros((batch, 4, 2), device='cuda')
        st = eng.make_stepper(mine, None, auto_reset=interval, graph_steps=GRAPH, between=lambda: mine.mul_(-1.0).add_(0.5), versus='camera')
        us, rate, frac = timed(st.run, steps, batch, eng.idle_steps)
        print(f'{workload} x {batch} versus_greedy (graph {GRAPH}, reset/{interval}, flow {eng.last_flow}): {us:.2f} us/step, {rate:.3g} executed env-steps/s, end_to_end_frac {frac:.3f}', flush=True)
        st.close()
        # the kernel itself, from the dispatch events (direct launches)
        for _ in range(8):
            eng.step_versus_greedy('camera', mine, auto_reset=interval)
        torch.cuda.synchronize()
        eng.kernel_time(enable=1)
        for _ in range(256):
            eng.step_versus_greedy('camera', mine, auto_reset=interval)
        torch.cuda.synchronize()
        ms, nl = eng.kernel_time(enable=False)
        print(f'    kernel of the one-launch form: {ms * 1e3:.2f} us x {nl} launches', flush=True)
        del st, eng
        torch.cuda.empty_cache()
    if 'groups' in flows:        # (3) two half-batch engines on two streams
        half = batch // 2
        streams = [torch.cuda.Stream(), torch.cuda.Stream()]
        engs, steppers, bufs = [], [], []
        for gi in range(2):
            with torch.cuda.stream(streams[gi]):
                e = Engine(cfg, half, seed=0, first_env_index=gi * half)
                e.reset()
                f = torch.rand(half * 12 * 2, device='cuda') * 2 - 1
                c, t = f[:half * 8].view(half, 4, 2), f[half * 8:].view(half, 8, 2)
                s = e.make_stepper(c, t, auto_reset=interval, graph_steps=GRAPH, between=(lambda f=f: f.mul_(-1.0)))
                engs.append(e); steppers.append(s); bufs.append(f)
        torch.cuda.synchronize()

        def run_two(n):
            for _ in range(n // GRAPH):
                for gi in range(2):
                    with torch.cuda.stream(streams[gi]):
                        steppers[gi].run(GRAPH)

        us, rate, frac = timed(run_two, steps, batch, lambda: engs[0].idle_steps() + engs[1].idle_steps())
        print(f'{workload} x {batch} external_actions as 2 groups of {half} on two streams: {us:.2f} us per step of the whole batch, {rate:.3g} env-steps/s, end_to_end_frac {frac:.3f}', flush=True)
        for s in steppers:
            s.close()
        del steppers, engs, bufs
        torch.cuda.empty_cache()
