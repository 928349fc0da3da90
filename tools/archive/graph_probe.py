"""Learner-in-the-loop stepping: host-counted direct launches against HIP-graph replay (run on the GPU box).

    python tools/graph_probe.py [batch] [steps]

step((camera_actions, target_actions)) with f32 joint actions in caller-owned device buffers, immediate auto-reset;
with and without a stand-in policy kernel rewriting the actions between steps."""
import sys
import time

import torch

sys.path.insert(0, '.')
from mate_amd.config import read_config  # noqa: E402
from mate_amd.engine import Engine  # noqa: E402

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
eng = Engine(read_config('MATE-4v8-9.yaml'), batch, seed=0)
eng.reset()
flat = torch.rand(batch * 12 * 2, device='cuda') * 2 - 1
cam = flat[:batch * 8].view(batch, 4, 2)
tgt = flat[batch * 8:].view(batch, 8, 2)
cam.mul_(torch.tensor([5.0, 2.5], device='cuda'))
tgt.mul_(20.0)


def policy():
    flat.mul_(-1.0)


def timed(fn, n):
    fn(64)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter()
        fn(n)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best / n * 1e6


def host_only(fn, n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn(n)
    dt = time.perf_counter() - t0
    torch.cuda.synchronize()
    return dt / n * 1e6


rows = []
for label, graph_steps, between in (('direct launches, no policy kernel', 0, None), ('direct launches + policy kernel', 0, policy),
                                    ('graph of 16 pairs, no policy kernel', 16, None), ('graph of 64 pairs, no policy kernel', 64, None),
                                    ('graph of 64 pairs + policy kernel', 64, policy), ('graph of 256 pairs + policy kernel', 256, policy)):
    st = eng.make_stepper(cam, tgt, auto_reset=True, graph_steps=graph_steps, between=between)
    us = timed(st.run, steps)
    host = host_only(st.run, steps)
    st.close()
    rows.append((label, us, host))
    print(f'{label:42s} {us:7.2f} us/step end to end  ({batch / us:7.1f} M env-steps/s)   host enqueue {host:6.2f} us/step', flush=True)
us = timed(lambda n: [eng.step_random(auto_reset=True) for _ in range(n)], steps)
print(f'{"step_random (on-device policy), direct":42s} {us:7.2f} us/step end to end  ({batch / us:7.1f} M env-steps/s)')
