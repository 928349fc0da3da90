"""Long parity soak (not collected by pytest; run by hand on an MI355X):

    python tests/soak_vs_oracle.py MATE-4v8-9.yaml 4096 400 [steps per fused launch, default 1]

Native reset + Philox random-policy rollout of N environments for S steps, the CPU oracle stepping the same
streams beside the GPU.  Counts environments whose masks / integer state ever differ from the oracle's and the
largest position error; a tangent-ray coin flip of the reference would show up here as a
diverging environment, which is why this is a census and not an assertion."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import gpu_util as U  # noqa: E402
from mate_amd.config import read_config  # noqa: E402
from mate_amd.engine import Engine  # noqa: E402
from oracle import oracle as O  # noqa: E402

workload = sys.argv[1] if len(sys.argv) > 1 else 'MATE-4v8-9.yaml'
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 400
fused = int(sys.argv[4]) if len(sys.argv) > 4 else 1      # > 1: the fused rollout kernel, `fused` steps per launch
threads = min(64, len(os.sched_getaffinity(0)))
cfg = read_config(workload)
eng = Engine(cfg, n, seed=99, first_env_index=12345, obs_dtype=torch.float32)
eng.reset()
torch.cuda.synchronize()
proto = U.oracle_proto_from_config(cfg, O)
batch = O.OracleBatch(proto, n, seed=99, first_env_index=12345)
batch.reset(threads=threads)
sd = eng.state_dict()
for k in U.STATE_KEYS:
    ref = batch.gather(k)
    assert np.array_equal(sd[k].reshape(ref.shape), ref), ('reset', k)
for e in range(n):                       # identical occlusion tables on both sides (the GPU's)
    for c in range(eng.num_cameras):
        batch.env(e).set_lut(c, *eng.lut_read(e, c))
MASKS = ['camera_target_view_mask', 'target_camera_view_mask', 'target_obstacle_view_mask', 'target_target_view_mask', 'camera_camera_view_mask']
INTS = ['tgt_colliding', 'tgt_goals', 'freights', 'bounties', 'remaining_cargoes', 'awaiting_cargo_counts', 'num_delivered_cargoes', 'episode_step']
bad = np.zeros(n, dtype=bool)
first_bad = {}                           # environment -> (step, what differed first)
worst = 0.0
t0 = time.time()
rows = None
for s in range(steps):
    if fused > 1:
        if s % fused == 0:
            rows = eng.rollout_random(min(fused, steps - s), auto_reset=False, want_masks=True)
        mask_words = eng._rollout['masks'][s % fused]
    else:
        eng.step_random(auto_reset=False, want_masks=True)
        mask_words = None
    batch.step(auto_reset=False, threads=threads)
    masks = eng.unpack_masks(mask_words)
    for m in MASKS:
        ref = batch.gather(m) != 0
        now = (masks[m].reshape(ref.shape) != ref).reshape(n, -1).any(axis=1)
        for e in np.nonzero(now & ~bad)[0]:
            first_bad.setdefault(int(e), (s, m))
        bad |= now
    if (fused > 1 and (s % fused == fused - 1 or s == steps - 1)) or (fused == 1 and (s % 10 == 9 or s == steps - 1)):    # state is current at launch ends
        sdg = eng.state_dict()
        for k in INTS:
            ref = batch.gather(k)
            now = (sdg[k].reshape(ref.shape) != ref).reshape(n, -1).any(axis=1)
            for e in np.nonzero(now & ~bad)[0]:
                first_bad.setdefault(int(e), (s, k))
            bad |= now
        good = ~bad
        worst = max(worst, float(np.abs(sdg['tgt_x'] - batch.gather('tgt_x'))[good].max()), float(np.abs(sdg['tgt_y'] - batch.gather('tgt_y'))[good].max()))
oc, ot = batch.observe()
to = (rows[1][(steps - 1) % fused] if fused > 1 else eng.target_obs).cpu().numpy()
obs_err = float(np.abs(to - ot)[~bad].max())
last_scalars = rows[2][(steps - 1) % fused] if fused > 1 else eng.scalars
rew_equal = bool(np.array_equal(last_scalars[:, 1].cpu().numpy()[~bad], batch.gather('reward_tgt').astype(np.float32)[~bad]))
print(f'{workload} ({"fused " + str(fused) + "-step launches" if fused > 1 else "one launch per step"}): {n} envs x {steps} steps = {n * steps} env-steps in {time.time() - t0:.0f} s; environments that ever diverged: {int(bad.sum())}; '
      f'max |position error| on the rest {worst:.2e}; final target-obs error {obs_err:.2e}; rewards equal: {rew_equal}')
if first_bad:
    print('  first differences (environment: step, field):', '; '.join(f'{e}: {st}, {what}' for e, (st, what) in sorted(first_bad.items())))
