#!/usr/bin/env python3
"""Per-phase cycles per step of the rollout kernel (needs the -DMATE_PHASE_CLOCKS build:
MATE_ENGINE_LIB=mate_amd/lib/libmate_engine_prof.so python tools/rollout_phases.py [workload] [batch] [R])."""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mate_amd.config import read_config  # noqa: E402
from mate_amd.engine import Engine  # noqa: E402
workload = sys.argv[1] if len(sys.argv) > 1 else 'MATE-4v8-9.yaml'
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
R = int(sys.argv[3]) if len(sys.argv) > 3 else 32
eng = Engine(read_config(workload), batch, seed=0)
eng.reset()
for _ in range(3):
    eng.rollout_random(R)
buf = torch.zeros((batch, 16), dtype=torch.int64, device='cuda')
eng.lib.mate_engine_debug_phase_clocks.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
eng.lib.mate_engine_debug_phase_clocks(eng._h, ctypes.c_void_p(buf.data_ptr()))
eng.rollout_random(R)
torch.cuda.synchronize()
raw = buf.cpu().numpy().astype(np.float64)
t = raw[:, :8] / R
names = ['draws', 'cameras', 'targets', 'view', 'assign', 'scratch', 'pack', 'loop']
print(f'{workload} batch {batch} R {R}: cycles per step per wave, mean / p50 / p99 over waves')
for i, n in enumerate(names):
    print(f'  {n:8s} {t[:, i].mean():8.0f} {np.percentile(t[:, i], 50):8.0f} {np.percentile(t[:, i], 99):8.0f}')
print(f'  total    {t.sum(axis=1).mean():8.0f}')
print('  s_memtime ticks per microsecond in this launch: %.0f' % (raw[:, 14].sum() / (raw[:, 15].sum() / 100.0)))
tot = t.sum(axis=1)
print('  per-wave total per step: p50 %.0f p90 %.0f p99 %.0f max %.0f (the launch lasts as long as its slowest wave)' % tuple(np.percentile(tot, [50, 90, 99, 100])))
for e in np.argsort(tot)[-6:]:
    print('   slow env %5d' % e, t[e].round(0))
hw = buf.cpu().numpy()[:, 13]
hwid, xcc = hw & 0xffffffff, (hw >> 32) & 0xf
simd, cu, sh, se = (hwid >> 4) & 3, (hwid >> 8) & 15, (hwid >> 12) & 1, (hwid >> 13) & 7
key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
import collections
per_cu = collections.Counter(key.tolist())
per_simd = collections.Counter((key * 4 + simd).tolist())
print('  CUs used %d; waves per CU histogram %s' % (len(per_cu), sorted(collections.Counter(per_cu.values()).items())))
print('  SIMDs used %d; waves per SIMD histogram %s' % (len(per_simd), sorted(collections.Counter(per_simd.values()).items())))
load = np.array([per_simd[k] for k in (key * 4 + simd).tolist()])
for n in sorted(set(load.tolist())):
    print('   waves sharing a SIMD with %d resident waves: %5d  mean cycles per step %.0f' % (n, (load == n).sum(), tot[load == n].mean()))
cul = np.array([per_cu[k] for k in key.tolist()])
for n in sorted(set(cul.tolist())):
    print('   waves on a CU with %2d resident waves: %5d  mean cycles per step %.0f' % (n, (cul == n).sum(), tot[cul == n].mean()))
# what makes an environment slow?  features of its state at the end of the rollout
sd = eng.state_dict()
cx, cy, phi, th = sd['cam_x'], sd['cam_y'], sd['cam_phi'], sd['cam_theta']
tx, ty = sd['tgt_x'], sd['tgt_y']
cfg = read_config(workload)
cam = cfg.get('camera', {})
area = cam.get('min_viewing_angle', 90.0) * cam.get('max_sight_range', 500.0) ** 2
sight = np.sqrt(area / th)
dx, dy = tx[:, None, :] - cx[:, :, None], ty[:, None, :] - cy[:, :, None]
dist = np.hypot(dx, dy)
ang = np.degrees(np.arctan2(dy, dx))
rel = np.abs(phi[:, :, None] - ang); rel = np.minimum(rel, 360 - rel)
in_sector = (dist <= sight[:, :, None]) & (2 * rel <= th[:, :, None])
n_in = in_sector.sum(axis=(1, 2))
in_range = (dist <= sight[:, :, None]).sum(axis=(1, 2))
print('  pairs inside a camera sector (occlusion lookups) per env: mean %.2f max %d' % (n_in.mean(), n_in.max()))
for k in range(0, int(n_in.max()) + 1):
    sel = n_in == k
    if sel.sum() >= 5:
        print('   envs with %2d lookups: %5d  view %.0f  total %.0f' % (k, sel.sum(), t[sel, 3].mean(), tot[sel].mean()))
print('  corr(view, lookups) %.2f  corr(view, pairs in range) %.2f  corr(total, lookups) %.2f' % (np.corrcoef(t[:, 3], n_in)[0, 1], np.corrcoef(t[:, 3], in_range)[0, 1], np.corrcoef(tot, n_in)[0, 1]))
# collision candidates: targets within one step of an obstacle / camera circle
ox, oy, orad = sd['obs_x'], sd['obs_y'], sd['obs_radius']
d_to = np.hypot(tx[:, :, None] - ox[:, None, :], ty[:, :, None] - oy[:, None, :]) - orad[:, None, :]
d_tc = np.hypot(tx[:, :, None] - cx[:, None, :], ty[:, :, None] - cy[:, None, :]) - cam.get('radius', 40.0)
near = (d_to < 20.0).sum(axis=(1, 2)) + (d_tc < 20.0).sum(axis=(1, 2))
print('  corr(targets phase, near pairs) %.2f   corr(total, near pairs) %.2f' % (np.corrcoef(t[:, 2], near)[0, 1], np.corrcoef(tot, near)[0, 1]))
for k in range(0, 6):
    sel = near == k
    if sel.sum() >= 5:
        print('   envs with %d near pairs: %5d  targets %.0f  total %.0f' % (k, sel.sum(), t[sel, 2].mean(), tot[sel].mean()))
# do SIMD-mates share their fate?
sk = key * 4 + simd
order = np.argsort(sk, kind='stable')
groups = tot[order].reshape(-1, 4)
print('  per-SIMD mean of the 4 resident waves: p50 %.0f p90 %.0f max %.0f; within-SIMD spread (max-min) mean %.0f' % (
    np.percentile(groups.mean(axis=1), 50), np.percentile(groups.mean(axis=1), 90), groups.mean(axis=1).max(), (groups.max(axis=1) - groups.min(axis=1)).mean()))
ck = np.argsort(key, kind='stable')
cg = tot[ck].reshape(-1, 16)
print('  per-CU mean of the 16 resident waves: p50 %.0f p90 %.0f max %.0f min %.0f' % (np.percentile(cg.mean(axis=1), 50), np.percentile(cg.mean(axis=1), 90), cg.mean(axis=1).max(), cg.mean(axis=1).min()))
xk = np.argsort(xcc, kind='stable')
print('  per-XCD mean:', [round(tot[xcc == x].mean()) for x in sorted(set(xcc.tolist()))])
# positional effects: the wave's SIMD inside its CU, the CU inside its shader array, the workgroup's generation
print('  mean by SIMD id:', [round(tot[simd == i].mean()) for i in range(4)])
print('  mean by CU id (0-15):', [round(tot[cu == i].mean()) for i in range(16) if (cu == i).any()])
print('  mean by shader array / engine:', [round(tot[(se * 2 + sh) == i].mean()) for i in range(16) if ((se * 2 + sh) == i).any()])
gen = np.arange(batch) // 1024
print('  mean by dispatch generation (env // 1024):', [round(tot[gen == i].mean()) for i in range(int(gen.max()) + 1)])
wv = (buf.cpu().numpy()[:, 13] & 0xf)
print('  mean by hardware wave slot:', {int(i): round(tot[wv == i].mean()) for i in sorted(set(wv.tolist()))})
