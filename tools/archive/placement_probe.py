"""Is the placement of workgroups on CUs / SIMDs the same from launch to launch?  (profiling build)"""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mate_amd.config import read_config
from mate_amd.engine import Engine
eng = Engine(read_config('MATE-4v8-9.yaml'), 4096, seed=0)
eng.reset()
buf = torch.zeros((4096, 16), dtype=torch.int64, device='cuda')
eng.lib.mate_engine_debug_phase_clocks.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
eng.lib.mate_engine_debug_phase_clocks(eng._h, ctypes.c_void_p(buf.data_ptr()))
keys = []
for k in range(6):
    eng.rollout_random(32)
    torch.cuda.synchronize()
    hw = buf.cpu().numpy()[:, 13]
    hwid, xcc = hw & 0xffffffff, (hw >> 32) & 0xf
    simd, cu, sh, se = (hwid >> 4) & 3, (hwid >> 8) & 15, (hwid >> 12) & 1, (hwid >> 13) & 7
    keys.append((((xcc * 8 + se) * 2 + sh) * 16 + cu) * 4 + simd)
for k in range(1, 6):
    print('launch', k, 'waves on the same SIMD as in launch 0: %.3f' % (keys[k] == keys[0]).mean(), ' same CU: %.3f' % ((keys[k] >> 2) == (keys[0] >> 2)).mean())
w = np.arange(4096) & 3
print('wave index == SIMD index: %.3f' % ((keys[0] & 3) == w).mean())
blk = np.arange(4096) >> 2
cu0 = keys[0] >> 2
print('blocks per CU:', np.bincount(np.unique(np.stack([cu0, blk], 1), axis=0)[:, 0]).tolist()[:8], '...')
first = {}
for b, cu_ in zip(blk[::4], cu0[::4]):
    first.setdefault(int(cu_), []).append(int(b))
print('blocks sharing the first CUs:', list(first.items())[:4])
for k in (0, 1):
    simd_k = keys[k] & 3
    off = (simd_k - w) & 3
    per_block_const = (off.reshape(-1, 4).max(axis=1) == off.reshape(-1, 4).min(axis=1)).mean()
    ob = off.reshape(-1, 4)[:, 0]                       # per block
    # blocks b, b+256, b+512, b+768 share a CU
    per_cu = ob.reshape(4, 256)
    same_in_cu = (per_cu.max(axis=0) == per_cu.min(axis=0)).mean()
    print('launch', k, ': SIMD = (wave + offset) mod 4 with one offset per block: %.3f; the 4 blocks of a CU share the offset: %.3f; offsets histogram %s' % (
        per_block_const, same_in_cu, np.bincount(ob, minlength=4).tolist()))
for k in (0, 1, 2):
    simd_k = keys[k] & 3
    for b in (0, 1):
        rows = [simd_k[4 * (b + 256 * j):4 * (b + 256 * j) + 4].tolist() for j in range(4)]
        print('launch', k, 'CU of block', b, ': SIMD of waves 0-3 of its 4 blocks:', rows)
