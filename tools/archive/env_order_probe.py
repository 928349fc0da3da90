#!/usr/bin/env python3
"""Does a cost-aware slot -> environment map shorten a one-generation fused launch?  (Experiment of round 4, answered NO --
profiles/r04_env_order_probe.txt -- and the kernel side, Ptrs::env_order + mate_engine_env_order in commit history, was removed
again: this script documents the method and needs that patch to run.  It needed the profiling build:
MATE_ENGINE_LIB=$PWD/mate_amd/lib/libmate_engine_prof.so python3 tools/env_order_probe.py [workload] [batch] [R] [rounds])

A launch of 4096 environments lasts as long as its slowest wave, and how slow a wave is depends on its own environment (pairs inside
sectors, overflowing occlusion degrees: persistent over an episode), on the three waves it shares a SIMD with and on its age
among them (the oldest is served first).  The probe measures every wave's cycles per step in launch n (per-wave s_memtime stamps),
reads the hardware placement of the slots (which slots share a SIMD, in which age order), builds a map that (a) gives the heaviest
environments the oldest slots and (b) deals the environments over the SIMDs in snake order so that every SIMD's four sum alike,
installs it (mate_engine_env_order) and times launch n + 1 -- against the same launches with slot = environment."""
import ctypes, os, statistics, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mate_amd.config import read_config  # noqa: E402
from mate_amd.engine import Engine  # noqa: E402
workload = sys.argv[1] if len(sys.argv) > 1 else 'MATE-4v8-9.yaml'
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
R = int(sys.argv[3]) if len(sys.argv) > 3 else 20
rounds = int(sys.argv[4]) if len(sys.argv) > 4 else 24
eng = Engine(read_config(workload), batch, seed=0)
eng.reset()
lib = eng.lib
if not hasattr(lib, 'mate_engine_env_order'):
    raise SystemExit('tools/archive/env_order_probe.py: this tree does not export mate_engine_env_order -- the kernel side of the experiment was removed '
                     'when it measured no gain (profiles/HISTORY.md section 5, profiles/r04_env_order_probe.txt); the script is kept as the record of how it was run')
lib.mate_engine_debug_phase_clocks.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
lib.mate_engine_env_order.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
buf = torch.zeros((batch, 16), dtype=torch.int64, device='cuda')
lib.mate_engine_debug_phase_clocks(eng._h, ctypes.c_void_p(buf.data_ptr()))
for _ in range(200 // R + 3):
    eng.rollout_random(R, auto_reset=True)


def launch():
    """one launch; (microseconds by the dispatch events, per-ENVIRONMENT cycles per step, per-environment hardware id word)"""
    eng.kernel_time(enable=1)
    eng.rollout_random(R, auto_reset=True)
    torch.cuda.synchronize()
    ms, n = eng.kernel_time(enable=False)
    raw = buf.cpu().numpy()
    return ms * 1e3, raw[:, :8].sum(axis=1).astype(np.float64) / R, raw[:, 13].copy()


def placement(hw):
    hwid, xcc = hw & 0xffffffff, (hw >> 32) & 0xf
    simd, cu, sh, se = (hwid >> 4) & 3, (hwid >> 8) & 15, (hwid >> 12) & 1, (hwid >> 13) & 7
    return ((((xcc * 8 + se) * 2 + sh) * 16 + cu) * 4 + simd), hwid & 0xf


order = torch.arange(batch, dtype=torch.int32, device='cuda')
identity = order.clone()
inv = np.arange(batch)                 # environment -> slot of the installed map
plain, mapped, spread_plain, spread_mapped = [], [], [], []
us, cyc, hw = launch()
for rnd in range(rounds):
    use_map = rnd % 2 == 1
    if use_map:
        # the previous launch ran with slot = environment: cyc / hw are indexed by environment = slot
        simd_of_slot, age_of_slot = placement(hw)
        slots_by_simd = {}
        for s in range(batch):
            slots_by_simd.setdefault(int(simd_of_slot[s]), []).append(s)
        groups = [sorted(v, key=lambda s: age_of_slot[s]) for v in slots_by_simd.values()]     # per SIMD: its slots, oldest first
        groups = [g for g in groups if len(g) == 4]
        # own weight of an environment: its cycles with the age effect of the slot it ran in taken out
        age_mean = {a: cyc[age_of_slot == a].mean() for a in set(age_of_slot.tolist())}
        own = cyc - np.array([age_mean[a] for a in age_of_slot.tolist()]) + cyc.mean()
        ranked = np.argsort(-own)          # heaviest first
        new = np.arange(batch)
        taken = set()
        if len(groups) * 4 == batch:
            n = len(groups)
            for k in range(4):             # age k of every SIMD: the k-th quarter of the ranking, dealt forwards / backwards alternately
                part = ranked[k * n:(k + 1) * n]
                if k % 2:
                    part = part[::-1]
                for gi, g in enumerate(groups):
                    new[g[k]] = part[gi]
            order.copy_(torch.from_numpy(new.astype(np.int32)))
            lib.mate_engine_env_order(eng._h, ctypes.c_void_p(order.data_ptr()))
        else:
            use_map = False
    else:
        lib.mate_engine_env_order(eng._h, None)
    us, cyc_now, hw_now = launch()
    (mapped if use_map else plain).append(us)
    (spread_mapped if use_map else spread_plain).append((np.percentile(cyc_now, 50), cyc_now.max()))
    if not use_map:
        cyc, hw = cyc_now, hw_now
lib.mate_engine_env_order(eng._h, None)
print(f'{workload} x {batch}, {R}-step launches (profiling build), alternating: slot = environment / cost-aware map from the launch before')
print('  plain : median %.1f us  min %.1f  (per-wave cycles per step p50 %.0f, max %.0f)' % (statistics.median(plain), min(plain), np.mean([s[0] for s in spread_plain]), np.mean([s[1] for s in spread_plain])))
if mapped:
    print('  mapped: median %.1f us  min %.1f  (per-wave cycles per step p50 %.0f, max %.0f)' % (statistics.median(mapped), min(mapped), np.mean([s[0] for s in spread_mapped]), np.mean([s[1] for s in spread_mapped])))
