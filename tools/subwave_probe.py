#!/usr/bin/env python3
"""Four environments per wave against one (mate_engine_set_sub_wave) on the fused rollouts of the small scenarios: kernel time from
the launch's dispatch events and the roofline fraction (SURVEY.md 8d bytes / kernel time / 8 TB/s), same engine state, same launches.

    python tools/subwave_probe.py [--batches 4096,16384,65536] [--cases target10,random,...]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from mate_amd.config import read_config  # noqa: E402
from mate_amd.engine import Engine  # noqa: E402
from bench import algorithmic_bytes  # noqa: E402

CASES = {
    # name: (scenario, flow, steps per launch)
    'target10': ('MATE-2v4-0.yaml', 'versus_target', 10),      # the target trainers: MultiTarget(GreedyCameraAgent) + FrameSkip(10)
    'camera5_2v4': ('MATE-2v4-0.yaml', 'versus_camera', 5),
    'camera5_4v8-9': ('MATE-4v8-9.yaml', 'versus_camera', 5),
    'greedy_4v8-9': ('MATE-4v8-9.yaml', 'greedy', 32),
    'greedy_4v8-0': ('MATE-4v8-0.yaml', 'greedy', 32),
}
for _s in ('1v1-0', '1v1-9', '1v2-0', '1v2-9', '2v2-0', '2v2-9', '2v4-0', '2v4-9', '4v2-0', '4v2-9', '4v4-0', '4v4-9'):
    CASES['random_' + _s] = (f'MATE-{_s}.yaml', 'random', 64)
    CASES['greedy_' + _s] = (f'MATE-{_s}.yaml', 'greedy', 32)


def run(case, batch, on, seconds=0.4):
    workload, flow, K = CASES[case]
    eng = Engine(read_config(workload), batch, seed=0)
    in_use = eng.set_sub_wave(on)
    eng.enable_policies()
    eng.reset()
    eng.reserve_rollout(K, search='none')
    if flow.startswith('versus'):
        team = flow.split('_')[1]
        k = eng.num_targets if team == 'target' else eng.num_cameras
        mine = (torch.rand((batch, k, 2), device=eng.device) * 2 - 1) * (10.0 if team == 'target' else 2.5)
        launch = lambda: eng.rollout_versus_greedy(team, mine, K, auto_reset=4)      # noqa: E731
    elif flow == 'greedy':
        launch = lambda: eng.rollout_greedy(K, auto_reset=2)      # noqa: E731
    else:
        launch = lambda: eng.rollout_random(K, auto_reset=2)      # noqa: E731
    for _ in range(8):
        launch()
    torch.cuda.synchronize()
    eng.kernel_time(enable=1)
    idle0, t0, n = eng.idle_steps(), time.perf_counter(), 0
    while time.perf_counter() - t0 < seconds:
        for _ in range(4):
            launch()
        n += 4
        torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    executed = batch * K * n - (eng.idle_steps() - idle0)
    km, timed = eng.kernel_time(enable=False)
    b = algorithmic_bytes(eng.num_cameras, eng.num_targets, eng.num_obstacles)
    out = {'case': case, 'batch': batch, 'envs_per_wave': in_use, 'steps_per_launch': K, 'kernel_avg_us': round(km * 1e3, 2),
           'kernel_frac': round(b * batch * K / (km * 1e-3) / 8e12, 4), 'value': round(executed / elapsed), 'end_to_end_frac': round(b * executed / elapsed / 8e12, 4)}
    eng.close()
    del eng
    torch.cuda.empty_cache()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batches', default='4096,16384,65536')
    ap.add_argument('--cases', default=','.join(CASES))
    ap.add_argument('--out', default='')
    args = ap.parse_args()
    rows = []
    for case in args.cases.split(','):
        for batch in (int(b) for b in args.batches.split(',')):
            pair = [run(case, batch, on) for on in (False, True)]
            rows += pair
            print(f"{case:14s} N={batch:6d}  one/wave {pair[0]['kernel_avg_us']:9.2f} us frac {pair[0]['kernel_frac']:.3f} e2e {pair[0]['end_to_end_frac']:.3f}   "
                  f"four/wave {pair[1]['kernel_avg_us']:9.2f} us frac {pair[1]['kernel_frac']:.3f} e2e {pair[1]['end_to_end_frac']:.3f}   x{pair[0]['kernel_avg_us'] / pair[1]['kernel_avg_us']:.2f}", flush=True)
    if args.out:
        with open(args.out, 'w') as fh:
            json.dump(rows, fh, indent=1)


if __name__ == '__main__':
    main()
