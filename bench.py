#!/usr/bin/env python3
"""bench.py -- env-steps/s of the HIP step engine on MATE-4v8-9, batch 4096 per GPU.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one `step()` of every environment of the batch (BASELINE.json config[1]:
MATE-4v8-9.yaml, 4096 environments per GPU, uniform random policy generated on-device by the
engine's Philox streams, auto-reset of finished episodes inside the timed loop).  State,
actions and observations are resident in HBM for the whole timed region.  With the random
policy the K timed steps run as fused `--rollout R`-step launches (default 128, the usual horizon of an on-policy
update, or 64 / 32 when K holds fewer than eight such launches: rollout_kernel
keeps an environment's records in LDS across the R steps and writes every step's observations,
rewards and masks to [R][N][...] buffers; an environment whose episode ends inside a rollout
idles until the reset launch that follows it, and those idle slots are NOT counted in `value`);
`--rollout 0` launches step_kernel once per step, and the default run reports that mode too
(`per_step_launch`).  For N > 1 the
batch is sharded (4096 environments per rank, env index = rank * 4096 + i; no data-path
collective); RCCL only all-gathers the episode statistics after the timed region.

Rank 0 prints ONE JSON line (see the driver contract).  Extra objects:
  roofline      HBM roofline of the dominant kernel (rollout_kernel, or step_kernel with --rollout 0):
                algorithmic bytes per launch (7504 B/env-step x the env-steps of one
                launch) / average launch duration measured with HIP events on the launch stream over
                the timed region.  `achieved_resident` / `frac_resident` price a rollout launch at the
                bytes it must really move (R observation sets, ONE state round trip and one geometry
                read per environment) -- the stricter figure.
  cpu_baseline  the CPU oracle (oracle/, a parity-checked port of the reference's step path)
                stepping + packing f32 observations for the same workload on the host cores.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOAD = 'MATE-4v8-9.yaml'
BATCH_PER_GPU = 4096
HBM_PEAK_GBS = 8000.0   # MI355X HBM3E vendor peak (MI355X_MICROARCH.md); measured copy peak ~6290 GB/s


def algorithmic_bytes(Nc, Nt, No):
    """B_alg per env-step."""
    Dc = 13 + 9 + 5 * Nt + 4 * No + 7 * Nc
    Dt = 13 + 14 + 7 * Nc + 4 * No + 5 * Nt
    return 4 * (Nc * Dc + Nt * Dt) + 8 * (Nc + Nt) + 2 * (16 * Nc + 35 * Nt + 72) + (24 * Nc + 24 * No + Nt) + 48


def measured_traffic(kernel='step_kernel'):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC summary (separate --pmc passes, counters in
    KiB; see profiles/README.md).  gfx950 correction per the microarchitecture guide and this repo's own calibration
    (tools/pmc_calibrate.py: a 256 MiB copy reports WRITE_SIZE 256.0 MiB and FETCH_SIZE 128.0 MiB): WRITE_SIZE is
    exact, FETCH_SIZE counts half of the bytes read and is doubled.  None when no profile is present."""
    path = os.path.join(ROOT, 'profiles', 'latest_pmc.json')
    try:
        with open(path) as fh:
            k = json.load(fh)[kernel]
        return (2.0 * k['FETCH_SIZE'] + k['WRITE_SIZE']) * 1024.0
    except Exception:
        return None


def cpu_baseline(seconds=10.0):
    """Oracle (CPU port of the reference step path) on the host cores: step + f32 observation pack."""
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from oracle import oracle as O
    import gpu_util as U
    from mate_amd.config import read_config
    cfg = read_config(WORKLOAD)
    cores = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    proto = U.oracle_proto_from_config(cfg, O)
    batch = O.OracleBatch(proto, BATCH_PER_GPU, seed=0, first_env_index=0)
    batch.reset(threads=min(cores, 32))
    # pick the thread count that is fastest on this box (containers often expose more CPUs than they may use)
    best = (0.0, 1)
    for threads in sorted({1, 8, 16, 32, 64, 128, cores}):
        if threads > cores:
            continue
        batch.step(auto_reset=True, threads=threads)
        t0 = time.perf_counter()
        for _ in range(2):
            batch.step(auto_reset=True, threads=threads)
            batch.observe(threads=threads)
        rate = 2 * BATCH_PER_GPU / (time.perf_counter() - t0)
        if rate > best[0]:
            best = (rate, threads)
    cores = best[1]
    t0 = time.perf_counter()
    steps = 0
    while True:
        batch.step(auto_reset=True, threads=cores)
        batch.observe(threads=cores)
        steps += 1
        if steps >= 20 and time.perf_counter() - t0 >= seconds:
            break
    dt = time.perf_counter() - t0
    return {
        'value': BATCH_PER_GPU * steps / dt, 'unit': 'env-steps/s', 'cores': cores, 'kind': 'port',
        'sample': f'{WORKLOAD} batch={BATCH_PER_GPU} x {steps} steps (random policy, f32 observation pack), '
                  f'{dt:.1f} s, OpenMP over environments',
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=2048)
    ap.add_argument('--warmup', type=int, default=128)
    ap.add_argument('--batch', type=int, default=BATCH_PER_GPU, help='environments per GPU')
    ap.add_argument('--workload', default=WORKLOAD)
    ap.add_argument('--policy', choices=['random', 'greedy'], default='random',
                    help='on-device policy: uniform random (headline) or GreedyCamera vs GreedyTarget (BASELINE config 3)')
    ap.add_argument('--reset-interval', type=int, default=32, help='greedy policy: batched auto-reset every k steps (1 = immediate)')
    ap.add_argument('--rollout', type=int, default=-1,
                    help='steps fused per launch (rollout_kernel / rollout_greedy_kernel); 0 = one launch per step; -1 (default) = 128 '
                         '(random) / 32 (greedy) while the batch is at most 64 / 32 environment-waves per CU, else 0 (the fused '
                         'kernels trade occupancy for registers and LDS); capped so that the [R][N][...] buffers stay under 4 GiB')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-seconds', type=float, default=10.0)
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from mate_amd.config import read_config
    from mate_amd.engine import Engine

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    distributed = world > 1
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X: the engine has no CPU fallback')
    torch.cuda.set_device(local_rank)
    if distributed:
        dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))

    cfg = read_config(args.workload)
    eng = Engine(cfg, args.batch, device=local_rank, seed=0, first_env_index=rank * args.batch)
    b_obs = 4 * (eng.num_cameras * eng.camera_obs_dim + eng.num_targets * eng.target_obs_dim) + 48   # written per env-step
    R = args.rollout
    if R < 0:
        cus = torch.cuda.get_device_properties(local_rank).multi_processor_count
        R = 32 if args.batch <= (64 if args.policy == 'random' else 32) * cus else 0
        if R and args.policy == 'random':       # longer launches while the timed region still holds eight of them
            R = next((r for r in (128, 64) if args.steps // r >= 8), 32)
    if R > 0:        # rollout buffers [R][N][...] capped at 4 GiB
        R = max(1, min(R, args.steps, (4 << 30) // (args.batch * b_obs)))
    if args.policy == 'greedy':
        eng.enable_policies()
        step = lambda: eng.step_greedy(auto_reset=args.reset_interval)     # noqa: E731
    else:
        step = lambda: eng.step_random(auto_reset=True)     # noqa: E731

    def run(steps):
        """exactly `steps` env.step()s of the whole batch"""
        if R > 0:
            rollout = eng.rollout_greedy if args.policy == 'greedy' else eng.rollout_random
            for _ in range(steps // R):
                rollout(R, auto_reset=True)
            if steps % R:
                rollout(steps % R, auto_reset=True)
        else:
            for _ in range(steps):
                step()

    eng.reset()
    run(args.warmup)
    if R > 0 and args.steps % R:
        run(args.steps % R)      # the remainder launch of the timed region, warmed too
    eng.kernel_time(enable=1 if R > 0 else 16)   # HIP-event pair around (every 16th) launch of the dominant kernel in the timed region

    def barrier():
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    barrier()
    idle0 = eng.idle_steps()
    t0 = time.perf_counter()
    run(args.steps)
    barrier()
    elapsed = time.perf_counter() - t0
    executed = args.batch * args.steps - (eng.idle_steps() - idle0)   # env-steps actually simulated by this rank
    kernel_ms, launches = eng.kernel_time(enable=False)
    flow = eng.last_flow
    per_step = None
    if R > 0:        # the same workload with one step_kernel launch per step, reported beside the headline
        k2 = min(args.steps, 1000)
        eng.kernel_time(enable=16)
        barrier()
        idle1 = eng.idle_steps()
        t1 = time.perf_counter()
        for _ in range(k2):
            step()
        barrier()
        e2 = time.perf_counter() - t1
        km2, _ = eng.kernel_time(enable=False)
        per_step = {'value': (args.batch * k2 - (eng.idle_steps() - idle1)) * world / e2, 'unit': 'env-steps/s', 'steps': k2, 'ms_per_step': e2 / k2 * 1e3,
                    'kernel': 'step_kernel', 'kernel_avg_us': km2 * 1e3,
                    'roofline_frac': (algorithmic_bytes(eng.num_cameras, eng.num_targets, eng.num_obstacles) * args.batch / (km2 * 1e-3) / 1e9 / HBM_PEAK_GBS) if km2 > 0 else 0.0}

    stats = eng.scalars[:, [1, 3, 6]].mean(dim=0)    # reward, coverage, delivered: logging only
    from mate_amd.distributed import reduce_job
    elapsed, executed, stats = reduce_job(elapsed, executed, stats, device='cuda')   # MAX time, SUM env-steps, gathered stats

    if rank == 0:
        total_envs = args.batch * world
        value = executed / elapsed            # == total_envs * steps / elapsed unless environments idled for a batched reset
        b_alg = algorithmic_bytes(eng.num_cameras, eng.num_targets, eng.num_obstacles)
        # env-steps of the average timed launch (rollouts: R, and the remainder launch when K % R != 0)
        steps_per_launch = (args.steps / launches) if (R > 0 and launches > 0) else 1.0
        bytes_per_launch = b_alg * args.batch * steps_per_launch
        achieved = bytes_per_launch / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
        # what a rollout launch must really move: every step's observations + scalars, the state and geometry once
        resident = (args.batch * (steps_per_launch * b_obs + (b_alg - b_obs - 8 * (eng.num_cameras + eng.num_targets)))
                    if R > 0 else bytes_per_launch)
        kernel = ('step_kernel' if R == 0 else 'rollout_greedy_kernel' if args.policy == 'greedy' else 'rollout_kernel')
        default_case = args.batch == BATCH_PER_GPU and args.workload == WORKLOAD and (R == 128 or args.policy == 'greedy' or R == 0)
        line = {
            'metric': f'env-steps/sec {args.workload[:-5]} batch={args.batch} per GPU ({args.policy} policy, auto-reset)',
            'value': value, 'unit': 'env-steps/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': elapsed / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
            'config': {'workload': f'{args.workload} batch={args.batch} envs per GPU, '
                                   + ('uniform random policy (on-device Philox), ' if args.policy == 'random' else 'on-device GreedyCamera vs GreedyTarget policies, ')
                                   + (f'fused {R}-step rollout launches, auto-reset after each launch' if R > 0 else 'one launch per step, auto-reset'),
                       'global_batch': total_envs, 'parallelism': f'env-shard x{world}', 'steps_per_launch': R if R > 0 else 1},
            'roofline': {
                'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                'frac': achieved / HBM_PEAK_GBS,
                'traffic': measured_traffic(kernel) if default_case and args.policy == 'random' else None,
                'traffic_unit': 'bytes per launch (rocprofv3 2*FETCH_SIZE + WRITE_SIZE, profiles/latest_pmc.json)',
                'kernel': '%s<float, %s, %s>' % (kernel, 'FixedShape' if eng.specialised else 'AnyShape', ('FLOW_ANY', 'FLOW_RANDOM', 'FLOW_ACT_F32', 'FLOW_GREEDY')[flow]),
                'kernel_avg_us': kernel_ms * 1e3, 'launches_timed': launches, 'env_steps_per_launch': args.batch * steps_per_launch,
                'algorithmic_bytes_per_launch': bytes_per_launch,
                'resident_bytes_per_launch': resident,
                'achieved_resident': resident / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0,
                'frac_resident': resident / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if kernel_ms > 0 else 0.0,
            },
            'episode_stats': {'mean_target_reward': float(stats[0]), 'mean_coverage_rate': float(stats[1]),
                              'mean_delivered': float(stats[2])},
        }
        if per_step is not None:
            line['per_step_launch'] = per_step
        if not args.no_cpu_baseline and world == 1 and args.policy == 'random' and args.workload == WORKLOAD:
            line['cpu_baseline'] = cpu_baseline(args.cpu_seconds)
        print(json.dumps(line), flush=True)
    if distributed:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
