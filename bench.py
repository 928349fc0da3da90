#!/usr/bin/env python3
"""bench.py -- env-steps/s of the HIP step engine on MATE-4v8-9, batch 4096 per GPU.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one `step()` of every environment of the batch (BASELINE.json config[1]: MATE-4v8-9.yaml, 4096 environments per
GPU, uniform random policy generated on-device by the engine's Philox streams, auto-reset of finished episodes inside the
timed loop).  State, actions and observations are resident in HBM for the whole timed region.  With the random policy the K
timed steps run as fused `--rollout R`-step launches (default 256, or 128 / 64 / 32 when K holds fewer than eight such
launches, or K itself when K is smaller -- the driver's `--steps 20` is ONE 20-step launch: rollout_kernel keeps an
environment's records in LDS across the R steps and writes every step's observations, rewards and masks to [R][N][...]
buffers; an environment whose episode ends inside a rollout idles until the reset launch that follows it, and those idle slots
are NOT counted in `value`); `--rollout 0` launches step_kernel once per step.

Timing: W untimed warm-up steps (plus one untimed launch of every launch shape of the timed region, so that no buffer is
allocated and no kernel is first loaded inside it), then the timed region of EXACTLY K steps, bracketed by a barrier +
torch.cuda.synchronize() on both sides, is run `--rep-warmup` (3) untimed + `--reps` (5; 21 when the region is one short
launch, e.g. the driver's `--steps 20`: 0.17 ms) timed times and the MEDIAN of the timed ones is reported (`ms_per_step` x
`steps` = the median repetition).  The episode-statistics gathers run INSIDE the region, each behind the launch whose episodes
it carries (a region of fewer launches than `--stats-interval` gathers once, behind its last launch).

`--gpus N` (N > 1) started without a torchrun environment launches the N ranks itself (a `torch.distributed.run` child process,
started before this process touches the GPU) and exits with its status; under torchrun WORLD_SIZE must equal N.  The batch is
sharded (4096 environments per rank, env index = rank * 4096 + i; no data-path collective); RCCL only all-gathers the episode
statistics on a side stream.

OUTPUT.  Rank 0 prints the driver's JSON line LAST, at most 3000 bytes (`headline_line`): the contract's keys, `config`,
`roofline`, `cpu_baseline` and `details_file`.  Everything else -- the full-precision record, every repetition, start-up times,
the side measurements of tools/bench_extras.py (`learner_flows`, `other_configs`, `reset_amortised`, `n1_api`) -- goes to the
details file (`--details`, default bench_details.json next to this file), and a few numbers of the side measurements are printed
as one `{"side": ...}` line (<= 3500 bytes) BEFORE the headline.
  roofline      HBM roofline of the dominant kernel (rollout_kernel, or step_kernel with --rollout 0): `achieved` = algorithmic
                bytes per launch (7504 B/env-step x the env-steps of one launch) / average launch duration measured
                with HIP events on the launch stream over the timed region; `frac` = achieved / `peak` (vendor 8 TB/s).
                `traffic` = HBM bytes of one launch of this shape from the committed rocprofv3 PMC passes, `frac_traffic` = traffic
                / this run's kernel time / peak: the HBM utilisation (a fused launch keeps state, geometry and actions in LDS, so
                `frac` prices bytes that never move).  `peak_measured` = the largest of the copy / fill / read rates of this box
                under the library's own streaming kernels (mate_engine_hbm_probe), each also listed.
  cpu_baseline  the CPU oracle (oracle/, a parity-checked port of the reference's step path) stepping + packing f32 observations
                for the same workload on the host cores.

`--force-collectives` runs every collective of the N-rank path on ONE rank: the `nccl` (= RCCL) process group with
`device_id`, `dist.barrier()`, the side-stream `all_gather` of StatsGather and the job-level reduction on device tensors -- the
single-GPU rehearsal of what `--gpus 8` executes (tests/test_gpu_multirank.py runs it as a child process).
`--dry-run` exercises the launcher, the job-level reduction and the output path without a GPU (gloo, fabricated timings; the
line says `"data": "dry-run"`): it exists for the CPU test of the N-rank launch path and measures nothing.
`--backend gloo` runs the REAL engine under N ranks that may share a GPU (rank r uses device r mod #devices) with gloo instead
of RCCL for the (CPU-staged) collectives -- the test of the multi-rank path on a one-GPU box.
`--deterministic` replaces the time-based clock warm-up by a fixed number of steps and `--dump PATH` makes every rank save
its final state (`PATH.rank<r>.pt`): a sharded run can then be compared with a single-process run of the whole batch.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOAD = 'MATE-4v8-9.yaml'
BATCH_PER_GPU = 4096
HBM_PEAK_GBS = 8000.0            # MI355X HBM3E vendor peak (MI355X_MICROARCH.md)
# `--rollout 0` (one launch per step in the main timed region): every 17th step_kernel launch carries dispatch events (a stride
# coprime to the reset interval).  The per_step_launch side measurement times its kernel in a pass of its own, see there.
STEP_SAMPLE = 17


def algorithmic_bytes(Nc, Nt, No):
    """B_alg per env-step."""
    Dc = 13 + 9 + 5 * Nt + 4 * No + 7 * Nc
    Dt = 13 + 14 + 7 * Nc + 4 * No + 5 * Nt
    return 4 * (Nc * Dc + Nt * Dt) + 8 * (Nc + Nt) + 2 * (16 * Nc + 35 * Nt + 72) + (24 * Nc + 24 * No + Nt) + 48


def measured_traffic(kernel, steps_per_launch, envs):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC summary (profiles/latest_pmc.json, written by
    tools/pmc_collect.py: separate --pmc passes, one entry per (kernel, steps per launch) at the launch shapes the benches
    run; counters in KiB).  gfx950 correction per the microarchitecture guide and this repo's own calibration
    (tools/pmc_calibrate.py: a 256 MiB copy reports WRITE_SIZE 256.0 MiB and FETCH_SIZE 128.0 MiB): WRITE_SIZE is exact,
    FETCH_SIZE counts half of the bytes read and is doubled.  Only an entry profiled at THIS launch length and batch is
    used (no scaling between launch lengths: a launch's fixed part does not scale); None otherwise."""
    path = os.path.join(ROOT, 'profiles', 'latest_pmc.json')
    try:
        with open(path) as fh:
            k = json.load(fh)[f'{kernel}@{int(steps_per_launch)}']
        if int(k['env_steps_per_launch']) != int(steps_per_launch) * int(envs):
            return None
        return (2.0 * k['FETCH_SIZE'] + k['WRITE_SIZE']) * 1024.0
    except Exception:
        return None


REFERENCE_NUMPY_PER_CORE = 391.0   # env-steps/s of the reference's own NumPy path on MATE-4v8-9, one core of the build container


def cpu_model():
    try:
        with open('/proc/cpuinfo') as fh:
            for line in fh:
                if line.startswith('model name'):
                    return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


def cpu_baseline(seconds=10.0):
    """Oracle (CPU port of the reference step path) on the host cores: step + f32 observation pack, single thread and
    OpenMP over environments.  About `seconds` of OpenMP work + ~3 s single-threaded + ~3 s of
    thread-count selection."""
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from oracle import oracle as O
    import gpu_util as U
    from mate_amd.config import read_config
    cfg = read_config(WORKLOAD)
    cores = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    proto = U.oracle_proto_from_config(cfg, O)

    def rate(batch, threads, min_seconds, min_steps=2):
        batch.step(auto_reset=True, threads=threads)
        t0, steps = time.perf_counter(), 0
        while steps < min_steps or time.perf_counter() - t0 < min_seconds:
            batch.step(auto_reset=True, threads=threads)
            batch.observe(threads=threads)
            steps += 1
        return batch.n * steps / (time.perf_counter() - t0), steps

    # single thread: a 256-environment slice of the same batch (one core steps ~10^4 environments a second)
    small = O.OracleBatch(proto, 256, seed=0, first_env_index=0)
    small.reset(threads=1)
    single, single_steps = rate(small, 1, 3.0)
    batch = O.OracleBatch(proto, BATCH_PER_GPU, seed=0, first_env_index=0)
    batch.reset(threads=min(cores, 32))
    # the thread count that is fastest on this box (containers often expose more CPUs than they may use): >= 0.4 s each
    scan = {}
    for threads in sorted({8, 16, 32, 64, 128, cores}):
        if threads <= cores:
            scan[threads] = rate(batch, threads, 0.4)[0]
    best = max(scan, key=scan.get) if scan else 1
    value, steps = rate(batch, best, seconds, min_steps=10)
    return {
        'value': value, 'unit': 'env-steps/s', 'cores': best, 'kind': 'port',
        'sample': f'{WORKLOAD} batch={BATCH_PER_GPU} x {steps} steps, random policy + f32 observation pack, OpenMP over envs on {best} threads '
                  f'(fastest of {sorted(scan)}); single thread: batch=256 x {single_steps} steps',
        'single_thread': single, 'host_cpus': cores, 'cpu_model': cpu_model(),
        'thread_scan': {str(k): round(v) for k, v in scan.items()},
        'reference_numpy_per_core': REFERENCE_NUMPY_PER_CORE,
        'reference_note': 'the reference\'s own NumPy step() on one core of the BUILD container (Intel Xeon 2.10 GHz; BASELINE.md section 2): '
                          'it is pure Python and never travels to the GPU box',
    }


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=2048)
    ap.add_argument('--warmup', type=int, default=128)
    ap.add_argument('--reps', type=int, default=0,
                    help='repetitions of the timed region; the median is reported (default 0 = 5, or 21 when the region is a single short launch -- '
                         'under a millisecond: one preempted host thread moves a median of five)')
    ap.add_argument('--rep-warmup', type=int, default=3,
                    help='untimed repetitions of the whole measuring loop (region + its bookkeeping) before the timed ones: the first '
                         'passes through that loop are slower than its steady state (clocks, lazily loaded code), whatever ran before')
    ap.add_argument('--batch', type=int, default=BATCH_PER_GPU, help='environments per GPU')
    ap.add_argument('--workload', default=WORKLOAD)
    ap.add_argument('--policy', choices=['random', 'greedy', 'external'], default='random',
                    help='on-device policy: uniform random (headline), GreedyCamera vs GreedyTarget (BASELINE config 3), or '
                         'external = step(actions) with the joint actions in a caller-owned device buffer (learner in the loop)')
    ap.add_argument('--reset-interval', type=int, default=32, help='greedy policy, one launch per step: batched auto-reset every k steps (1 = immediate)')
    ap.add_argument('--rollout-reset-interval', type=int, default=-1,
                    help='fused rollouts: restart finished environments after every k-th launch (1 = after every launch; default: about every '
                         '128 steps for the random policy (k = 128 // launch length), 2 for Greedy vs Greedy, whose ~1.2k-step episodes end '
                         'somewhere in the batch at every step)')
    ap.add_argument('--buffer-gib', type=float, default=48.0, help='cap of the rollout-shaped output buffers [R][N][...] (limits --rollout; 48 of the 288 GiB: 64-step launches of 65536 x MATE-4v8-9)')
    ap.add_argument('--rollout', type=int, default=-1,
                    help='steps fused per launch (rollout_kernel / rollout_greedy_kernel); 0 = one launch per step; -1 (default) = 256 / 128 / 64 '
                         '(random: the longest of which the timed region holds eight) / 48 (greedy), at every batch size (rounds 1-3 went back to '
                         'one launch per step beyond 64 environment-waves per CU, which the fused kernels have since overtaken there too: '
                         '65536 x MATE-4v8-9 7.3e8 against 5.7e8 env-steps/s); capped so that the [R][N][...] buffers stay under --buffer-gib')
    ap.add_argument('--step-reset-interval', type=int, default=32,
                    help='one launch per step (per_step_launch / external_actions / --rollout 0 with the random policy): restart finished '
                         'environments with one reset launch per k steps (1 = a reset launch behind every step); a finished environment idles at most k - 1 steps')
    ap.add_argument('--stats-interval', type=int, default=8, help='launches between two episode-statistics gathers inside the timed loop (0 = none)')
    ap.add_argument('--versus-reset-interval', type=int, default=64,
                    help='learner_flows / versus_greedy: one restart of the finished environments per this many steps (Greedy episodes end after ~1.2 k steps)')
    ap.add_argument('--graph-steps', type=int, default=64, help='external policy: step + auto-reset pairs captured per HIP graph (0 = direct launches)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-extras', action='store_true', help='skip the per_step_launch / external_actions side measurements')
    ap.add_argument('--cpu-seconds', type=float, default=10.0)
    ap.add_argument('--dry-run', action='store_true', help='launcher / reduction plumbing on CPU with gloo: measures nothing')
    ap.add_argument('--backend', choices=['nccl', 'gloo'], default='nccl',
                    help='process-group backend for N > 1: nccl (= RCCL, one GPU per rank) or gloo (ranks may share a GPU: the real engine '
                         'under N ranks on a one-GPU box; statistics are staged through the host)')
    ap.add_argument('--deterministic', action='store_true', help='fixed-length clock warm-up instead of the time-based one (reproducible final state)')
    ap.add_argument('--dump', default='', help='every rank saves its final state and last outputs to <path>.rank<r>.pt')
    ap.add_argument('--max-episode-steps', type=int, default=0, help='override the scenario\'s time limit (tests: episodes that end inside a short run)')
    ap.add_argument('--no-other-configs', action='store_true', help='skip the BASELINE config 3 / 4-shard / 5-shard side measurements')
    ap.add_argument('--other-seconds', type=float, default=1.0, help='timed seconds per entry of other_configs')
    ap.add_argument('--force-collectives', action='store_true',
                    help='one rank: initialise the process group anyway (nccl = RCCL with device_id, or --backend gloo) and run every collective of the '
                         'N-rank path -- barrier, the side-stream all_gather of the episode statistics, the job-level reduction')
    ap.add_argument('--no-side-measurements', action='store_true', help='skip reset_amortised / n1_api')
    ap.add_argument('--details', default='', help='file the full record is written to (default: bench_details.json next to bench.py); the printed line carries its path')
    return ap.parse_args(argv)


def free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def launch_ranks(args):
    """`--gpus N` without a torchrun environment: start the N ranks (one process per GPU over RCCL) and exit with
    their status.  Runs before anything here initialises the GPU (counting devices does not)."""
    if not args.dry_run:
        import torch
        have = torch.cuda.device_count()
        if have < (1 if args.backend == 'gloo' else args.gpus):     # (gloo: ranks may share a GPU)
            raise SystemExit(f'bench.py --gpus {args.gpus}: only {have} GPU(s) visible on this node')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}',
           '--master-addr', '127.0.0.1', '--master-port', str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    env.setdefault('OMP_NUM_THREADS', '1')
    return subprocess.call(cmd, env=env)


class StatsGather:
    """Episode statistics all-gathered OFF the critical path.  The engine accumulates the record of
    every finished episode on the device (Engine.episode_stats: count, return, length, coverage, delivered); a gather is a
    40-byte snapshot of the accumulators on the launch stream (one 64-thread launch, Engine.snapshot_episode_stats) and an event
    behind it, and on a SIDE stream, behind that event, the RCCL all-gather of the snapshot -- with one rank the all-gather is
    the identity and the side stream has nothing to do.  With the gloo backend (ranks sharing a GPU in the one-box test) the side
    stream copies the snapshot into pinned host memory and the collective of the LAST submitted copy runs on the host in result()."""

    def __init__(self, torch, dist, distributed, eng, host_staged=False):
        self.torch, self.dist, self.distributed, self.eng, self.host_staged = torch, dist, distributed, eng, host_staged
        self.side = torch.cuda.Stream(device=eng.device)
        world = dist.get_world_size() if distributed else 1
        self.slots = [torch.zeros(5, dtype=torch.float64, device=eng.device) for _ in range(2)]
        self.gathered = [[torch.zeros(5, dtype=torch.float64, device=eng.device) for _ in range(world)] for _ in range(2)]
        self.host = [torch.zeros(5, dtype=torch.float64).pin_memory() for _ in range(2)] if host_staged else None
        self.events = [torch.cuda.Event() for _ in range(2)]
        self.consumed = [None, None]      # per slot: the side stream's event behind its last reader
        self.turn = self.last = 0
        self.count = 0

    def submit(self):
        torch = self.torch
        if self.consumed[self.turn] is not None:                     # (the collective of two gathers ago has read this slot: long done, no stall)
            torch.cuda.current_stream(self.eng.device).wait_event(self.consumed[self.turn])
        self.eng.snapshot_episode_stats(self.slots[self.turn])      # on the launch stream: what has finished up to here
        if self.host_staged or self.distributed:
            ready = self.events[self.turn]
            ready.record()
            with torch.cuda.stream(self.side):
                self.side.wait_event(ready)
                if self.host_staged:
                    self.host[self.turn].copy_(self.slots[self.turn], non_blocking=True)
                else:
                    self.dist.all_gather(self.gathered[self.turn], self.slots[self.turn])
                if self.consumed[self.turn] is None:
                    self.consumed[self.turn] = torch.cuda.Event()
                self.consumed[self.turn].record(self.side)
        else:
            self.gathered[self.turn][0] = self.slots[self.turn]      # one rank: the all-gather is the identity
        self.last = self.turn
        self.turn ^= 1
        self.count += 1

    def result(self):
        self.side.synchronize()
        if not self.count:
            return None
        if self.host_staged:
            mine = self.host[self.last].clone()
            parts = [self.torch.zeros_like(mine) for _ in range(self.dist.get_world_size())] if self.distributed else [mine]
            if self.distributed:
                self.dist.all_gather(parts, mine)
            total = self.torch.stack(parts).sum(dim=0).tolist()
        else:
            total = self.torch.stack(self.gathered[self.last]).sum(dim=0).tolist()
        episodes = max(total[0], 1.0)
        return {'gathers_in_timed_loops': self.count, 'episodes_finished': total[0], 'mean_episode_return': total[1] / episodes,
                'mean_episode_length': total[2] / episodes, 'mean_final_coverage_rate': total[3] / episodes, 'mean_delivered': total[4] / episodes}


def dry_run(args, world, rank):
    """The N-rank launch path and the job-level reduction with gloo and fabricated numbers (CPU test only)."""
    import torch
    import torch.distributed as dist
    from mate_amd.distributed import reduce_job, shard_of
    if world > 1:
        dist.init_process_group('gloo')
    first, count = shard_of(args.batch * world, rank, world)
    elapsed, executed = 1.0 + 0.25 * rank, float(count * args.steps)
    elapsed, executed, stats = reduce_job(elapsed, executed, torch.tensor([float(rank), float(first), 1.0], dtype=torch.float64))
    if rank == 0:
        full = {'metric': 'dry-run (no GPU work)', 'value': executed / elapsed, 'unit': 'env-steps/s', 'n_gpus': world, 'steps': args.steps,
                'warmup': args.warmup, 'ms_per_step': elapsed / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'weak',
                'vs_baseline': None, 'dtype': 'f64', 'data': 'dry-run', 'config': {'workload': 'none', 'global_batch': args.batch * world},
                'ranks': world, 'shard_first_env_mean': float(stats[1])}
        emit(full, args.details)
    if world > 1:
        dist.destroy_process_group()


OUT = sys.stdout       # where emit() prints; main() points it at the process's REAL stdout and sends everything else to stderr


def own_stdout():
    """The driver parses the LAST stdout line.  Libraries write to file descriptor 1 behind Python's back -- RCCL prints `Librccl path : ...`
    through C stdio, flushed when the process ends, i.e. AFTER the headline (and from every rank of an N-rank run) --, so descriptor 1 is
    pointed at stderr for the whole process, and the lines of emit() go to a private duplicate of the original stdout."""
    global OUT
    sys.stdout.flush()
    OUT = os.fdopen(os.dup(1), 'w')
    os.dup2(2, 1)


def main():
    t_main = time.perf_counter()
    args = parse_args()
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        raise SystemExit(launch_ranks(args))
    own_stdout()
    if args.force_collectives and 'WORLD_SIZE' not in os.environ:      # a one-rank rendezvous of our own
        os.environ.update({'WORLD_SIZE': '1', 'RANK': '0', 'LOCAL_RANK': '0', 'MASTER_ADDR': '127.0.0.1', 'MASTER_PORT': str(free_port())})
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        raise SystemExit(f'bench.py --gpus {args.gpus} started with WORLD_SIZE={world}: launch one rank per GPU')
    if args.dry_run:
        return dry_run(args, world, rank)

    import torch
    import torch.distributed as dist
    from mate_amd.config import read_config
    from mate_amd.engine import Engine

    distributed = world > 1 or args.force_collectives
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X: the engine has no CPU fallback')
    host_staged = args.backend == 'gloo'
    if host_staged:
        local_rank = local_rank % torch.cuda.device_count()      # ranks may share a GPU
    torch.cuda.set_device(local_rank)
    startup = {}
    t_phase = time.perf_counter()
    if distributed:
        if host_staged:
            dist.init_process_group('gloo')
        else:
            dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
            # the communicator and RCCL's own device buffers exist BEFORE anything here takes a large share of the HBM (the
            # candidate search of reserve_rollout holds up to 45 % of the free memory for seconds): one barrier and one all_gather
            # of the statistics record's shape, on the side of the first launch
            warm = torch.zeros(5, dtype=torch.float64, device='cuda')
            dist.barrier()
            dist.all_gather([torch.zeros_like(warm) for _ in range(world)], warm)
            torch.cuda.synchronize()
    startup['process_group_s'] = time.perf_counter() - t_phase

    t_phase = time.perf_counter()
    cfg = read_config(args.workload, **({'max_episode_steps': args.max_episode_steps} if args.max_episode_steps > 0 else {}))
    eng = Engine(cfg, args.batch, device=local_rank, seed=0, first_env_index=rank * args.batch)
    torch.cuda.synchronize()
    startup['engine_create_s'] = time.perf_counter() - t_phase
    b_obs = 4 * (eng.num_cameras * eng.camera_obs_dim + eng.num_targets * eng.target_obs_dim) + 48   # written per env-step
    R = args.rollout
    if args.policy == 'external':
        R = 0
    if R < 0:
        if args.policy == 'random':             # longer launches while the timed region still holds eight of them
            R = next((r for r in (256, 128, 64) if args.steps // r >= 8), 32)
        else:                                   # Greedy vs Greedy: 48 steps (longer launches lose more to the idle slots of finished
            R = 48 if args.steps >= 8 * 48 else 32     # episodes than they save in launches; measured 16 .. 96)
    if R > 0:        # rollout buffers [R][N][...] capped (default 8 GiB of the 288)
        R = max(1, min(R, args.steps, (int(args.buffer_gib * (1 << 30))) // (args.batch * b_obs)))
    external = None
    if args.policy == 'greedy':
        eng.enable_policies()
        step = lambda: eng.step_greedy(auto_reset=args.reset_interval)     # noqa: E731
    elif args.policy == 'external':
        sys.path.insert(0, os.path.join(ROOT, 'tools'))
        from bench_extras import ExternalActions
        external = ExternalActions(torch, eng, args.graph_steps, args.step_reset_interval)
        step = external.step
    else:
        step = lambda: eng.step_random(auto_reset=args.step_reset_interval)     # noqa: E731
    rollout_fn = eng.rollout_greedy if args.policy == 'greedy' else eng.rollout_random
    # default restart cadence of the fused flows: Greedy vs Greedy every 2 launches (profiles/HISTORY.md 3.1c: k = 2..4 are within 1.5 %); random policy about every 128
    # steps whatever the launch length (after each 128-step launch; after every 6th 20-step launch): a finished environment then idles
    # ~64 of its 10^4 steps (idle slots are not counted in `value`) and short launches do not each drag an idle reset launch behind them
    rollout_resets = args.rollout_reset_interval if args.rollout_reset_interval > 0 else (2 if args.policy == 'greedy' else max(1, 128 // max(R, 1)))
    rollout = lambda n, auto_reset=True: rollout_fn(n, auto_reset=rollout_resets)     # noqa: E731
    gather = StatsGather(torch, dist, distributed, eng, host_staged=host_staged) if args.stats_interval > 0 else None

    def run(steps, timed=False):
        """exactly `steps` env.step()s of the whole batch"""
        if external is not None:
            external.run(steps)
            if timed and gather is not None:
                gather.submit()
        elif R > 0:
            lengths = [R] * (steps // R) + ([steps % R] if steps % R else [])
            # a gather every `stats_interval` launches, each enqueued behind the launch whose episodes it carries (its copy and its
            # collective run on the side stream under the launches that follow).  A region that holds fewer launches than the
            # interval gathers ONCE, behind its last launch: the gather then carries the region's own episodes, and its copy +
            # collective sit between the kernel's end and the region's closing synchronise -- they are part of `value`
            # (through round 5 `value` was taken with that gather's marker ahead of the region: +8 % on the driver's 20-step region)
            short = timed and gather is not None and len(lengths) < args.stats_interval
            for i, n in enumerate(lengths):
                rollout(n, auto_reset=True)
                if timed and gather is not None and ((i + 1) % args.stats_interval == 0 or (short and i == len(lengths) - 1)):
                    gather.submit()
        else:
            every = args.stats_interval * 128
            short = timed and gather is not None and steps < every
            for i in range(steps):
                step()
                if timed and gather is not None and ((i + 1) % every == 0 or (short and i == steps - 1)):
                    gather.submit()

    t_phase = time.perf_counter()
    eng.reset()
    torch.cuda.synchronize()
    startup['first_reset_s'] = time.perf_counter() - t_phase
    if R > 0:
        # [R][N][...] output buffers: allocated here, never inside the timed region.  This process owns the GPU, so it asks for the
        # deep search of the observation blocks (seconds, and a transient footprint of up to 45 % of the free HBM: Engine.reserve_rollout)
        eng.reserve_rollout(R, search='deep')
        startup['reserve_rollout_s'] = eng.reserve_seconds
    run(args.warmup)
    # one untimed pass over every launch shape of the timed region (kernel code objects loaded, graphs instantiated)
    if R > 0:
        run(R + (args.steps % R))
    elif external is not None:
        external.run(min(args.steps, max(args.graph_steps, 1) + args.steps % max(args.graph_steps, 1)))
    # ... and, whatever W is, at least ~0.25 s of the timed region's own launches: the GPU raises its clocks under load, and a
    # 20-step region lasts 0.2 ms (reported as `warmup_extra_steps`; W itself is honoured above)
    extra_steps = 0
    eng.kernel_time(enable=1 if R > 0 else STEP_SAMPLE)     # ... with the dispatch-event launch path of the timed region (events created, runtime warmed)
    t_warm = time.perf_counter()
    while (extra_steps < 4 * args.steps) if args.deterministic else (time.perf_counter() - t_warm < 0.25):
        chunk = min(args.steps, 8 * R) if R > 0 else 64      # back to back like the timed region (sustained, not boost, clocks)
        run(chunk)
        extra_steps += chunk
        torch.cuda.synchronize()
    eng.kernel_time(enable=False)
    if gather is not None:
        gather.submit()              # side stream, copies and (N > 1) the RCCL communicator warmed outside the timed region
        gather.result()
        gather.count = 0

    def barrier():
        torch.cuda.synchronize()
        if distributed:          # (one rank: the barrier is the synchronisation itself; a second one would only add its own 3-4 us to a 0.16 ms region)
            dist.barrier()
            torch.cuda.synchronize()

    if args.reps <= 0:
        args.reps = 21 if (R > 0 and args.steps <= R and args.steps * args.batch <= (1 << 18)) else 5
    from mate_amd.distributed import reduce_job
    startup['startup_s'] = time.perf_counter() - t_main      # everything of this rank's main() before the first pass through the measuring loop
    short_region = (R > 0 and -(-args.steps // R) < args.stats_interval) or (R == 0 and external is None and args.steps < args.stats_interval * 128)
    rep_ms, rep_executed, kernel_times = [], [], []
    for rep in range(max(0, args.rep_warmup) + max(1, args.reps)):
        eng.kernel_time(enable=1 if R > 0 else STEP_SAMPLE)   # HIP-event pair around every launch of the dominant kernel (every 17th one-step launch)
        barrier()
        idle0 = eng.idle_steps()
        allocated0 = torch.cuda.memory_allocated()
        t0 = time.perf_counter()
        run(args.steps, timed=True)
        barrier()
        elapsed = time.perf_counter() - t0
        executed = args.batch * args.steps - (eng.idle_steps() - idle0)   # env-steps actually simulated by this rank
        assert torch.cuda.memory_allocated() <= allocated0, 'allocation inside the timed region'
        kernel_times.append(eng.kernel_time(enable=False))
        stats = eng.scalars[:, [1, 3, 6]].mean(dim=0)    # reward, coverage, delivered: logging only
        if host_staged:
            stats = stats.cpu()
        elapsed, executed, stats = reduce_job(elapsed, executed, stats, device='cpu' if host_staged else 'cuda', force=args.force_collectives)   # MAX time, SUM env-steps, gathered stats
        if rep < max(0, args.rep_warmup):
            kernel_times.pop()
            if gather is not None:
                gather.count = 0
            continue                                  # an untimed pass through the measuring loop
        rep_ms.append(elapsed * 1e3)
        rep_executed.append(executed)
    order = sorted(range(len(rep_ms)), key=lambda i: rep_ms[i])
    mid = order[len(order) // 2]                      # the median repetition (upper median for an even count)
    elapsed, executed = rep_ms[mid] * 1e-3, rep_executed[mid]
    kernel_ms, launches = kernel_times[mid]
    flow = eng.last_flow
    gathered_stats = gather.result() if gather is not None else None
    keys = ('process_group_s', 'engine_create_s', 'first_reset_s', 'reserve_rollout_s', 'startup_s')
    mine = torch.tensor([startup.get(k, 0.0) for k in keys], dtype=torch.float64, device='cpu' if host_staged or not distributed else 'cuda')
    per_rank = [torch.zeros_like(mine) for _ in range(world)] if distributed else [mine]
    if distributed:
        dist.all_gather(per_rank, mine)
    startup_line = {k: [round(float(t[i]), 3) for t in per_rank] for i, k in enumerate(keys)}

    if args.dump:      # every rank: the final state and the last outputs of its shard (compared across shardings by the tests)
        torch.cuda.synchronize()
        torch.save({'rank': rank, 'world': world, 'first_env_index': rank * args.batch, 'state': eng.export_state().cpu(),
                    'scalars': eng.scalars.cpu(), 'episode_stats': eng.episode_stats.cpu(), 'idle_steps': eng.idle_steps(),
                    'last_rollout_scalars': (eng._rollout['scalars'].cpu() if getattr(eng, '_rollout', None) else None)},
                   f'{args.dump}.rank{rank}.pt')
    if rank == 0:
        from mate_amd._native import hbm_rates
        rates = hbm_rates(local_rank) if not args.dump else {}
        total_envs = args.batch * world
        value = executed / elapsed            # == total_envs * steps / elapsed unless environments idled for a batched reset
        b_alg = algorithmic_bytes(eng.num_cameras, eng.num_targets, eng.num_obstacles)
        # env-steps of the average timed launch (rollouts: R, and the remainder launch when K % R != 0)
        steps_per_launch = (args.steps / launches) if (R > 0 and launches > 0) else 1.0
        bytes_per_launch = b_alg * args.batch * steps_per_launch
        kernel_s = kernel_ms * 1e-3
        achieved = bytes_per_launch / kernel_s / 1e9 if kernel_ms > 0 else 0.0
        kernel = ('step_kernel' if R == 0 else 'rollout_greedy_kernel' if args.policy == 'greedy' else 'rollout_kernel')
        headline_case = args.batch == BATCH_PER_GPU and args.workload == WORKLOAD and args.policy != 'greedy'
        traffic = measured_traffic(kernel, steps_per_launch, args.batch) if (headline_case and float(steps_per_launch).is_integer()) else None
        policy_text = {'random': 'uniform random policy (on-device Philox)', 'greedy': 'on-device GreedyCamera vs GreedyTarget', 'external': 'joint actions from a caller-owned device buffer'}[args.policy]
        full = {
            'metric': f'env-steps/sec {args.workload[:-5]} batch={args.batch} per GPU ({args.policy} policy, auto-reset)',
            'value': value, 'unit': 'env-steps/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': elapsed / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
            'config': {'workload': f'{args.workload} batch={args.batch} envs per GPU, {policy_text}, '
                                   + (f'fused {R}-step launches, restarts every {rollout_resets} launch(es)' if R > 0 else 'one launch per step, auto-reset'),
                       'global_batch': total_envs, 'parallelism': f'env-shard x{world}', 'steps_per_launch': R if R > 0 else 1,
                       'backend': ('gloo (ranks may share a GPU; statistics staged through the host)' if host_staged else 'nccl (RCCL)') if distributed else 'single process',
                       'collectives': ('forced on one rank: barrier, side-stream all_gather, job reduction' if args.force_collectives and world == 1
                                       else 'barrier, side-stream all_gather, job reduction' if distributed else 'none (one rank)')},
            'roofline': {
                'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBS,
                # HBM bytes of one launch of this shape from the committed PMC passes (2 * FETCH_SIZE + WRITE_SIZE, profiles/latest_pmc.json), and
                # what that is of the peak at THIS run's kernel time: the honest utilisation (a fused launch keeps the state, the geometry and
                # the actions of SURVEY 8d's per-step figure in LDS: they never reach HBM, so `frac` prices bytes that are not moved)
                'traffic': traffic, 'frac_traffic': (traffic / kernel_s / 1e9 / HBM_PEAK_GBS) if (traffic and kernel_ms > 0) else None,
                'kernel': '%s<float, %s, %s>' % (kernel, 'FixedShape' if eng.specialised else 'AnyShape', ('FLOW_ANY', 'FLOW_RANDOM', 'FLOW_ACT_F32', 'FLOW_GREEDY')[flow]),
                'kernel_avg_us': kernel_ms * 1e3, 'launches_timed': launches, 'env_steps_per_launch': args.batch * steps_per_launch,
                'algorithmic_bytes_per_env_step': b_alg,
                'end_to_end_frac': b_alg * value / world / 1e9 / HBM_PEAK_GBS,
                # the rates of this box under the library's own streaming kernels (mate_engine_hbm_probe, 1 GiB, median of 5): a copy's
                # read + write bytes, a write-only fill, a read-only pass; peak_measured = the largest of them
                'peak_measured': max(rates.values()) if rates else None, 'copy_gbs': rates.get('copy'), 'fill_gbs': rates.get('fill'), 'read_gbs': rates.get('read'),
                'fused_note': 'fused launch: state, geometry and actions of the 8d per-step bytes stay in LDS; frac_traffic is the HBM utilisation' if R > 0 else None,
            },
            'timing': {'reps': len(rep_ms), 'rep_ms': [round(v, 4) for v in rep_ms], 'rep_warmup': max(0, args.rep_warmup), 'warmup_extra_steps': extra_steps,
                       'definition': 'median repetition of the K-step timed region; statistics gathers inside the region, each behind the launch whose episodes it carries'},
            'episode_stats': {'mean_target_reward': float(stats[0]), 'mean_coverage_rate': float(stats[1]),
                              'mean_delivered': float(stats[2]), 'gathered_in_loop': gathered_stats,
                              'stats_gather': 'none' if gather is None else 'once, behind the last launch of the region' if short_region
                                              else f'every {args.stats_interval} launches inside the region'},
            'startup': startup_line,
            'observation_block_candidates_gbs': [[round(x) for x in r] for r in getattr(eng, 'block_rates', [])],
        }
        default_case = world == 1 and args.policy == 'random' and args.workload == WORKLOAD and args.batch == BATCH_PER_GPU
        side_wanted = not args.dump and world == 1 and args.policy == 'random'
        if side_wanted and not (args.no_side_measurements and args.no_other_configs and args.no_extras):
            sys.path.insert(0, os.path.join(ROOT, 'tools'))
            import bench_extras as X
            if default_case and not args.no_side_measurements:
                full['reset_amortised'] = X.measure_reset_amortised(torch, eng, cfg, value, elapsed / args.steps)
            eng.close()          # (the headline engine's buffers are released first)
            eng._rollout = None
            torch.cuda.empty_cache()
            if default_case and not args.no_other_configs:
                # the other BASELINE configurations that fit one GPU, about a second each
                full['other_configs'] = [X.measure_other_config(torch, local_rank, spec, args.other_seconds, args.buffer_gib) for spec in X.OTHER_CONFIGS]
            if not args.no_extras and R > 0:
                # the learner-facing per-step flows (profiles/HISTORY.md 3.1e), at this run's batch and -- the default run -- at the two larger batches
                batches = [args.batch] + ([b for b in X.LEARNER_BATCHES if b != args.batch] if default_case else [])
                full['learner_flows'] = [X.measure_learner_flows(torch, local_rank, args.workload, b, args.graph_steps, args.step_reset_interval, args.versus_reset_interval)
                                         for b in batches]
            if default_case and not args.no_side_measurements:
                full['n1_api'] = X.measure_n1_api(torch)
        if not args.no_cpu_baseline and world == 1 and args.policy == 'random' and args.workload == WORKLOAD:
            full['cpu_baseline'] = cpu_baseline(args.cpu_seconds)
        emit(full, args.details)
    if distributed:
        dist.destroy_process_group()


LINE_LIMIT = 3000      # bytes of the final stdout line (the driver keeps ~8 KB of stdout: round 5's 22 KB line lost its head and was unparsable)
SIDE_LIMIT = 3500      # ... and of the `side` summary line printed before it


def _sig(v, digits=4):
    """Numbers of the printed lines to `digits` significant digits (the details file keeps full precision)."""
    if isinstance(v, float):
        return float(f'%.{digits}g' % v)
    if isinstance(v, dict):
        return {k: _sig(x, digits) for k, x in v.items() if x is not None or k in ('vs_baseline', 'traffic')}
    if isinstance(v, (list, tuple)):
        return [_sig(x, digits) for x in v]
    return v


def headline_line(full, details_file):
    """The ONE line the driver parses: the contract's keys, `roofline` and `cpu_baseline` -- nothing else.  <= LINE_LIMIT bytes."""
    keep = ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype', 'data')
    line = {k: full[k] for k in keep if k in full}
    line['config'] = {k: full['config'][k] for k in ('workload', 'global_batch', 'parallelism', 'steps_per_launch') if k in full.get('config', {})}
    if 'roofline' in full:
        r = full['roofline']
        line['roofline'] = {k: r.get(k) for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'frac_traffic', 'kernel', 'kernel_avg_us', 'launches_timed',
                                                   'algorithmic_bytes_per_env_step', 'end_to_end_frac', 'peak_measured', 'copy_gbs', 'fill_gbs', 'read_gbs', 'fused_note')}
    if 'cpu_baseline' in full:
        c = full['cpu_baseline']
        line['cpu_baseline'] = {k: c.get(k) for k in ('value', 'unit', 'cores', 'kind', 'sample', 'single_thread', 'cpu_model', 'reference_numpy_per_core')}
    if 'timing' in full:
        line['reps'] = full['timing']['reps']
    line['details_file'] = details_file
    line = _sig(line, 5)
    for key in ('value', 'ms_per_step'):           # the two the driver checks against its own clock: full precision
        if key in full:
            line[key] = full[key]
    return line


def side_line(full):
    """A few numbers of the side measurements (their full records are in the details file), printed on the line BEFORE the headline."""
    side = {}
    for oc in full.get('other_configs', []):
        side.setdefault('other_configs', []).append({'config': oc['config'], 'value': oc['value'], 'kernel_avg_us': oc['kernel_avg_us'], 'frac': oc['frac'],
                                                     'end_to_end_frac': oc['end_to_end_frac']})
    for fl in full.get('learner_flows', []):
        entry = {'batch': fl['batch']}
        for name, e in fl.items():
            if isinstance(e, dict) and 'value' in e:
                entry[name] = {'value': e['value'], 'us_per_step': e['us_per_step'], 'end_to_end_frac': e['end_to_end_frac']}
                if 'roofline_frac' in e:
                    entry[name]['roofline_frac'] = e['roofline_frac']
        side.setdefault('learner_flows', []).append(entry)
    if 'reset_amortised' in full:
        side['reset_amortised'] = {k: full['reset_amortised'][k] for k in ('whole_batch_reset_ms', 'value_with_resets', 'cost_frac')}
    if 'n1_api' in full:
        side['n1_api'] = {k: full['n1_api'][k] for k in ('value', 'reference_numpy')}
    if 'episode_stats' in full and full['episode_stats'].get('gathered_in_loop'):
        side['episodes_gathered'] = full['episode_stats']['gathered_in_loop'].get('episodes_finished')
    return {'side': _sig(side, 3)} if side else None


def emit(full, details_file):
    """Full record -> `details_file`; stdout: the `side` summary line (if any), then the headline line LAST."""
    details_file = details_file or os.path.join(ROOT, 'bench_details.json')
    try:
        with open(details_file, 'w') as fh:
            json.dump(full, fh, indent=1)
    except OSError as exc:
        sys.stderr.write(f'bench.py: details not written to {details_file}: {exc}\n')
        details_file = None
    side = side_line(full)
    if side is not None:
        text = json.dumps(side, separators=(',', ':'))
        if len(text) > SIDE_LIMIT:       # (never at the cost of the headline's place in the driver's tail)
            text = json.dumps({'side': 'see details_file'})
        print(text, file=OUT, flush=True)
    shown = details_file and (os.path.relpath(details_file, ROOT) if os.path.abspath(details_file).startswith(ROOT + os.sep) else os.path.abspath(details_file))
    line = headline_line(full, shown)
    text = json.dumps(line)
    if len(text) > LINE_LIMIT:
        for key in ('cpu_baseline.sample', 'roofline.fused_note', 'roofline.kernel'):
            group, item = key.split('.')
            if group in line and item in line[group]:
                line[group][item] = str(line[group][item])[:60]
        text = json.dumps(line)
    assert len(text) <= LINE_LIMIT, len(text)
    print(text, file=OUT, flush=True)


if __name__ == '__main__':
    main()
