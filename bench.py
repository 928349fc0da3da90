#!/usr/bin/env python3
"""bench.py -- env-steps/s of the HIP step engine on MATE-4v8-9, batch 4096 per GPU.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one `step()` of every environment of the batch (BASELINE.json config[1]:
MATE-4v8-9.yaml, 4096 environments per GPU, uniform random policy generated on-device by the
engine's Philox streams, auto-reset of finished episodes inside the timed loop).  State,
actions and observations are resident in HBM for the whole timed region.  With the random
policy the K timed steps run as fused `--rollout R`-step launches (default 256, or 128 / 64 / 32 when K holds fewer than
eight such launches, or K itself when K is smaller -- the driver's `--steps 20` is ONE 20-step launch: rollout_kernel
keeps an environment's records in LDS across the R steps and writes every step's observations,
rewards and masks to [R][N][...] buffers; an environment whose episode ends inside a rollout
idles until the reset launch that follows it, and those idle slots are NOT counted in `value`);
`--rollout 0` launches step_kernel once per step.  The default run also reports the flows a LEARNER calls, one launch per
step (`learner_flows`, measure_learner_flows: `per_step_launch`, `external_actions` = step(actions) with the joint actions in a
caller-owned device buffer rewritten by a policy kernel before every step, `versus_greedy` = MultiCamera(GreedyTargetAgent) with the
opponents on the device, and the same batch as two groups on two streams), each replayed from HIP graphs, at 4096, 16 384 and
65 536 environments; the headline batch's entries also stand at the top level of the line.

Timing: W untimed warm-up steps (plus one untimed launch of every launch shape of the timed region, so that no
buffer is allocated and no kernel is first loaded inside it), then the timed region of EXACTLY K steps, bracketed by a
barrier + torch.cuda.synchronize() on both sides, is run `--rep-warmup` (3) untimed + `--reps` (5; 21 when the region is one
short launch, e.g. the driver's `--steps 20`: 0.17 ms) timed times and the MEDIAN
of the timed ones is reported (`ms_per_step` x `steps` = the median repetition; every timed repetition is listed in `rep_ms`).

`--gpus N` (N > 1) started without a torchrun environment launches the N ranks itself (a `torch.distributed.run`
child process, started before this process touches the GPU) and exits with its status; under torchrun WORLD_SIZE must
equal N.  The batch is sharded (4096 environments per rank, env index = rank * 4096 + i; no data-path collective);
RCCL only all-gathers the episode statistics: every `--stats-interval` launches inside the timed loop on a side
stream, and once after it.

Rank 0 prints ONE JSON line (see the driver contract).  Extra objects:
  roofline      HBM roofline of the dominant kernel (rollout_kernel, or step_kernel with --rollout 0):
                algorithmic bytes per launch (7504 B/env-step x the env-steps of one
                launch) / average launch duration measured with HIP events on the launch stream over
                the timed region.  `achieved_resident` / `frac_resident` price a rollout launch at the
                bytes it must really move (R observation sets, ONE state round trip and one geometry
                read per environment) -- the stricter figure.  `peak` is the vendor HBM peak, `peak_measured` the
                device-to-device copy rate measured on this pool.
  cpu_baseline  the CPU oracle (oracle/, a parity-checked port of the reference's step path)
                stepping + packing f32 observations for the same workload on the host cores.

  other_configs the other BASELINE.json configurations that fit one GPU, timed for about a second each after the
                headline in the default N = 1 run: config 3 (MATE-8v8-9 x 8192, on-device Greedy vs Greedy, fused
                48-step launches), the per-GPU shard of config 4 (MATE-4v8-0 x 8192) and of config 5
                (MATE-Navigation x 4096), and the WHOLE batches of configs 4 and 5 (65536 / 32768 environments) on this one
                GPU, each with its dominant kernel's dispatch-event average and roofline fraction.

  reset_amortised  what the driver's region never contains: all environments of the random-policy batch hit the time limit together
                every max_episode_steps + 1 steps and restart in one whole-batch reset; `value_with_resets` = the headline with
                that reset's measured time spread over an episode.
  n1_api        BASELINE config 1: the N = 1 NumPy API (mate_amd.MultiAgentTracking, a PCIe copy + sync per step) under the
                evaluate-style harness, random actions, steps/s beside the reference's own 1101.
  startup       seconds per rank: process group, engine creation, first reset, reserve_rollout (candidate search included),
                everything up to the first timed repetition.

`--force-collectives` runs every collective of the N-rank path on ONE rank: the `nccl` (= RCCL) process group with
`device_id`, `dist.barrier()`, the side-stream `all_gather` of StatsGather and the job-level reduction on device tensors -- the
single-GPU rehearsal of what `--gpus 8` executes (tests/test_gpu_multirank.py runs it as a child process).

`--dry-run` exercises the launcher and the job-level reduction without a GPU (gloo, fabricated timings; the line says
`"data": "dry-run"`): it exists for the CPU test of the N-rank launch path and measures nothing.
`--backend gloo` runs the REAL engine under N ranks that may share a GPU (rank r uses device r mod #devices): the
sharded `Engine(first_env_index = rank * batch)`, the device-side statistics accumulators, `StatsGather` and the job-level
reduction, with gloo instead of RCCL for the (CPU-staged) collectives -- the test of the multi-rank path on a one-GPU box.
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
LEARNER_BATCHES = (4096, 16384, 65536)     # `learner_flows` of the default line: the per-step flows at the batches a learner on one MI355X runs
HBM_PEAK_GBS = 8000.0            # MI355X HBM3E vendor peak (MI355X_MICROARCH.md)
HBM_PEAK_MEASURED_GBS = 6290.0   # device-to-device copy on this pool (tools/pmc_calibrate.py, profiles/README.md); the line reports the rate measured IN the run (measure_hbm_copy_peak)
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
        'sample': f'{WORKLOAD} batch={BATCH_PER_GPU} x {steps} steps (random policy, f32 observation pack), OpenMP over environments on '
                  f'{best} threads (fastest of {sorted(scan)}); single thread: batch=256 x {single_steps} steps',
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
    every finished episode on the device (Engine.episode_stats: count, return, length, coverage, delivered); a gather is
    an event on the launch stream, and behind it on a SIDE stream a 40-byte copy of the accumulators and the RCCL
    all-gather of the copy.  Nothing is enqueued on the launch stream but the event.  With the gloo backend (ranks
    sharing a GPU in the one-box test) the side stream copies into pinned host memory and the collective of the LAST
    submitted copy runs on the host in result()."""

    def __init__(self, torch, dist, distributed, eng, host_staged=False):
        self.torch, self.dist, self.distributed, self.eng, self.host_staged = torch, dist, distributed, eng, host_staged
        self.side = torch.cuda.Stream(device=eng.device)
        world = dist.get_world_size() if distributed else 1
        self.slots = [torch.zeros(5, dtype=torch.float64, device=eng.device) for _ in range(2)]
        self.gathered = [[torch.zeros(5, dtype=torch.float64, device=eng.device) for _ in range(world)] for _ in range(2)]
        self.host = [torch.zeros(5, dtype=torch.float64).pin_memory() for _ in range(2)] if host_staged else None
        self.events = [torch.cuda.Event() for _ in range(2)]
        self.turn = self.last = 0
        self.count = 0
        self.marked = False

    def mark(self):
        """The point of the launch stream the next submit() gathers at (default: where submit() itself is called)."""
        self.events[self.turn].record()
        self.marked = True

    def submit(self):
        torch = self.torch
        ready = self.events[self.turn]
        if not self.marked:
            ready.record()
        self.marked = False
        with torch.cuda.stream(self.side):
            self.side.wait_event(ready)
            self.slots[self.turn].copy_(self.eng.episode_stats, non_blocking=True)
            if self.host_staged:
                self.host[self.turn].copy_(self.slots[self.turn], non_blocking=True)
            elif self.distributed:
                self.dist.all_gather(self.gathered[self.turn], self.slots[self.turn])
            else:
                self.gathered[self.turn][0] = self.slots[self.turn]      # one rank: the all-gather is the identity (no second copy)
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


def measure_other_config(torch, device_index, spec, seconds, buffer_gib):
    """One entry of `other_configs`: a BASELINE.json configuration other than the headline, in its default flow (fused
    launches; Greedy vs Greedy restarts finished episodes after every 2nd launch, the random policy about every 128
    steps), timed for about `seconds` of back-to-back launches after an untimed pass and 0.25 s of clock warm-up."""
    from mate_amd.config import read_config
    from mate_amd.engine import Engine
    workload, batch, policy, label = spec[:4]
    eng = Engine(read_config(workload), batch, device=device_index, seed=0, first_env_index=0)
    b_alg = algorithmic_bytes(eng.num_cameras, eng.num_targets, eng.num_obstacles)
    b_obs = 4 * (eng.num_cameras * eng.camera_obs_dim + eng.num_targets * eng.target_obs_dim) + 48
    cap = int(buffer_gib * (1 << 30)) // (batch * b_obs)
    if len(spec) > 4:
        cap = min(cap, spec[4])
    if policy == 'greedy':
        eng.enable_policies()
        R, resets, fn, kernel = min(48, cap), 2, eng.rollout_greedy, 'rollout_greedy_kernel'
    else:
        R = next((r for r in (256, 128, 64, 32) if r <= cap), max(1, cap))
        resets, fn, kernel = max(1, 128 // R), eng.rollout_random, 'rollout_kernel'
    eng.reset()
    eng.reserve_rollout(R, search='deep')
    for _ in range(2 * resets):
        fn(R, auto_reset=resets)
    torch.cuda.synchronize()
    t0, n_warm = time.perf_counter(), 0
    while time.perf_counter() - t0 < 0.25:
        for _ in range(resets):
            fn(R, auto_reset=resets)
        n_warm += resets
        torch.cuda.synchronize()
    per_launch = (time.perf_counter() - t0) / n_warm
    launches = max(resets, int(seconds / per_launch) // resets * resets)
    eng.kernel_time(enable=1)
    torch.cuda.synchronize()
    idle0, t0 = eng.idle_steps(), time.perf_counter()
    for _ in range(launches):
        fn(R, auto_reset=resets)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    executed = batch * R * launches - (eng.idle_steps() - idle0)
    kernel_ms, timed = eng.kernel_time(enable=False)
    value = executed / elapsed
    out = {'config': label, 'workload': f'{workload} batch={batch} envs, {policy} policy, fused {R}-step launches, restarts every {resets} launch(es)',
           'value': value, 'unit': 'env-steps/s (executed: idle slots of finished episodes excluded)', 'seconds': elapsed, 'launches': launches,
           'kernel': kernel, 'kernel_avg_us': kernel_ms * 1e3, 'launches_timed': timed,
           'algorithmic_bytes_per_env_step': b_alg,
           'frac': (b_alg * batch * R / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if kernel_ms > 0 else 0.0,
           'end_to_end_frac': b_alg * value / 1e9 / HBM_PEAK_GBS}
    eng.close()
    del eng
    torch.cuda.empty_cache()
    return out


def measure_reset_amortised(torch, eng, cfg, value, seconds_per_step):
    """The cost the timed region never contains.  Under the random policy cargo never runs out, so every environment of the
    batch hits the time limit on the same step, every max_episode_steps + 1 steps, and the batch restarts in ONE whole-batch
    reset (placement, Nc occlusion tables per environment, first view).  Its time, measured here (median of 7, each
    bracketed by synchronisations), spread over an episode: value_with_resets = N / (t_step + t_reset / (max_episode_steps + 1))."""
    times = []
    for _ in range(8):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.reset()
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
    t_reset = sorted(times[1:])[len(times[1:]) // 2]
    episode = int(cfg['max_episode_steps']) + 1
    with_resets = eng.num_envs / (seconds_per_step + t_reset / episode)
    return {'whole_batch_reset_ms': t_reset * 1e3, 'episode_steps': episode, 'value_with_resets': with_resets, 'unit': 'env-steps/s',
            'cost_frac': 1.0 - with_resets / value if value > 0 else None,
            'note': 'value = N / t_step as timed; value_with_resets = N / (t_step + t_reset / episode_steps): one whole-batch reset per episode of the random-policy batch'}


def measure_n1_api(torch, steps=1500):
    """BASELINE config 1 (MATE-4v2-9.yaml, one environment, random actions, the reference's evaluate loop, mate/evaluate.py:85-167
    -> mate_amd/evaluate.py) on the N = 1 NumPy API: every step is a launch, a synchronisation and a PCIe copy of the observations
    and the state -- the compatibility path the reference's own wrappers use, not a throughput path."""
    import mate_amd
    from mate_amd.evaluate import evaluate, random_policy
    env = mate_amd.MultiAgentTracking('MATE-4v2-9.yaml', max_episode_steps=steps)
    env.seed(0)
    evaluate(env, random_policy(0))                  # untimed: code objects, allocations
    history = []
    t0 = time.perf_counter()
    evaluate(env, random_policy(1), history=history)
    elapsed = time.perf_counter() - t0
    out = {'workload': 'MATE-4v2-9.yaml, 1 environment, uniform random actions from NumPy, mate_amd.evaluate (reset + one episode)',
           'value': len(history) / elapsed, 'unit': 'env-steps/s', 'steps': len(history), 'seconds': elapsed,
           'reference_numpy': 1101.0, 'reference_note': 'mate/evaluate.py FPS of the reference on one core of the build container (BASELINE.md section 2)'}
    env.close()
    return out


def measure_hbm_copy_peak(torch, device_index, gib=1.0, reps=5):
    """The rate of a plain device-to-device copy ON THIS BOX, in this run: a 1 GiB tensor copied `reps` times, the median of the
    timed copies, read + write bytes per second (the figure `roofline.peak_measured` used to take from a constant measured on
    another box of the pool)."""
    n = int(gib * (1 << 30))
    with torch.cuda.device(device_index):
        a = torch.empty(n, dtype=torch.uint8, device='cuda')
        b = torch.empty_like(a)
        a.zero_()
        b.copy_(a)
        torch.cuda.synchronize()
        rates = []
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            b.copy_(a)
            e1.record()
            torch.cuda.synchronize()
            rates.append(2.0 * n / (e0.elapsed_time(e1) * 1e-3) / 1e9)
        del a, b
        torch.cuda.empty_cache()
    return sorted(rates)[len(rates) // 2]


def measure_hbm_fill_peak(torch, device_index, gib=1.0, reps=5):
    """... and of a plain fill (write-only, like the observation rows the dominant kernels stream out): 1 GiB zeroed `reps` times, median."""
    n = int(gib * (1 << 30)) // 4
    with torch.cuda.device(device_index):
        a = torch.empty(n, dtype=torch.float32, device='cuda')
        a.zero_()
        torch.cuda.synchronize()
        rates = []
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            a.zero_()
            e1.record()
            torch.cuda.synchronize()
            rates.append(4.0 * n / (e0.elapsed_time(e1) * 1e-3) / 1e9)
        del a
        torch.cuda.empty_cache()
    return sorted(rates)[len(rates) // 2]


def measure_learner_flows(torch, device_index, workload, batch, graph_steps, reset_interval, versus_reset_interval, world=1):
    """The flows a learner calls, one launch (or one launch + the learner's own kernel) per step, at `batch` environments:
      per_step_launch   step_random: the engine's step kernel back to back, no caller kernel in between
      external_actions  step(actions): the joint actions in a caller-owned f32 buffer that a stand-in policy kernel rewrites before
                        every step (mate/environment.py:590-676 behind a learner), `graph_steps` (policy, step) pairs per HIP graph
      versus_greedy     MultiCamera(GreedyTargetAgent) (mate/wrappers/single_team.py:245-264; every examples/*/camera/config.py): the
                        learner's stand-in policy kernel writes the camera team's joint action, the on-device greedy targets act and
                        the environment steps in ONE launch (step_greedy_kernel)
      versus_greedy_frameskip5     ... with FrameSkip(5) on top (one fused launch per learner action): the camera trainers' whole flow
      target_learner_frameskip10   the TARGET trainers' flow (examples/*/target/config.py): MATE-2v4-0, MultiTarget(GreedyCameraAgent),
                        FrameSkip(10) -- a small scenario (rows a fifth of MATE-4v8-9's: a wave's lanes are mostly idle)
      external_actions_two_groups   the same batch as two half-batch engines on two streams, graphs replayed alternately -- a learner
                        that alternates between two groups of environments (double-buffered sampling): one group's step runs under
                        the other group's policy and launch ramp
    each timed over whole graphs (median of three passes, all three listed), with one reset launch per `reset_interval` steps
    (`versus_reset_interval` against the greedy opponents, whose episodes end after ~1.2 k steps: a hundred of 4096 environments per
    32 steps, and their restart -- placement, occlusion tables, first view: four latency-bound launches -- costs 3.5 us per step at
    32, half of it at 64); idle slots of finished environments are excluded from `value`.  `kernel_avg_us` / `roofline_frac`: a separate pass of direct launches with a
    dispatch-event pair on every launch."""
    from mate_amd.config import read_config
    from mate_amd.engine import Engine
    cfg = read_config(workload)
    G = max(int(graph_steps), reset_interval)
    G -= G % reset_interval
    steps = max(G, (1024 if batch <= 16384 else 256) // G * G)
    out = {'batch': batch, 'workload': workload, 'steps': steps, 'reset_interval': reset_interval, 'graph_steps': G}

    def timed(run, idle, n_envs):
        run(2 * G)
        torch.cuda.synchronize()
        times = []
        for _ in range(3):
            i0, t0 = idle(), time.perf_counter()
            run(steps)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            times.append((dt, n_envs * steps - (idle() - i0)))
        passes[:] = [t[0] / steps * 1e6 for t in times]
        return sorted(times)[1]

    passes = []

    def entry(eng, dt, executed, kernel=None, km=0.0, flow=None):
        b_alg = algorithmic_bytes(eng.num_cameras, eng.num_targets, eng.num_obstacles)
        e = {'value': executed * world / dt, 'unit': 'env-steps/s', 'us_per_step': dt / steps * 1e6,
             'passes_us_per_step': [round(v, 3) for v in passes],
             'end_to_end_frac': b_alg * executed / dt / 1e9 / HBM_PEAK_GBS}
        if kernel:
            e.update({'kernel': kernel, 'kernel_avg_us': km * 1e3, 'roofline_frac': b_alg * batch / (km * 1e-3) / 1e9 / HBM_PEAK_GBS if km > 0 else 0.0})
        if flow:
            e['flow'] = flow
        return e

    with torch.cuda.device(device_index):
        # ---- step_random, direct launches (the host enqueues ahead of the GPU from 4096 environments on)
        eng = Engine(cfg, batch, device=device_index, seed=0)
        eng.reset()
        dt, ex = timed(lambda n: [eng.step_random(auto_reset=reset_interval) for _ in range(n)], eng.idle_steps, batch)
        eng.kernel_time(enable=1)
        for _ in range(256):
            eng.step_random(auto_reset=reset_interval)
        torch.cuda.synchronize()
        km, _ = eng.kernel_time(enable=False)
        out['per_step_launch'] = entry(eng, dt, ex, 'step_kernel', km)
        # ---- step(actions) from a HIP graph
        ext = ExternalActions(torch, eng, G, reset_interval)
        dt, ex = timed(ext.run, eng.idle_steps, batch)
        out['external_actions'] = entry(eng, dt, ex, flow=f'step(actions): f32 joint actions rewritten by a policy kernel in a caller-owned device buffer before every step; '
                                                              f'{ext.graph_steps} (policy kernel, step) pairs + one reset launch per {reset_interval} steps per HIP graph replay')
        ext.stepper.close()
        eng.close()
        del ext, eng
        torch.cuda.empty_cache()
        # ---- learner versus the on-device greedy opponents
        eng = Engine(cfg, batch, device=device_index, seed=0)
        if eng.num_cameras:
            eng.enable_policies()
            eng.reset()
            mine = (torch.rand((batch, eng.num_cameras, 2), device=eng.device) * 2 - 1) * torch.tensor([5.0, 2.5], device=eng.device)
            Gv = max(G, versus_reset_interval) // versus_reset_interval * versus_reset_interval
            st = eng.make_stepper(mine, None, auto_reset=versus_reset_interval, graph_steps=Gv, between=lambda: mine.mul_(-1.0), versus='camera')
            dt, ex = timed(st.run, eng.idle_steps, batch)
            st.close()
            eng.kernel_time(enable=1)
            for _ in range(256):
                eng.step_versus_greedy('camera', mine, auto_reset=versus_reset_interval)
            torch.cuda.synchronize()
            km, _ = eng.kernel_time(enable=False)
            out['versus_greedy'] = entry(eng, dt, ex, 'step_greedy_kernel' if eng.last_flow == 4 else 'rollout_greedy_kernel (one step)', km,
                                         flow='MultiCamera(GreedyTargetAgent): the learner\'s stand-in policy (one elementwise kernel) writes the camera team\'s joint action, '
                                              'the greedy targets act and the environment steps in one launch; executed env-steps (idle slots of finished episodes excluded); '
                                              f'one restart of the finished environments per {versus_reset_interval} steps')
            out['versus_greedy']['reset_interval'] = versus_reset_interval
            del st
            # ---- ... and with FrameSkip(5) on top, what every example trainer's make_env ends with (examples/ippo/camera/config.py:
            # frame_skip = 5; examples/utils/wrappers.py:301-323): ONE launch per learner action (rollout_versus_greedy), replayed from
            # a HIP graph like the per-step flows (Stepper(frame_skip=K): the device-resident step counter advances by K per launch)
            K = 5
            per = max(1, versus_reset_interval // K)               # launches per reset interval
            Gs = max(per, (G // K) // per * per)                   # launches per graph
            launches = max(Gs, steps // K // Gs * Gs)
            st = eng.make_stepper(mine, None, auto_reset=per, graph_steps=Gs, between=lambda: mine.mul_(-1.0), versus='camera', frame_skip=K)
            st.run(2 * Gs)
            torch.cuda.synchronize()
            times = []
            for _ in range(3):
                i0, t0 = eng.idle_steps(), time.perf_counter()
                st.run(launches)
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
                times.append((dt, batch * launches * K - (eng.idle_steps() - i0)))
            dt, ex = sorted(times)[1]
            st.close()
            del st
            eng.kernel_time(enable=1)
            for _ in range(4 * per):
                eng.rollout_versus_greedy('camera', mine, K, auto_reset=per)
            torch.cuda.synchronize()
            km, _ = eng.kernel_time(enable=False)
            b_alg = algorithmic_bytes(eng.num_cameras, eng.num_targets, eng.num_obstacles)
            out['versus_greedy_frameskip5'] = {
                'value': ex * world / dt, 'unit': 'env-steps/s', 'us_per_step': dt / (launches * K) * 1e6, 'us_per_launch': dt / launches * 1e6,
                'passes_us_per_step': [round(t[0] / (launches * K) * 1e6, 3) for t in times],
                'end_to_end_frac': b_alg * ex / dt / 1e9 / HBM_PEAK_GBS, 'kernel': 'rollout_greedy_kernel', 'kernel_avg_us': km * 1e3,
                'roofline_frac': b_alg * batch * K / (km * 1e-3) / 1e9 / HBM_PEAK_GBS if km > 0 else 0.0,
                'flow': f'FrameSkip({K}) over MultiCamera(GreedyTargetAgent): one policy kernel and ONE fused launch per learner action ({K} frames, the greedy targets '
                        f'act anew on every frame), {Gs} (policy kernel, launch) pairs per HIP graph replay, one restart of the finished environments per {per} launches; '
                        'executed env-steps'}
            del mine
        eng.close()
        del eng
        torch.cuda.empty_cache()
        # ---- the TARGET learner's flow of the example trainers (examples/ippo/target/config.py:20-67 and its siblings): MATE-2v4-0,
        # MultiTarget(GreedyCameraAgent), FrameSkip(10) -- one policy kernel and one ten-frame launch per learner action, from a HIP graph
        if workload == 'MATE-4v8-9.yaml':
            cfg_t = read_config('MATE-2v4-0.yaml')
            eng = Engine(cfg_t, batch, device=device_index, seed=0)
            eng.enable_policies()
            eng.reset()
            K = 10
            per = max(1, versus_reset_interval // K)
            Gs = max(per, (G // K) // per * per)
            launches = max(Gs, steps // K // Gs * Gs)
            eng.reserve_rollout(K, search='none')
            mine = (torch.rand((batch, eng.num_targets, 2), device=eng.device) * 2 - 1) * 10.0
            st = eng.make_stepper(None, mine, auto_reset=per, graph_steps=Gs, between=lambda: mine.mul_(-1.0), versus='target', frame_skip=K)
            st.run(2 * Gs)
            torch.cuda.synchronize()
            times = []
            for _ in range(3):
                i0, t0 = eng.idle_steps(), time.perf_counter()
                st.run(launches)
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
                times.append((dt, batch * launches * K - (eng.idle_steps() - i0)))
            dt, ex = sorted(times)[1]
            st.close()
            del st
            eng.kernel_time(enable=1)
            for _ in range(4 * per):
                eng.rollout_versus_greedy('target', mine, K, auto_reset=per)
            torch.cuda.synchronize()
            km, _ = eng.kernel_time(enable=False)
            b_alg = algorithmic_bytes(eng.num_cameras, eng.num_targets, eng.num_obstacles)
            out['target_learner_frameskip10'] = {
                'workload': 'MATE-2v4-0.yaml', 'value': ex * world / dt, 'unit': 'env-steps/s', 'us_per_step': dt / (launches * K) * 1e6, 'us_per_launch': dt / launches * 1e6,
                'passes_us_per_step': [round(t[0] / (launches * K) * 1e6, 3) for t in times], 'algorithmic_bytes_per_env_step': b_alg,
                'end_to_end_frac': b_alg * ex / dt / 1e9 / HBM_PEAK_GBS, 'kernel': 'rollout_greedy_kernel', 'kernel_avg_us': km * 1e3,
                'roofline_frac': b_alg * batch * K / (km * 1e-3) / 1e9 / HBM_PEAK_GBS if km > 0 else 0.0,
                'flow': f'FrameSkip({K}) over MultiTarget(GreedyCameraAgent) on MATE-2v4-0 (the target trainers\' scenario): one policy kernel and ONE fused launch per learner '
                        f'action ({K} frames, the greedy cameras act anew on every frame), {Gs} (policy kernel, launch) pairs per HIP graph replay, one restart of the finished '
                        f'environments per {per} launches; executed env-steps; rows of this scenario are a fifth of MATE-4v8-9\'s'}
            del mine
            eng.close()
            del eng
            torch.cuda.empty_cache()
        # ---- two half-batch groups on two streams (mate_amd.engine.EngineGroups): step(actions), and the learner versus the greedy opponents
        if batch % 2 == 0 and batch >= 2048:
            from mate_amd.engine import EngineGroups
            half = batch // 2
            for key, versus in (('external_actions_two_groups', False), ('versus_greedy_two_groups', True)):
                if versus and 'versus_greedy' not in out:      # (a scenario without cameras)
                    continue
                interval = versus_reset_interval if versus else reset_interval
                Gk = max(G, interval) // interval * interval
                groups = EngineGroups(cfg, batch, groups=2, device=device_index, seed=0, policies=versus)
                groups.reset()
                keep = []

                def make(g, e):
                    if versus:
                        mine = (torch.rand((half, e.num_cameras, 2), device=e.device) * 2 - 1) * torch.tensor([5.0, 2.5], device=e.device)
                        keep.append(mine)
                        return e.make_stepper(mine, None, auto_reset=interval, graph_steps=Gk, between=(lambda m=mine: m.mul_(-1.0)), versus='camera')
                    ext = ExternalActions(torch, e, Gk, interval)
                    keep.append(ext)
                    return ext.stepper

                steppers = groups.each(make)
                torch.cuda.synchronize()
                # (HIP maps streams onto a handful of hardware queues, and two streams that share one run their graphs one after the
                # other -- 23 instead of 13 us per step at 4096 when that happens: the second group's stream is chosen by a short trial)
                trials = groups.pick_streams(lambda g, e: steppers[g].run(Gk), candidates=3, warm=2, timed=4)
                dt, ex = timed(lambda n: [groups.each(lambda g, e: steppers[g].run(Gk)) for _ in range(n // Gk)], groups.idle_steps, batch)
                out[key] = entry(groups.engines[0], dt, ex, flow=f'{"MultiCamera(GreedyTargetAgent)" if versus else "step(actions)"} as two engines of {half} environments (global indices 0.. and {half}..) on two streams '
                                                                   '(mate_amd.engine.EngineGroups), their HIP graphs replayed alternately: us_per_step = per step of the WHOLE batch')
                out[key]['stream_trials_us_per_step'] = [round(t / (4 * Gk) * 1e6, 2) for t in trials]
                out[key]['reset_interval'] = interval
                for st in steppers:
                    st.close()
                groups.close()
                del steppers, groups, keep
                torch.cuda.empty_cache()
    return out


OTHER_CONFIGS = (
    ('MATE-8v8-9.yaml', 8192, 'greedy', 'BASELINE config 3'),
    ('MATE-4v8-0.yaml', 8192, 'random', 'BASELINE config 4, the shard of one of its 8 GPUs'),
    ('MATE-Navigation.yaml', 4096, 'random', 'BASELINE config 5, the shard of one of its 8 GPUs'),
    # the same two configurations WHOLE on this one GPU (sixteen / eight generations of resident waves; launches of 64 steps: 19 / 17 GB of rows)
    ('MATE-4v8-0.yaml', 65536, 'random', 'BASELINE config 4, its whole batch on ONE GPU', 64),
    ('MATE-Navigation.yaml', 32768, 'random', 'BASELINE config 5, its whole batch on ONE GPU', 64),
)


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
        print(json.dumps({'metric': 'dry-run (no GPU work)', 'value': executed / elapsed, 'unit': 'env-steps/s', 'n_gpus': world, 'steps': args.steps,
                          'warmup': args.warmup, 'ms_per_step': elapsed / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'weak',
                          'vs_baseline': None, 'dtype': 'f64', 'data': 'dry-run', 'config': {'workload': 'none', 'global_batch': args.batch * world},
                          'ranks': world, 'shard_first_env_mean': float(stats[1])}), flush=True)
    if world > 1:
        dist.destroy_process_group()


def main():
    t_main = time.perf_counter()
    args = parse_args()
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        raise SystemExit(launch_ranks(args))
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
        external = ExternalActions(torch, eng, args.graph_steps, args.step_reset_interval)
        step = external.step
    else:
        step = lambda: eng.step_random(auto_reset=args.step_reset_interval)     # noqa: E731
    rollout_fn = eng.rollout_greedy if args.policy == 'greedy' else eng.rollout_random
    # default restart cadence of the fused flows: Greedy vs Greedy every 2 launches (k = 2..4 are within 1.5 %); random policy about every 128
    # steps whatever the launch length (after each 128-step launch; after every 6th 20-step launch): a finished environment then idles
    # ~64 of its 10^4 steps (idle slots are not counted in `value`) and short launches do not each drag an idle reset launch behind them
    rollout_resets = args.rollout_reset_interval if args.rollout_reset_interval > 0 else (2 if args.policy == 'greedy' else max(1, 128 // max(R, 1)))
    rollout = lambda n, auto_reset=True: rollout_fn(n, auto_reset=rollout_resets)     # noqa: E731
    gather = StatsGather(torch, dist, distributed, eng, host_staged=host_staged) if args.stats_interval > 0 else None

    gather_inside = [False]      # a short region's one gather: its marker BEHIND the region's last launch (it carries the region's own episodes) instead of ahead of the region

    def run(steps, timed=False):
        """exactly `steps` env.step()s of the whole batch"""
        if external is not None:
            external.run(steps)
            if timed and gather is not None:
                gather.submit()
        elif R > 0:
            lengths = [R] * (steps // R) + ([steps % R] if steps % R else [])
            # a gather every `stats_interval` launches: enqueued behind a launch, it runs on the side stream under the launches
            # that follow.  A region that holds fewer launches than the interval gathers once, AHEAD of its first launch (the
            # episodes finished so far: the previous repetitions'), so that it too runs under a launch instead of behind the
            # last one, where its copy + collective would sit between the kernel's end and the region's closing synchronise
            # (15 us of a 200 us region at the driver's `--steps 20`)
            short = timed and gather is not None and len(lengths) < args.stats_interval
            if short and not gather.marked and not gather_inside[0]:
                gather.mark()            # (an event record: the copy + collective are enqueued behind the first launch, below,
            for i, n in enumerate(lengths):          # so that the host prepares them while the GPU already runs it)
                rollout(n, auto_reset=True)
                if timed and gather is not None and ((i + 1) % args.stats_interval == 0 or (short and i == (len(lengths) - 1 if gather_inside[0] else 0))):
                    gather.submit()
        else:
            every = args.stats_interval * 128
            short = timed and gather is not None and steps < every
            if short and not gather.marked:
                gather.mark()
            for i in range(steps):
                step()
                if timed and gather is not None and ((i + 1) % every == 0 or (short and i == 0)):
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
    inside_ms, inside_executed = [], []      # the same repetitions once more with the short region's gather marker INSIDE the region (see gather_inside)
    n_main = max(0, args.rep_warmup) + max(1, args.reps)
    n_inside = max(1, args.reps) if (gather is not None and short_region and R > 0 and not args.dump) else 0
    for rep_index in range(n_main + n_inside):
        rep = rep_index if rep_index < n_main else n_main       # (the second pass needs no warm-up repetitions of its own)
        gather_inside[0] = rep_index >= n_main
        eng.kernel_time(enable=1 if R > 0 else STEP_SAMPLE)   # HIP-event pair around every launch of the dominant kernel (every 17th one-step launch)
        # a region shorter than the gather interval gathers the episodes finished BEFORE it (see run): the marker of what that
        # gather may read is recorded here, behind the previous repetition's last launch -- an event record in front of the
        # region's only launch delays it by 4 us (tools/region_probe.py); the gather itself is enqueued inside the region
        if gather is not None and short_region and not gather_inside[0]:
            gather.mark()
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
        if gather_inside[0]:
            kernel_times.pop()
            inside_ms.append(elapsed * 1e3)
            inside_executed.append(executed)
            continue
        rep_ms.append(elapsed * 1e3)
        rep_executed.append(executed)
    gather_inside[0] = False
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
    startup_line['note'] = ('seconds per rank; startup_s = main() entry to the first pass through the measuring loop (imports, process group, engine, reset, '
                            'reserve_rollout with its candidate search, warm-up); reserve_rollout_s is bounded by MATE_BLOCK_SECONDS / MATE_BLOCK_GIB')

    if args.dump:      # every rank: the final state and the last outputs of its shard (compared across shardings by the tests)
        torch.cuda.synchronize()
        torch.save({'rank': rank, 'world': world, 'first_env_index': rank * args.batch, 'state': eng.export_state().cpu(),
                    'scalars': eng.scalars.cpu(), 'episode_stats': eng.episode_stats.cpu(), 'idle_steps': eng.idle_steps(),
                    'last_rollout_scalars': (eng._rollout['scalars'].cpu() if getattr(eng, '_rollout', None) else None)},
                   f'{args.dump}.rank{rank}.pt')
    if rank == 0:
        copy_peak = measure_hbm_copy_peak(torch, local_rank) if not args.dump else HBM_PEAK_MEASURED_GBS
        fill_peak = measure_hbm_fill_peak(torch, local_rank) if not args.dump else None
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
        headline_case = args.batch == BATCH_PER_GPU and args.workload == WORKLOAD and args.policy != 'greedy'
        policy_text = {'random': 'uniform random policy (on-device Philox), ', 'greedy': 'on-device GreedyCamera vs GreedyTarget policies, ',
                       'external': 'joint actions read from a caller-owned device buffer (learner in the loop), '}[args.policy]
        line = {
            'metric': f'env-steps/sec {args.workload[:-5]} batch={args.batch} per GPU ({args.policy} policy, auto-reset)',
            'value': value, 'unit': 'env-steps/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': elapsed / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
            'reps': len(rep_ms), 'rep_ms': [round(v, 4) for v in rep_ms], 'timing': 'median repetition of the K-step timed region',
            'warmup_extra_steps': extra_steps, 'rep_warmup': max(0, args.rep_warmup),
            'config': {'workload': f'{args.workload} batch={args.batch} envs per GPU, ' + policy_text
                                   + (f'fused {R}-step rollout launches, auto-reset after ' + ('each launch' if rollout_resets == 1 else f'every {rollout_resets} launches')
                                      if R > 0 else 'one launch per step, auto-reset'),
                       'global_batch': total_envs, 'parallelism': f'env-shard x{world}', 'steps_per_launch': R if R > 0 else 1},
            'roofline': {
                'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                'frac': achieved / HBM_PEAK_GBS,
                'traffic': measured_traffic(kernel, steps_per_launch, args.batch) if (headline_case and float(steps_per_launch).is_integer()) else None,
                'traffic_unit': 'bytes per launch (rocprofv3 2*FETCH_SIZE + WRITE_SIZE of this kernel at this launch length)',
                'traffic_source': 'profiles/latest_pmc.json: PMC passes of the builder on a box of the same pool (tools/pmc_collect.py), NOT collected in this run',
                'peak_measured': copy_peak, 'frac_of_measured_peak': achieved / copy_peak,
                'peak_measured_source': 'device-to-device copy of 1 GiB on this box in this run, median of 5 (read + write bytes)',
                'fill_measured': fill_peak, 'fill_measured_source': 'torch zero_() of 1 GiB on this box in this run, median of 5 (write-only, like the rows this kernel streams out)',
                'kernel': '%s<float, %s, %s>' % (kernel, 'FixedShape' if eng.specialised else 'AnyShape', ('FLOW_ANY', 'FLOW_RANDOM', 'FLOW_ACT_F32', 'FLOW_GREEDY')[flow]),
                'kernel_avg_us': kernel_ms * 1e3, 'launches_timed': launches, 'env_steps_per_launch': args.batch * steps_per_launch,
                'algorithmic_bytes_per_launch': bytes_per_launch,
                'resident_bytes_per_launch': resident,
                'achieved_resident': resident / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0,
                'frac_resident': resident / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if kernel_ms > 0 else 0.0,
                'end_to_end_frac': b_alg * value / world / 1e9 / HBM_PEAK_GBS,
                # where the observation blocks lie decides up to a fifth of a fused launch (the stores): the store
                # rates [GB/s] of the candidates Engine.reserve_rollout probed for the camera and the target block (it kept the fastest)
                'observation_block_candidates_gbs': [[round(x) for x in r] for r in getattr(eng, 'block_rates', [])],
            },
            'episode_stats': {'mean_target_reward': float(stats[0]), 'mean_coverage_rate': float(stats[1]),
                              'mean_delivered': float(stats[2]), 'gathered_in_loop': gathered_stats},
        }
        line['startup'] = startup_line
        line['episode_stats']['stats_gather'] = (
            'none' if gather is None else
            'region shorter than the gather interval: ONE gather per repetition, copy + collective enqueued on the side stream inside the region behind its '
            'first launch; the marker of what it may read is recorded BEFORE the region (behind the previous repetition), so it carries the episodes '
            'finished before this region, not its own' if short_region else
            f'every {args.stats_interval} launches inside the region, each behind the launch whose episodes it carries')
        if inside_ms:      # the same region under the other definition, so that both stand in one line
            order_i = sorted(range(len(inside_ms)), key=lambda i: inside_ms[i])
            mid_i = order_i[len(order_i) // 2]
            line['gather_marker'] = {
                'before': {'value': value, 'ms_per_step': elapsed / args.steps * 1e3},
                'inside': {'value': inside_executed[mid_i] / (inside_ms[mid_i] * 1e-3), 'ms_per_step': inside_ms[mid_i] / args.steps,
                           'rep_ms': [round(v, 4) for v in inside_ms]},
                'note': '`value` is the `before` form: the one statistics gather of a region shorter than the gather interval reads what finished BEFORE the region '
                        '(marker recorded behind the previous repetition, copy + collective under the region\'s launch).  `inside`: the marker behind the region\'s last launch '
                        '-- the gather carries the region\'s own episodes, and its copy + collective sit between the kernel\'s end and the closing synchronise'}
        line['config']['backend'] = ('gloo (ranks may share a GPU; statistics staged through the host)' if host_staged else 'nccl (RCCL)') if distributed else 'single process'
        line['config']['collectives'] = ('forced on one rank: barrier, side-stream all_gather, job reduction' if args.force_collectives and world == 1
                                         else 'barrier, side-stream all_gather, job reduction' if distributed else 'none (one rank)')
        default_case = world == 1 and args.policy == 'random' and args.workload == WORKLOAD and args.batch == BATCH_PER_GPU
        if default_case and not args.no_side_measurements and not args.dump:
            line['reset_amortised'] = measure_reset_amortised(torch, eng, cfg, value, elapsed / args.steps)
        if default_case and not args.no_other_configs and not args.dump:
            # the other BASELINE configurations that fit one GPU, about a second each (the headline engine's buffers are released first)
            eng.close()
            eng._rollout = None
            torch.cuda.empty_cache()
            line['other_configs'] = [measure_other_config(torch, local_rank, spec, args.other_seconds, args.buffer_gib) for spec in OTHER_CONFIGS]
        if world == 1 and not args.no_extras and args.policy == 'random' and R > 0 and not args.dump:
            # the learner-facing per-step flows, at this run's batch and -- the default line -- at the two larger
            # batches a learner on one MI355X runs; the headline batch's entries also stand at the top level of the line
            eng.close()
            eng._rollout = None
            torch.cuda.empty_cache()
            batches = [args.batch] + ([b for b in LEARNER_BATCHES if b != args.batch] if default_case else [])
            flows = [measure_learner_flows(torch, local_rank, args.workload, b, args.graph_steps, args.step_reset_interval, args.versus_reset_interval) for b in batches]
            line['learner_flows'] = flows
            # next to the fractions of the vendor peak: of the copy rate measured on this box in this run (what a kernel that reads
            # and writes HBM can reach here: 4.6-5.3 TB/s on this pool) -- the per-step flows at 65 536 environments run at it
            for fl in flows:
                for entry in fl.values():
                    if isinstance(entry, dict) and 'end_to_end_frac' in entry:
                        entry['end_to_end_frac_of_measured_copy_peak'] = entry['end_to_end_frac'] * HBM_PEAK_GBS / copy_peak
            for key in ('per_step_launch', 'external_actions', 'versus_greedy'):
                if key in flows[0]:
                    line[key] = dict({'reset_interval': args.step_reset_interval}, **flows[0][key], batch=args.batch)
        if default_case and not args.no_side_measurements and not args.dump:
            line['n1_api'] = measure_n1_api(torch)
        if not args.no_cpu_baseline and world == 1 and args.policy == 'random' and args.workload == WORKLOAD:
            line['cpu_baseline'] = cpu_baseline(args.cpu_seconds)
        print(json.dumps(line), flush=True)
    if distributed:
        dist.destroy_process_group()


class ExternalActions:
    """The learner-in-the-loop flow: `step((camera_actions, target_actions))` with the joint actions in caller-owned
    device buffers that a policy rewrites between steps.  Here the "policy" is one elementwise torch kernel per team
    that refreshes the buffers in place (a stand-in for a network's output layer), so the environment really consumes
    new externally produced actions at every step."""

    def __init__(self, torch, eng, graph_steps, reset_interval=1):
        self.torch, self.eng, self.graph_steps = torch, eng, int(graph_steps)
        self.graph_steps -= self.graph_steps % max(1, reset_interval)      # a graph holds whole reset intervals
        N, Nc, Nt = eng.num_envs, eng.num_cameras, eng.num_targets
        gen = torch.Generator(device=eng.device)
        gen.manual_seed(1234)
        self.flat = torch.rand(N * (Nc + Nt) * 2, device=eng.device, generator=gen) * 2 - 1      # one buffer, two views
        self.cam = self.flat[:N * Nc * 2].view(N, Nc, 2)
        self.tgt = self.flat[N * Nc * 2:].view(N, Nt, 2)
        self.cam.mul_(torch.tensor([5.0, 2.5], device=eng.device))
        self.tgt.mul_(20.0)
        self.stepper = eng.make_stepper(self.cam, self.tgt, auto_reset=max(1, reset_interval), graph_steps=self.graph_steps, between=self.policy)

    def policy(self):
        # a new joint action every step, produced on the device by "someone else's" kernel
        self.flat.mul_(-1.0)

    def step(self):
        self.stepper.run(1)

    def run(self, steps):
        self.stepper.run(steps)


if __name__ == '__main__':
    main()
