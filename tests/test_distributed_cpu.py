"""world_size-2 gloo test of the only collective on the path: the episode-statistics all-gather."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from mate_amd.distributed import EpisodeStats, gather_episode_stats, shard_of
    first, count = shard_of(10, rank, world)
    stats = EpisodeStats()
    env_ids = torch.arange(first, first + count, dtype=torch.float64)
    done = (env_ids % 2 == 0)
    stats.update(done, episode_return=env_ids * 10, episode_length=env_ids + 100, coverage_rate=env_ids / 10, delivered=env_ids)
    total, parts = gather_episode_stats(stats)
    # the benchmark's job-level reduction: slowest rank's time, all ranks' env-steps, averaged statistics
    from mate_amd.distributed import reduce_job
    job = reduce_job(1.0 + rank, 1000.0 * (rank + 1), torch.tensor([float(rank), 10.0 * rank, 1.0]))
    if rank == 0:
        out.put((total, [p.tolist() for p in parts], (job[0], job[1], job[2].tolist())))
    dist.destroy_process_group()


def test_gather_episode_stats_two_ranks():
    ctx = mp.get_context('spawn')
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    total, parts, job = out.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # even environment indices 0,2,4,6,8 finished: shard 0 holds 0..4, shard 1 holds 5..9
    assert total['episodes'] == 5
    assert total['mean_return'] == pytest.approx((0 + 20 + 40 + 60 + 80) / 5)
    assert total['mean_length'] == pytest.approx(100 + 4)
    assert parts[0][0] == 3 and parts[1][0] == 2
    assert job[0] == 2.0 and job[1] == 3000.0 and job[2] == [0.5, 5.0, 1.0]


def test_bench_gpus_flag_launches_that_many_ranks():
    """`python bench.py --gpus 2` outside torchrun starts two ranks itself (torch.distributed.run child, rendezvous on
    127.0.0.1) and the line rank 0 prints carries the process group's world size; `--dry-run` swaps RCCL for gloo and
    the engine for fabricated timings, so the launch path and the job-level reduction run on CPU."""
    import json
    import subprocess
    import tempfile
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    details = os.path.join(tempfile.mkdtemp(), 'details.json')
    done = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--dry-run', '--steps', '10', '--warmup', '1', '--details', details],
                          env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True, timeout=300)
    assert done.returncode == 0, done.stderr[-2000:]
    printed = [l for l in done.stdout.splitlines() if l.startswith('{')]
    assert len(printed) == 1                      # rank 0 only (a dry run has no side measurements, so no `side` line)
    # what the driver reads: the LAST stdout line, a JSON object of at most 3000 bytes (it keeps ~8 KB of stdout; round 5's 22 KB line lost its head)
    assert done.stdout.rstrip().splitlines()[-1] == printed[0] and len(printed[0]) <= 3000
    line = json.loads(printed[0])
    assert line['n_gpus'] == 2 and line['data'] == 'dry-run' and line['details_file'] == details
    assert line['config']['global_batch'] == 2 * 4096
    # MAX over ranks of the (fabricated) times 1.0 s and 1.25 s, SUM of the env-steps of both shards
    assert line['ms_per_step'] == pytest.approx(1250.0 / 10) and line['value'] == pytest.approx(2 * 4096 * 10 / 1.25)
    with open(details) as fh:
        full = json.load(fh)                      # the full record: everything the printed line leaves out
    assert full['ranks'] == 2 and full['shard_first_env_mean'] == pytest.approx(2048.0) and full['value'] == line['value']
    # a rank count that contradicts the torchrun environment is refused, not silently benchmarked on one GPU
    env_bad = dict(env, WORLD_SIZE='1', RANK='0', LOCAL_RANK='0')
    bad = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--dry-run'], env=env_bad,
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True, timeout=120)
    assert bad.returncode != 0 and 'WORLD_SIZE=1' in bad.stderr


def _fabricated_full_record(n_flows=3):
    """A record shaped like a default N = 1 run of bench.py with every side measurement present and strings as long as the real ones."""
    flow = {'value': 123456789.123456, 'unit': 'env-steps/s', 'us_per_step': 12.3456789, 'passes_us_per_step': [12.345, 12.346, 12.347],
            'end_to_end_frac': 0.123456789, 'kernel': 'step_greedy_kernel', 'kernel_avg_us': 17.123456, 'roofline_frac': 0.23456789, 'flow': 'x' * 400}
    names = ('per_step_launch', 'external_actions', 'versus_greedy', 'versus_greedy_frameskip5', 'target_learner_frameskip10',
             'external_actions_two_groups', 'versus_greedy_two_groups')
    return {
        'metric': 'env-steps/sec MATE-4v8-9 batch=4096 per GPU (random policy, auto-reset)', 'value': 538123456.789, 'unit': 'env-steps/s', 'n_gpus': 1,
        'steps': 20, 'warmup': 5, 'ms_per_step': 0.00761234567, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
        'config': {'workload': 'MATE-4v8-9.yaml batch=4096 envs per GPU, uniform random policy (on-device Philox), fused 20-step launches, restarts every 6 launch(es)',
                   'global_batch': 4096, 'parallelism': 'env-shard x1', 'steps_per_launch': 20, 'backend': 'single process', 'collectives': 'none (one rank)'},
        'roofline': {'bound': 'hbm', 'achieved': 4861.123456, 'peak': 8000.0, 'unit': 'GB/s', 'frac': 0.6076404, 'traffic': 529612345.0, 'frac_traffic': 0.5234567,
                     'kernel': 'rollout_kernel<float, FixedShape, FLOW_RANDOM>', 'kernel_avg_us': 126.481234, 'launches_timed': 21, 'env_steps_per_launch': 81920.0,
                     'algorithmic_bytes_per_env_step': 7504, 'end_to_end_frac': 0.5047123, 'peak_measured': 6789.1234, 'copy_gbs': 5012.345, 'fill_gbs': 6789.1234,
                     'read_gbs': 4567.89, 'fused_note': 'fused launch: state, geometry and actions of the 8d per-step bytes stay in LDS; frac_traffic is the HBM utilisation'},
        'timing': {'reps': 21, 'rep_ms': [0.1523] * 21, 'rep_warmup': 3, 'warmup_extra_steps': 32100, 'definition': 'y' * 200},
        'episode_stats': {'mean_target_reward': 0.1, 'mean_coverage_rate': 0.2, 'mean_delivered': 0.3, 'gathered_in_loop': {'gathers_in_timed_loops': 21, 'episodes_finished': 4096.0}},
        'startup': {'startup_s': [12.345]},
        'reset_amortised': {'whole_batch_reset_ms': 12.3456, 'episode_steps': 10001, 'value_with_resets': 5.3e8, 'unit': 'env-steps/s', 'cost_frac': 0.012345, 'note': 'z' * 300},
        'other_configs': [{'config': 'BASELINE config %d, its whole batch on ONE GPU' % i, 'workload': 'w' * 150, 'value': 2.3456789e8, 'unit': 'u' * 60, 'seconds': 1.01,
                           'launches': 100, 'kernel': 'rollout_greedy_kernel', 'kernel_avg_us': 1405.6789, 'launches_timed': 100, 'algorithmic_bytes_per_env_step': 11568,
                           'frac': 0.404123, 'end_to_end_frac': 0.341234} for i in range(5)],
        'learner_flows': [dict({'batch': 4096 << (2 * i), 'workload': 'MATE-4v8-9.yaml', 'steps': 1024, 'reset_interval': 32, 'graph_steps': 64}, **{n: dict(flow) for n in names})
                          for i in range(n_flows)],
        'n1_api': {'workload': 'v' * 120, 'value': 2345.678, 'unit': 'env-steps/s', 'steps': 1500, 'seconds': 0.64, 'reference_numpy': 1101.0, 'reference_note': 'r' * 120},
        'cpu_baseline': {'value': 1234567.89, 'unit': 'env-steps/s', 'cores': 128, 'kind': 'port',
                         'sample': 'MATE-4v8-9.yaml batch=4096 x 3012 steps, random policy + f32 observation pack, OpenMP over envs on 128 threads '
                                   '(fastest of [8, 16, 32, 64, 128, 192]); single thread: batch=256 x 123 steps',
                         'single_thread': 12345.678, 'host_cpus': 192, 'cpu_model': 'AMD EPYC 9575F 64-Core Processor', 'thread_scan': {'8': 1, '16': 2},
                         'reference_numpy_per_core': 391.0, 'reference_note': 'n' * 200},
    }


def test_bench_prints_a_headline_the_driver_can_keep(tmp_path, capsys):
    """The driver keeps ~8 KB of stdout and parses the LAST line: bench.py's headline must be one JSON object of at most 3000 bytes with
    the contract's keys, `roofline` and `cpu_baseline` in it, whatever the side measurements produced; they go to the details file in
    full and to one `side` line of at most 3500 bytes before the headline (round 5 printed 22 KB on one line: BENCH_r05.parsed == null)."""
    import json
    sys.path.insert(0, ROOT)
    import bench
    full = _fabricated_full_record()
    details = str(tmp_path / 'details.json')
    bench.emit(full, details)
    out = capsys.readouterr().out.rstrip().splitlines()
    assert len(out) == 2 and all(len(l) <= 3500 for l in out) and len(out[-1]) <= bench.LINE_LIMIT == 3000
    assert len('\n'.join(out)) < 7000                                  # both lines inside the driver's tail
    line, side = json.loads(out[-1]), json.loads(out[0])['side']
    for key in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype', 'data',
                'config', 'roofline', 'cpu_baseline', 'details_file'):
        assert key in line, key
    assert line['value'] == full['value'] and line['ms_per_step'] == full['ms_per_step'] and line['vs_baseline'] is None      # full precision where the driver checks
    assert set(line['config']) == {'workload', 'global_batch', 'parallelism', 'steps_per_launch'}
    for key in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'frac_traffic', 'kernel', 'kernel_avg_us', 'peak_measured'):
        assert key in line['roofline'], key
    for key in ('value', 'unit', 'cores', 'kind', 'sample', 'single_thread', 'cpu_model', 'reference_numpy_per_core'):
        assert key in line['cpu_baseline'], key
    # no fraction of a "measured peak" above 1 can appear: the line holds none but fractions of the vendor peak
    assert all(v <= 1.0 for k, v in line['roofline'].items() if k.startswith('frac') or k.endswith('_frac'))
    assert len(side['other_configs']) == 5 and side['other_configs'][0]['end_to_end_frac'] == pytest.approx(0.341, abs=1e-3)
    assert [f['batch'] for f in side['learner_flows']] == [4096, 16384, 65536] and side['learner_flows'][1]['target_learner_frameskip10']['roofline_frac'] > 0
    with open(details) as fh:
        assert json.load(fh) == json.loads(json.dumps(full))           # nothing lost
    # a side summary that would not fit gives way; the headline never does
    bench.emit(_fabricated_full_record(n_flows=12), details)
    out = capsys.readouterr().out.rstrip().splitlines()
    assert len(out[-1]) <= 3000 and json.loads(out[0]) == {'side': 'see details_file'}


def test_reduce_job_runs_its_collectives_on_one_rank_when_forced():
    """`bench.py --force-collectives` (the one-GPU rehearsal of the N-rank path) needs the job-level reduction to really call
    all_reduce / all_gather on a world of one: gloo here, RCCL in tests/test_gpu_multirank.py."""
    import torch
    import torch.distributed as dist
    from mate_amd.distributed import reduce_job
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    dist.init_process_group('gloo', init_method=f'tcp://127.0.0.1:{port}', rank=0, world_size=1)
    try:
        calls = []
        real = dist.all_reduce
        dist.all_reduce = lambda *a, **k: (calls.append('all_reduce'), real(*a, **k))[1]
        try:
            stats = torch.tensor([1.0, 2.0, 3.0], dtype=torch.float64)
            assert reduce_job(0.5, 100.0, stats) == (0.5, 100.0, stats) and not calls          # identity on one rank ...
            elapsed, executed, out = reduce_job(0.5, 100.0, stats, force=True)                  # ... unless forced
            assert calls == ['all_reduce', 'all_reduce'] and (elapsed, executed) == (0.5, 100.0) and torch.equal(out, stats)
        finally:
            dist.all_reduce = real
    finally:
        dist.destroy_process_group()
