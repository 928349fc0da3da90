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
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    done = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--dry-run', '--steps', '10', '--warmup', '1'],
                          env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True, timeout=300)
    assert done.returncode == 0, done.stderr[-2000:]
    lines = [json.loads(l) for l in done.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1                      # rank 0 only
    line = lines[0]
    assert line['n_gpus'] == 2 and line['ranks'] == 2 and line['data'] == 'dry-run'
    assert line['config']['global_batch'] == 2 * 4096
    # MAX over ranks of the (fabricated) times 1.0 s and 1.25 s, SUM of the env-steps of both shards
    assert line['ms_per_step'] == pytest.approx(1250.0 / 10) and line['value'] == pytest.approx(2 * 4096 * 10 / 1.25)
    assert line['shard_first_env_mean'] == pytest.approx(2048.0)
    # a rank count that contradicts the torchrun environment is refused, not silently benchmarked on one GPU
    env_bad = dict(env, WORLD_SIZE='1', RANK='0', LOCAL_RANK='0')
    bad = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--dry-run'], env=env_bad,
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True, timeout=120)
    assert bad.returncode != 0 and 'WORLD_SIZE=1' in bad.stderr


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
