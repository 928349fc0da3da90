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
