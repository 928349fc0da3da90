"""Multi-GPU layout: the environment batch shards embarrassingly (one process per GPU, contiguous
environment-index blocks, no data-path collective).  The only exchange is an all-gather of episode
statistics for logging (RCCL over xGMI when the process group backend is "nccl"; gloo in CPU tests).
"""
import torch
import torch.distributed as dist

__all__ = ['shard_of', 'gather_episode_stats', 'EpisodeStats', 'reduce_job']


def shard_of(global_batch, rank, world_size):
    """(first_env_index, num_envs) of `rank`: contiguous blocks, remainder spread over the first ranks."""
    base, extra = divmod(int(global_batch), int(world_size))
    count = base + (1 if rank < extra else 0)
    first = rank * base + min(rank, extra)
    return first, count


class EpisodeStats:
    """Running sums of per-episode results for the environments of this rank."""

    FIELDS = ('episodes', 'return_sum', 'length_sum', 'coverage_sum', 'delivered_sum')

    def __init__(self, device='cpu'):
        self.sums = torch.zeros(len(self.FIELDS), dtype=torch.float64, device=device)

    def update(self, done, episode_return, episode_length, coverage_rate, delivered):
        done = done.to(torch.float64)
        self.sums += torch.stack([done.sum(), (done * episode_return).sum(), (done * episode_length).sum(),
                                  (done * coverage_rate).sum(), (done * delivered).sum()])

    def as_dict(self, sums=None):
        sums = self.sums if sums is None else sums
        n = max(float(sums[0]), 1.0)
        return {'episodes': float(sums[0]), 'mean_return': float(sums[1]) / n, 'mean_length': float(sums[2]) / n,
                'mean_coverage_rate': float(sums[3]) / n, 'mean_delivered': float(sums[4]) / n}


def gather_episode_stats(stats):
    """All-gather the per-rank sums and return (global dict, [per-rank tensors]).  A few dozen bytes per
    rank: latency-bound, issued off the critical path."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return stats.as_dict(), [stats.sums.clone()]
    parts = [torch.zeros_like(stats.sums) for _ in range(dist.get_world_size())]
    dist.all_gather(parts, stats.sums)
    total = torch.stack(parts).sum(dim=0)
    return stats.as_dict(total), parts


def reduce_job(elapsed, executed, stats, device='cpu', force=False):
    """Whole-job numbers of a sharded benchmark run: (MAX over ranks of the timed region, SUM of the env-steps every
    rank executed, per-rank statistics averaged).  Three tiny collectives after the timed region; identity on one rank
    (`force`: run them on one rank too -- the single-GPU rehearsal of the N-rank path)."""
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size() == 1 and not force):
        return float(elapsed), float(executed), stats
    t = torch.tensor([elapsed], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    ex = torch.tensor([executed], dtype=torch.float64, device=device)
    dist.all_reduce(ex, op=dist.ReduceOp.SUM)
    gathered = [torch.zeros_like(stats) for _ in range(dist.get_world_size())]
    dist.all_gather(gathered, stats)              # the only collective of the path: episode statistics
    return float(t.item()), float(ex.item()), torch.stack(gathered).mean(dim=0)
