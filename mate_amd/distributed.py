"""Multi-GPU layout: the environment batch shards embarrassingly (one process per GPU, contiguous
environment-index blocks, no data-path collective).  The only exchange is an all-gather of episode
statistics for logging (RCCL over xGMI when the process group backend is "nccl"; gloo in CPU tests).
"""
import torch
import torch.distributed as dist

__all__ = ['shard_of', 'gather_episode_stats', 'EpisodeStats']


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
