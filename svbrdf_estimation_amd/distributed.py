"""Multi-GPU sharding of the rendering loss: one process per GPU, batch split by rank.

The hot path has NO exchange step: every (batch item, scene, pixel) is independent and the
loss of a batch is the mean of its items' losses, so each rank computes the loss and
d loss/d input of ITS shard with the same kernels and no data-path collective.  With equal
shards, DDP's gradient averaging of the U-Net (stock PyTorch-ROCm, RCCL over xGMI -- not
part of this package) turns the per-rank means into the global-batch mean.  The only
collective offered here is the one-scalar all-reduce for REPORTING a global loss.

Per-rank scene RNG: the reference samples scenes from the global CPU generator
(losses.py:35); ranks must not draw identical scenes, hence ``rank_seed``.
"""
import torch


def rank_seed(base_seed, rank):
    return int(base_seed) + int(rank)


def shard_bounds(global_batch, rank, world_size):
    """contiguous, equal shards; the global batch must divide evenly (as DistributedSampler pads to)."""
    if global_batch % world_size != 0:
        raise ValueError("global batch %d is not divisible by world size %d" % (global_batch, world_size))
    per = global_batch // world_size
    return rank * per, (rank + 1) * per


def shard(tensor, rank, world_size):
    lo, hi = shard_bounds(tensor.shape[0], rank, world_size)
    return tensor[lo:hi]


class ContiguousShardSampler:
    """Sampler for verification runs: global batch k is the index block [k*G, (k+1)*G), G = per_rank_batch * world,
    and rank r reads its contiguous slice of it, unshuffled -- so that N ranks together see exactly the batches one
    process with batch G sees, in the same item order (DistributedSampler interleaves the ranks' items)."""

    def __init__(self, length, per_rank_batch, rank, world_size):
        self.per, self.rank, self.world = int(per_rank_batch), int(rank), int(world_size)
        self.batches = int(length) // (self.per * self.world)

    def __len__(self):
        return self.batches * self.per

    def __iter__(self):
        G = self.per * self.world
        for k in range(self.batches):
            lo = k * G + self.rank * self.per
            yield from range(lo, lo + self.per)


def global_mean(local_mean):
    """mean over ranks of a per-rank mean (equal shards) -- for logging only."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return local_mean
    t = local_mean.detach().clone().reshape(1)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return (t / dist.get_world_size()).reshape(())
