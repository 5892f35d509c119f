"""Frame sharding across the GPUs of one node (one process per GPU).

Frames are independent through the whole path, so the only multi-GPU machinery is a
contiguous split of a batch over ranks (keeps video order) and an order-preserving
gather of the variable-length per-frame results on the host.  No data-path collective:
`torch.distributed` (RCCL on the GPU box, gloo in CPU tests) is used for the barrier and
for gathering Python result objects only.
"""


def shard_bounds(n, world_size, rank):
    """[lo, hi) of the contiguous slice of `n` frames owned by `rank` (sizes differ by <= 1)."""
    base, rem = divmod(n, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard(seq, world_size, rank):
    lo, hi = shard_bounds(len(seq), world_size, rank)
    return seq[lo:hi]


def gather_results(local_results, dist=None, dst=0):
    """Concatenate per-frame result lists of all ranks in rank order on `dst`.

    `dist` is `torch.distributed` (already initialised) or None for a single process."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return list(local_results)
    world, rank = dist.get_world_size(), dist.get_rank()
    bucket = [None] * world if rank == dst else None
    dist.gather_object(list(local_results), bucket, dst=dst)
    if rank != dst:
        return None
    out = []
    for part in bucket:
        out.extend(part)
    return out


def run_sharded(frames, fn, dist=None, dst=0):
    """Run `fn(frames_slice) -> list of per-frame results` on this rank's shard and gather."""
    if dist is None or not dist.is_initialized():
        return fn(frames)
    lo, hi = shard_bounds(len(frames), dist.get_world_size(), dist.get_rank())
    return gather_results(fn(frames[lo:hi]), dist, dst)
