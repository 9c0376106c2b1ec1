"""Image-level sharding for multi-GPU runs: images are independent
units, rank r of W owns a contiguous chunk, there is no data-path collective —
only a host-side barrier and a max-over-ranks of the elapsed time."""


def shard_range(n_items, rank, world):
    """Contiguous, balanced partition: sizes differ by at most one."""
    if world < 1 or not 0 <= rank < world:
        raise ValueError("bad rank/world %r/%r" % (rank, world))
    base, extra = divmod(n_items, world)
    start = rank * base + min(rank, extra)
    return range(start, start + base + (1 if rank < extra else 0))


def shard_items(items, rank, world):
    r = shard_range(len(items), rank, world)
    return [items[i] for i in r]


def aggregate_throughput(local_units, local_seconds, dist=None, device=None):
    """Whole-job units/s: sum of units over ranks / max of time over ranks.

    `dist` is torch.distributed (initialised) or None for a single process."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return local_units / local_seconds, local_units, local_seconds
    import torch
    t = torch.tensor([local_seconds], dtype=torch.float64, device=device)
    u = torch.tensor([float(local_units)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(u, op=dist.ReduceOp.SUM)
    return u.item() / t.item(), u.item(), t.item()
