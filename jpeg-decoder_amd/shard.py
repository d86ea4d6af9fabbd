"""Sharding of independent images over ranks: no data-path collective.

Huffman time follows the compressed size, pixel time follows width*height; the greedy policy
balances a weighted sum of both.  The only cross-rank traffic is the timing/pixel-count
reduction done by the caller (torch.distributed, RCCL on GPUs / gloo on CPU)."""


def shard_round_robin(n, world, rank):
    return list(range(rank, n, world))


def shard_greedy(costs, world, rank):
    """Longest-processing-time-first: deterministic, identical on every rank."""
    order = sorted(range(len(costs)), key=lambda i: (-costs[i], i))
    load = [0.0] * world
    owner = [0] * len(costs)
    for i in order:
        r = min(range(world), key=lambda k: (load[k], k))
        owner[i] = r
        load[r] += costs[i]
    return [i for i in range(len(costs)) if owner[i] == rank]


def image_cost(scan_bytes, pixels):
    """relative decode cost of one image: measured stage times on MI355X are ~9 us of Huffman
    work per 365 KB of scan and ~4 us of pixel work per 2.07 Mpixel"""
    return scan_bytes * (9.0 / 365e3) + pixels * (4.0 / 2.07e6)


def reduce_throughput(dist, elapsed_s, pixels, device=None):
    """whole-job Mpixel/s: all ranks' pixels over the slowest rank's time"""
    import torch
    t = torch.tensor([elapsed_s], dtype=torch.float64, device=device)
    p = torch.tensor([float(pixels)], dtype=torch.float64, device=device)
    if dist is not None:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(p, op=dist.ReduceOp.SUM)
    return float(p.item()) / float(t.item()) / 1e6, float(t.item()), float(p.item())
