"""CPU, world_size 2 over gloo: the N>1 control flow of the image-sharded path (sharding is
disjoint + complete + balanced, the reduction is max-time / sum-pixels).  The decode itself has
no collective, so there is nothing else to rehearse without GPUs."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    import __graft_entry__ as ge
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pkg = ge.load_package()
    from conftest import all_golden_names, golden_stream
    files = [golden_stream(n) for n in all_golden_names()]
    descs = [pkg.parse(f) for f in files]
    costs = [pkg.shard.image_cost(d.scan_len, d.width * d.height) for d in descs]
    mine = pkg.shard.shard_greedy(costs, world, rank)
    rr = pkg.shard.shard_round_robin(len(files), world, rank)
    pixels = sum(descs[i].width * descs[i].height for i in mine)
    mpix, t, p = pkg.shard.reduce_throughput(dist, 0.5 + rank, pixels)
    gathered = [None] * world
    dist.all_gather_object(gathered, (mine, rr, sum(costs[i] for i in mine)))
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, mpix, t, p, gathered, len(files), sum(d.width * d.height for d in descs)))


def test_two_ranks_gloo():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    for rank, mpix, t, p, gathered, n, total_px in res:
        assert t == pytest.approx(1.5)                     # max over ranks
        assert p == pytest.approx(total_px)                # every image counted once
        assert mpix == pytest.approx(total_px / 1.5 / 1e6)
        a, b = gathered[0][0], gathered[1][0]
        assert sorted(a + b) == list(range(n)) and not set(a) & set(b)
        assert sorted(gathered[0][1] + gathered[1][1]) == list(range(n))
        la, lb = gathered[0][2], gathered[1][2]
        assert abs(la - lb) <= 0.25 * (la + lb)            # greedy balance


def test_greedy_is_deterministic_and_complete():
    sys.path.insert(0, ROOT)
    import __graft_entry__ as ge
    sh = ge.load_package().shard
    costs = [((i * 7919) % 97) + 1 for i in range(1000)]
    parts = [sh.shard_greedy(costs, 8, r) for r in range(8)]
    assert sorted(sum(parts, [])) == list(range(1000))
    loads = [sum(costs[i] for i in p) for p in parts]
    assert max(loads) - min(loads) <= max(costs)
    assert parts == [sh.shard_greedy(costs, 8, r) for r in range(8)]
