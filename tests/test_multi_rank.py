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


def _bench(args, env_extra, timeout=300):
    import json
    import subprocess
    env = dict(os.environ, **env_extra)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        if k not in env_extra:
            env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True,
                       text=True, timeout=timeout)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{"metric"')]
    return r, [json.loads(ln) for ln in lines]


def test_bench_gpus_2_starts_two_ranks_itself():
    """`python bench.py --gpus 2` with no launcher around it (the command line the driver uses for N = 1,
    with another N): two child ranks, a rendezvous over 127.0.0.1, each rank's own shard of the global
    config-5 list, ONE JSON line from rank 0 with n_gpus = 2 and both ranks' pixel counts.
    JD_BENCH_DRYRUN leaves out the GPU only."""
    r, out = _bench(["--gpus", "2", "--workload", "c5", "--batch", "48", "--distinct", "12", "--steps", "1", "--warmup", "0"],
                    {"JD_BENCH_DRYRUN": "1"})
    assert r.returncode == 0, r.stderr[-2000:]
    assert len(out) == 1
    o = out[0]
    assert o["n_gpus"] == 2 and o["scaling"] == "weak" and o["dry_run"] is True
    per_rank = o["config"]["pixels_per_step_per_rank"]
    assert len(per_rank) == 2 and all(p > 0 for p in per_rank)
    assert per_rank[0] == o["config"]["pixels_per_step_per_gpu"]
    assert o["config"]["batch_per_gpu"] in (47, 48, 49)         # 96 images over 2 ranks, balanced by cost
    assert o["ms_per_step"] == pytest.approx(2000.0)             # max over ranks of the dry run's 1 s + rank


def test_bench_refuses_gpus_that_contradict_the_launcher():
    r, out = _bench(["--gpus", "2", "--steps", "1", "--warmup", "0"],
                    {"JD_BENCH_DRYRUN": "1", "WORLD_SIZE": "4", "RANK": "0", "LOCAL_RANK": "0",
                     "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29999"}, timeout=60)
    assert r.returncode != 0 and not out
    assert "WORLD_SIZE=4" in r.stderr


def test_bench_reports_a_failed_rank():
    """a rank that dies takes the run's exit status with it, and the launcher ends the surviving rank
    (which would otherwise sit in the rendezvous until its time-out)"""
    import time
    t0 = time.time()
    r, out = _bench(["--gpus", "2", "--workload", "c5", "--batch", "48", "--distinct", "12", "--steps", "1", "--warmup", "0"],
                    {"JD_BENCH_DRYRUN": "1", "JD_BENCH_FAIL_RANK": "1"}, timeout=200)
    assert r.returncode != 0 and not out
    assert "ranks failed" in r.stderr
    assert time.time() - t0 < 120
