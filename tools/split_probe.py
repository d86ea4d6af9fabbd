#!/usr/bin/env python3
"""Development aid: does a batch decode faster as k sub-batches on k streams (their kernels overlap: the tail of
one kernel under the head of another, the bandwidth-bound unstuff stage under the issue-bound ones)?
usage: split_probe.py [images] [parts ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package()
import bench
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
parts = [int(x) for x in sys.argv[2:]] or [1, 2, 4]
files = bench.make_files(pkg, "c3", 64)
pkg.Decoder(0)
descs = pkg.parse_many(files)
for k in parts:
    bs = [pkg.Batch([descs[i % 64] for i in range(j * N // k, (j + 1) * N // k)]) for j in range(k)]
    for _ in range(2):
        for b in bs: b.decode(sync=False)
        for b in bs: b.sync()
    t0 = time.perf_counter()
    for _ in range(10):
        for b in bs: b.decode(sync=False)
        for b in bs: b.sync()
    dt = (time.perf_counter() - t0) / 10
    px = sum(b.total_pixels for b in bs)
    print(f"{N} images as {k} batch(es) on {k} stream(s): {dt * 1e3:.3f} ms per {N}, {px / dt / 1e9:.1f} Gpixel/s", flush=True)
    for b in bs: b.close()
# steady state with two batches in flight, decoded alternately (what a loader's stream of batches looks like)
for n in (N, N // 2):
    A = pkg.Batch([descs[i % 64] for i in range(n)]); B = pkg.Batch([descs[(i + 7) % 64] for i in range(n)])
    A.decode(); B.decode()
    A.decode(sync=False)
    t0 = time.perf_counter(); steps = 0
    for _ in range(10):
        B.decode(sync=False); A.sync(); steps += 1
        A.decode(sync=False); B.sync(); steps += 1
    A.sync()
    dt = (time.perf_counter() - t0) / steps
    print(f"two batches of {n} in flight, alternating: {dt * 1e3:.3f} ms per batch, {A.total_pixels / dt / 1e9:.1f} Gpixel/s", flush=True)
    A.close(); B.close()
