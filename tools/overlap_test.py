#!/usr/bin/env python3
"""Development aid: do two batches decoded on their own streams overlap usefully?  (Huffman stages
stall on latency, the pixel stage on f64 issue.)  Compares one batch of 2048 with 2 x 1024 and
4 x 512 enqueued back to back without waiting in between."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package()
import bench
files = bench.make_files(pkg, "c3", 64)
dec = pkg.Decoder(0)
descs = [pkg.parse(f) for f in files]
for parts in (1, 2, 4):
    n = 2048 // parts
    batches = [pkg.Batch([descs[(i + p * n) % 64] for i in range(n)]) for p in range(parts)]
    for _ in range(2):
        for b in batches: b.decode(sync=False)
        for b in batches: b.sync()
    t = time.perf_counter()
    K = 10
    for _ in range(K):
        for b in batches: b.decode(sync=False)
        for b in batches: b.sync()
    el = (time.perf_counter() - t) / K
    px = sum(b.total_pixels for b in batches)
    print(f"{parts} x {n}: {el*1e3:.2f} ms per 2048 images, {px/el/1e9:.1f} Gpixel/s, retries {[b.retries for b in batches]}")
    for b in batches: b.close()
