#!/usr/bin/env python3
"""Development aid (GPU box): what one image that overflows the narrow units costs a batch.
A 1024 x 1080p batch alone, and with one checkerboard image whose last AC coefficient does not fit
4 bits appended (the safety net marks it wide and a second pass does that image again)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package()
import bench
dec = pkg.Decoder(0)
files = bench.make_files(pkg, "c3", 16)
batch = [files[i % 16] for i in range(1024)]
y, x = np.mgrid[0:64, 0:64]
checker = np.where((x + y) % 2 == 0, 255, 0).astype(np.uint8)
bad = pkg.synth.encode_rgb(np.stack([checker] * 3, -1), quality=85, subsamp=0)
for name, fl in (("1024 x 1080p", batch), ("the same + 1 overflowing image", batch + [bad])):
    b = pkg.Batch(pkg.parse_many(fl))
    b.decode()
    t = time.perf_counter(); b.decode(); dt = time.perf_counter() - t
    print(f"{name}: {dt*1e3:.2f} ms, retries {b.retries}, statuses ok {all(b.status(i) == 0 for i in range(b.n))}")
    b.close()
