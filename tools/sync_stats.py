#!/usr/bin/env python3
"""Development aid: how many segments the start-state repair lists per round (run on the GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package()
import bench
wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
files = bench.make_files(pkg, wl, 16)
dec = pkg.Decoder(0)
use = [files[i % len(files)] for i in range(256)]
b = dec.batch(use)
b.decode()
nseg = sum(len(f) for f in use) // pkg.segment_bytes()
print("segments ~", nseg, "listed per round", b.repair_counts(), "retries", b.retries)
