#!/usr/bin/env python3
"""Development aid: repair list sizes per round on the headline workload (C3).  usage: c3_repair_probe.py [batch]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package()
import bench
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
files = bench.make_files(pkg, "c3", 16)
descs = pkg.parse_many(files)
pkg.Decoder(0)
b = pkg.Batch([descs[i % 16] for i in range(n)], flags=pkg.PROFILE)
for _ in range(3):
    b.decode()
print({k: round(v, 4) for k, v in b.stage_ms().items()})
sb = pkg.segment_bytes()
segs = sum(-(-descs[i % 16].scan_len // sb) for i in range(n))
rc = b.repair_counts(10)
print("segments", segs, "repair lists per round:", rc, [round(c / segs, 4) for c in rc], "retries", b.retries)
