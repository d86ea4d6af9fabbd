#!/usr/bin/env python3
"""Development aid: config 2 (examples/no_subsampling x1024) stage times, repair list sizes per round and
segment statistics.  usage: c2_probe.py [batch]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
files = pkg.workloads.c2_files()
descs = pkg.parse_many(files)
pkg.Decoder(0)
b = pkg.Batch([descs[i % 4] for i in range(n)], flags=pkg.PROFILE)
for _ in range(3):
    b.decode()
acc = {}
for _ in range(10):
    b.decode()
    for k, v in b.stage_ms().items():
        acc[k] = acc.get(k, 0) + v / 10
print({k: round(v, 4) for k, v in acc.items()}, "Gpixel/s", round(b.total_pixels / acc["total"] / 1e6, 2))
print("repair lists per round:", b.repair_counts(10), "retries", b.retries)
sb = pkg.segment_bytes()
print("segments per image:", [-(-d.scan_len // sb) for d in descs], "scan bytes", [d.scan_len for d in descs])
