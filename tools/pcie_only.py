import sys, json
sys.path.insert(0, "/root/repo")
import bench, __graft_entry__ as ge
pkg = ge.load_package(); pkg.Decoder(0)
files = bench.make_files(pkg, "c3", 64)
for r in range(2):
    p = bench.pcie_inclusive(pkg, files, 4096)
    print({k:p[k] for k in ("unpipelined","pipelined","pipelined_device_headers","pipelined_pageable_files","pcie_ceiling")}, p["ms_per_batch"], p["create_ms"], flush=True)
