#!/usr/bin/env python3
"""Development aid: what the caller side costs (run on the GPU box): header parsing, batch
creation (plan + gather + upload; JDGPU_TRACE=1 prints the parts) and the PCIe-inclusive rate."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package()
import bench
files = bench.make_files(pkg, "c3", 64)
dec = pkg.Decoder(0)
t = time.perf_counter(); descs = [pkg.parse(f) for f in files]; tp = (time.perf_counter() - t) / len(files)
use = [descs[i % 64] for i in range(4096)]
# the headers of 4096 files through jdgpu_parse_files (C call only; the arrays are built before)
import ctypes as C, numpy as np
L = pkg.lib()
keep = [np.frombuffer(files[i % 64], dtype=np.uint8) for i in range(4096)]
ptrs = (C.c_void_p * 4096)(*[k.ctypes.data for k in keep]); lens = (C.c_size_t * 4096)(*[k.size for k in keep])
arr = (pkg.ImageDesc * 4096)()
for th in (1, 4, 8, 16, 0):
    t = time.perf_counter(); bad = L.jdgpu_parse_files(ptrs, lens, 4096, arr, None, th); tt = time.perf_counter() - t
    print(f"jdgpu_parse_files(4096 files, threads={th}): {tt*1e3:.1f} ms, {bad} rejected")
for attempt in ("first", "second"):
    t = time.perf_counter(); b = pkg.Batch(use); tc = time.perf_counter() - t
    b.decode()
    t = time.perf_counter(); b.decode(); td = time.perf_counter() - t
    print(f"{attempt} batch: parse {tp*1e6:.1f} us/image; batch_create(4096) {tc*1e3:.1f} ms; decode {td*1e3:.1f} ms; "
          f"scan {b.total_scan_bytes/1e9:.2f} GB; PCIe-inclusive {b.total_pixels/(tc+td)/1e6:.0f} Mpixel/s, "
          f"resident {b.total_pixels/td/1e6:.0f} Mpixel/s")
    b.close()
# the same from whole files with the headers parsed on the device (jdgpu_batch_create_from_files)
h = C.c_void_p(); status = (C.c_int * 4096)()
for attempt in ("first", "second", "third"):
    t = time.perf_counter()
    rc = L.jdgpu_batch_create_from_files(ptrs, lens, 4096, pkg.OUT_RGB, C.byref(h), status); tc = time.perf_counter() - t
    assert rc == 0 and not any(status)
    L.jdgpu_batch_decode(h); L.jdgpu_batch_sync(h)
    t = time.perf_counter(); L.jdgpu_batch_decode(h); L.jdgpu_batch_sync(h); td = time.perf_counter() - t
    px = L.jdgpu_batch_total_pixels(h)
    print(f"{attempt} from_files: create {tc*1e3:.1f} ms (no host parse); decode {td*1e3:.1f} ms; PCIe-inclusive {px/(tc+td)/1e6:.0f} Mpixel/s")
    L.jdgpu_batch_destroy(h)
# device parse alone, files resident
import torch
blob = torch.from_numpy(np.concatenate(keep)).cuda()
offs = np.concatenate([[0], np.cumsum([k.size for k in keep])]).astype(np.uint64)
for attempt in range(3):
    t = time.perf_counter(); d = pkg.parse_device(blob.data_ptr(), offs); tt = time.perf_counter() - t
    print(f"jdgpu_parse_files_device(4096 resident files): {tt*1e3:.2f} ms, {sum(1 for s in d.status if s)} rejected")
