#!/usr/bin/env python3
"""Development aid: repair-round statistics of k_huff_sync (needs a -DJD_STATS build, JPEGAMD_LIB=...)."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package()
import bench
wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
files = bench.make_files(pkg, wl, 16, 0)
dec = pkg.Decoder(0)
b = dec.batch([files[i % len(files)] for i in range(256)])
b.decode()
out = (C.c_uint32 * 16)()
L = pkg.lib()
L.jdgpu_batch_debug_counters.argtypes = [C.c_void_p, C.POINTER(C.c_uint32)]
L.jdgpu_batch_debug_counters(b._h, out)
nseg = sum(len(f) for f in [files[i % len(files)] for i in range(256)]) // pkg.segment_bytes()
print("segments ~", nseg, "work-groups ~", nseg // 256, "counters", list(out)[:6])
