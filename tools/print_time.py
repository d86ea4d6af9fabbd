#!/usr/bin/env python3
"""Development aid (GPU box): image_print's text for a 1080p image, host C (to /dev/null) against
jdgpu_image_format."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package()
pkg.Decoder(0)
L = pkg.lib()
img = np.random.default_rng(1).integers(0, 256, (1080, 1920, 3), dtype=np.uint8)
for _ in range(3):
    t = time.perf_counter(); text = pkg.format_image(img, 1 << 30, 1080, 2); tg = time.perf_counter() - t
print(f"jdgpu_image_format 1920x1080, 2 spaces: {len(text)/1e6:.1f} MB of text in {tg*1e3:.1f} ms (length query + format)")
# host image_print into /dev/null
L.image_print.argtypes = [C.c_void_p] + [C.c_uint] * 5
libc = C.CDLL(None)
sys.stdout.flush()
saved = os.dup(1); null = os.open("/dev/null", os.O_WRONLY); os.dup2(null, 1)
t = time.perf_counter(); L.image_print(img.ctypes.data, 1920, 1080, 1 << 30, 1080, 2); libc.fflush(None); th = time.perf_counter() - t
os.dup2(saved, 1)
print(f"image_print (host C, stdout = /dev/null): {th*1e3:.1f} ms")
