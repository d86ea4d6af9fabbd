#!/usr/bin/env python3
"""Development aid: decode rate and stage times per chroma layout (1080p q85, 1024 images resident):
4:2:0 and 4:4:4 have their own pixel-kernel instantiation, the others take the generic one."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package()
dec = pkg.Decoder(0)
for ss, name in ((1, "4:2:0"), (0, "4:4:4"), (2, "4:2:2"), (3, "4:4:0"), (4, "gray")):
    files = [pkg.synth.make_jpeg(1920, 1080, seed=100 + i, quality=85, subsamp=ss) for i in range(16)]
    descs = pkg.parse_many(files)
    b = pkg.Batch([descs[i % 16] for i in range(1024)], flags=pkg.OUT_RGB | pkg.PROFILE)
    b.decode(); b.decode()
    t = time.perf_counter()
    for _ in range(5): b.decode()
    el = (time.perf_counter() - t) / 5
    ms = b.stage_ms()
    print(f"{name}: {b.total_pixels / el / 1e9:6.1f} Gpixel/s, {el * 1e3:.2f} ms; " + " ".join(f"{k} {v:.2f}" for k, v in ms.items()), flush=True)
    b.close()
