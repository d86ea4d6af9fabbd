#!/usr/bin/env python3
"""Development aid (round-2 measurement): output of a build of the library (JPEGAMD_LIB, e.g. the
all-f32 IDCT variant made by tools/build_variant.sh f32 -DJD_ABL_PIX_F32) against the oracle: how many channel values
differ, and by how much, over every golden stream and a few 1080p images."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package()
from oracle.oracle import Oracle
from conftest import plain_names, golden_stream
o = Oracle()
files = [golden_stream(n) for n in plain_names()] + [pkg.synth.make_jpeg(1920, 1080, seed=1234 + i, quality=85, subsamp=1) for i in range(6)]
out = pkg.Decoder(0).decode(files)
tot = bad = 0
mx = 0
for i, f in enumerate(files):
    rc, want = o.rgb(f)
    got = out.rgb(i)
    d = np.abs(got.astype(int) - want.astype(int))
    tot += d.size; bad += int((d > 0).sum()); mx = max(mx, int(d.max()))
print(f"lib {os.environ.get('JPEGAMD_LIB', 'default')}: {bad} of {tot} channel values differ from the oracle ({bad / tot:.2e}), max |d| = {mx}")
