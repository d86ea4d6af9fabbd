#!/usr/bin/env python3
"""Pin oracle/jpeg_oracle.c against the UNMODIFIED reference (oracle/_ref/libjpegref.so),
live, on the reference's six examples and on synthetic streams larger than the committed
fixtures.  Needs /root/reference (build container only).  Exit code 0 = pinned.

    make -C oracle && python3 oracle/pin_oracle.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from oracle.oracle import Oracle, Reference  # noqa: E402
import __graft_entry__ as ge  # noqa: E402


def main():
    synth = ge.load_package().synth if os.path.exists(os.path.join(ROOT, "jpeg-decoder_amd", "__init__.py")) else None
    o, r = Oracle(), Reference()
    ref_root = os.environ.get("REFERENCE_ROOT", "/root/reference")
    streams = []
    for dp, _, fs in os.walk(os.path.join(ref_root, "examples")):
        streams += [(os.path.join(dp, f), open(os.path.join(dp, f), "rb").read()) for f in fs if f.endswith(".jpg")]
    if synth is not None:
        for (w, h, seed, q, ss) in [(1920, 1080, 1234, 85, 1), (1280, 720, 7, 60, 0), (642, 482, 9, 92, 1),
                                    (1000, 10, 11, 75, 0), (24, 1000, 12, 40, 1)]:
            streams.append((f"synth {w}x{h} q{q} ss{ss}", synth.make_jpeg(w, h, seed, q, ss, 0)))
    bad = 0
    for name, data in streams:
        rc_r, rgb_r = r.decode(data, rgb=True)
        _, ycc_r = r.decode(data, rgb=False)
        rc_o, ycc_o = o.ycbcr(data, honour_dri=False)
        rgb_o = o.to_rgb(ycc_o)
        ok = rc_r == 0 and rc_o == 0 and np.array_equal(ycc_o, ycc_r) and np.array_equal(rgb_o, rgb_r)
        print("PIN" if ok else "MISMATCH", name, rgb_r.shape)
        bad += not ok
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
