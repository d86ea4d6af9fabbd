"""GPU (-m gpu): the library used from several host threads at once (SURVEY 8b "Threading": re-entrant on
distinct objects; INTEGRATION.md: create batch k+1 on another thread while batch k decodes).  Every thread
creates, decodes, reads back and destroys its own batches -- single images (the jpeg_read_image path: a batch
of one, a packed plan, a recycled stream), small mixed batches, batches from files -- and calls the
colour-conversion entry points, which share the static scratch.  ctypes releases the interpreter lock during
the calls, so the library's caches, staging buffer and upload lock really are contended.  Everything is
compared with results computed beforehand on one thread (those are pinned to the oracle by the other GPU
tests)."""
import hashlib
import threading

import numpy as np
import pytest

from conftest import all_golden_names, golden_stream, load_pkg

pytestmark = pytest.mark.gpu


def test_batches_from_eight_host_threads_at_once():
    pkg = load_pkg()
    dec = pkg.Decoder(0)
    names = all_golden_names()
    files = [golden_stream(n) for n in names]
    layouts = (pkg.OUT_RGB, pkg.OUT_YCBCR, pkg.OUT_RGB | pkg.OUT_4BYTE, pkg.OUT_RGB | pkg.OUT_PLANAR, pkg.OUT_RGB | pkg.OUT_PITCHED)
    # what every (stream, layout) must give, from one thread
    want = {}
    for fl in layouts:
        out = dec.decode(files, fl)
        for i in range(len(files)):
            assert out.status(i) == 0
            want[i, fl] = hashlib.md5(out.pixels(i).tobytes()).hexdigest()
    ycc = {i: dec.decode([files[i]], pkg.OUT_YCBCR).pixels(0).copy() for i in range(len(files))}
    rgb = {i: dec.decode([files[i]], pkg.OUT_RGB).pixels(0).copy() for i in range(len(files))}
    L = pkg.lib()
    errors = []

    def worker(t):
        try:
            rng = np.random.default_rng(1000 + t)
            for it in range(120):
                fl = layouts[int(rng.integers(len(layouts)))]
                kind = it % 4
                if kind == 0:      # a single image, as jpeg_read_image creates it
                    i = int(rng.integers(len(files)))
                    out = dec.decode([files[i]], fl)
                    assert out.status(0) == 0 and hashlib.md5(out.pixels(0).tobytes()).hexdigest() == want[i, fl], (t, it, names[i])
                elif kind == 1:    # a small mixed batch
                    idx = [int(x) for x in rng.integers(0, len(files), int(rng.integers(2, 40)))]
                    out = dec.decode([files[i] for i in idx], fl)
                    for k, i in enumerate(idx):
                        assert out.status(k) == 0 and hashlib.md5(out.pixels(k).tobytes()).hexdigest() == want[i, fl], (t, it, names[i])
                elif kind == 2:    # whole files in, headers parsed on the device
                    idx = [int(x) for x in rng.integers(0, len(files), int(rng.integers(1, 24)))]
                    out = pkg.Batch.from_files([files[i] for i in idx], fl).decode()
                    for k, i in enumerate(idx):
                        assert out.status(k) == 0 and hashlib.md5(out.pixels(k).tobytes()).hexdigest() == want[i, fl], (t, it, names[i])
                else:              # the colour conversion on host pixels, both entry points (one static scratch)
                    i = int(rng.integers(len(files)))
                    h, w = ycc[i].shape[:2]
                    for fn in (L.jdgpu_ycbcr_to_rgb, L.jdgpu_ycbcr_to_rgb_noalloc):
                        px = ycc[i].copy()
                        assert fn(px.ctypes.data, w, h) == 0
                        assert np.array_equal(px, rgb[i]), (t, it, names[i])
        except BaseException as e:   # noqa: BLE001 -- handed to the main thread
            errors.append(e)

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(8)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors[0]
