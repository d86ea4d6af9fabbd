#!/bin/bash
# Development aid: single-image API latency (jpeg_read_info + jpeg_read_image + image_ycbcr_to_rgb)
# of this library in a warm process against the unmodified reference on the same host, for
# birb.jpg and a 1080p 4:2:0 q85 file.  usage (GPU box, repo root): bash tools/single_image_latency.sh
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import sys, time; sys.path.insert(0, '.')
import __graft_entry__ as ge
pkg = ge.load_package()
open('gpurun_out/lat_1080p.jpg', 'wb').write(pkg.synth.make_jpeg(1920, 1080, seed=1234, quality=85, subsamp=1))
from oracle.oracle import Reference
if Reference.available():
    ref = Reference()
    for name, path in (('birb', 'tests/golden/streams/ex_birb.jpg'), ('1080p', 'gpurun_out/lat_1080p.jpg')):
        f = open(path, 'rb').read()
        ts = []
        for _ in range(5):
            t = time.perf_counter(); ref.decode(f, rgb=True); ts.append(time.perf_counter() - t)
        print(f'reference {name}: best of 5 {min(ts) * 1e3:.2f} ms (two passes: size probe + decode)')
PY
jpeg-decoder_amd/bin/latency tests/golden/streams/ex_birb.jpg 12 | tail -6 || exit 1
jpeg-decoder_amd/bin/latency gpurun_out/lat_1080p.jpg 12 | tail -6
