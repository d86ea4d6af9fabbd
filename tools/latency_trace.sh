#!/bin/bash
# Development aid: HIP API + kernel timeline of one warm single-image call (bin/latency) under rocprofv3.
# usage (GPU box, repo root): bash tools/latency_trace.sh [file.jpg]   -> gpurun_out/lat_trace.txt
R=$GRAFT_REPO_ROOT
F=${1:-$R/tests/golden/streams/ex_birb.jpg}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/lat_prof
timeout -k 10 120 rocprofv3 --kernel-trace --hip-trace -d $R/gpurun_out/lat_prof -o lat -- $R/jpeg-decoder_amd/bin/latency $F 40 > $R/gpurun_out/lat_prof.log 2>&1 || exit 1
python3 - $R/gpurun_out/lat_prof/lat_results.db > $R/gpurun_out/lat_trace.txt <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
ev = [(s, e, 'API ' + n) for n, s, e in db.execute("select name,start,end from regions")]
ev += [(s, e, 'K   ' + n[:44]) for n, s, e in db.execute("select name,start,end from kernels")]
ev.sort()
starts = [s for s, e, n in ev if n.startswith('API hipSetDevice')]
t0 = starts[-2]          # the last call's jpeg_read_image (the one after it is image_ycbcr_to_rgb)
for s, e, n in ev:
    if s >= t0 and not n.startswith('API __hip'):
        print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:8.1f}  {n}")
PY
