#!/usr/bin/env python3
"""Development aid: randomized parity soak on the GPU box.  N random streams (sizes 1..400, all five
layouts, the four with a sampling factor of 4 and every luma H x V of 1..4, qualities 1..100, restart intervals, white-noise and flat content, random corruptions)
decoded in batches of 256 in every output layout (dense and pitched rows; every other batch through
jdgpu_batch_create_from_files), each compared with the oracle: status class,
coefficients and pixels bit for bit.  usage: soak.py [N] [seed] [max_side, default 400]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package()
from oracle.oracle import Oracle

N = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
SIDE = int(sys.argv[3]) if len(sys.argv) > 3 else 400
CHUNK = int(os.environ.get("SOAK_CHUNK", "0")) or (256 if SIDE <= 400 else 32)   # images per batch
rng = np.random.default_rng(seed)
oracle = Oracle()
dec = pkg.Decoder(0)
want_bits = {-2: pkg.IMG_EOF, -3: pkg.IMG_NOSYMBOL}
LAYOUTS = ((pkg.OUT_RGB, lambda a: a), (pkg.OUT_4BYTE, lambda a: a[..., :3]), (pkg.OUT_PLANAR, lambda a: a.transpose(1, 2, 0)),
           (pkg.OUT_RGB | pkg.OUT_PITCHED, lambda a: a), (pkg.OUT_4BYTE | pkg.OUT_PITCHED, lambda a: a[..., :3]),
           (pkg.OUT_PLANAR | pkg.OUT_PITCHED, lambda a: a.transpose(1, 2, 0)))
bad = done = failed_streams = 0
t0 = time.time()
while done < N:
    files = []
    for _ in range(min(CHUNK, N - done)):
        ss = int(rng.integers(0, 9))
        w, h = int(rng.integers(1, SIDE + 1)), int(rng.integers(1, SIDE + 1))
        q = int(rng.integers(1, 101))
        mw = {1: 16, 2: 16, 5: 32, 6: 32, 8: 16}.get(ss, 8)
        mh = {1: 16, 3: 16, 6: 16, 7: 32, 8: 32}.get(ss, 8)
        if rng.random() < 0.25:   # any luma H x V of 1..4 (factor 3, 4x4 = 18 data units per MCU), chroma 1x1
            lh, lv = int(rng.integers(1, 5)), int(rng.integers(1, 5))
            ss, mw, mh = 0x100 | lh << 4 | lv, 8 * lh, 8 * lv
        mcus = -(-w // mw) * -(-h // mh)
        ri = 0 if rng.random() < 0.5 else int(rng.integers(1, mcus + 2))
        kind = rng.random()
        if kind < 0.7:
            f = pkg.synth.make_jpeg(w, h, seed=int(rng.integers(1 << 30)), quality=q, subsamp=ss, restart_mcus=ri)
        elif kind < 0.85:
            f = pkg.synth.encode_rgb(rng.integers(0, 256, (h, w, 3), dtype=np.uint8), quality=q, subsamp=ss, restart_mcus=ri)
        else:
            f = pkg.synth.encode_rgb(np.full((h, w, 3), int(rng.integers(256)), np.uint8), quality=q, subsamp=ss, restart_mcus=ri)
        if rng.random() < 0.15:   # corrupt the scan
            g = bytearray(f)
            off = oracle.parse(f)[1].scan_offset
            pos = off + int(rng.integers(0, max(1, len(g) - off - 2)))
            c = rng.random()
            if c < 0.4: g[pos] ^= 1 << int(rng.integers(8))
            elif c < 0.7: g = g[:pos]
            else: g[pos:pos + 2] = bytes([0xFF, int(rng.integers(256))])
            f = bytes(g)
        files.append(f)
    flags, view = LAYOUTS[(done // CHUNK) % 6]
    # every other batch: whole files uploaded, headers parsed and scans gathered on the device
    out = pkg.Batch.from_files(files, flags).decode() if (done // CHUNK) % 2 else dec.decode(files, flags)
    for i, f in enumerate(files):
        rc, info, qc = oracle.qcoef(f)
        if rc:
            failed_streams += 1
            ok = bool(out.status(i) & want_bits[rc])
        else:
            ok = out.status(i) == 0 and np.array_equal(out.qcoef(i).astype(np.int32), qc) and \
                np.array_equal(view(out.pixels(i)), oracle.to_rgb(oracle.pixels_from_qcoef(info, qc)))
        if not ok:
            bad += 1
            d = pkg.parse(f)
            print(f"MISMATCH stream {done + i}: {d.width}x{d.height} ncomp {d.ncomp} H{d.comp[0].H}V{d.comp[0].V} ri {d.restart_interval} "
                  f"len {len(f)} oracle rc {rc} status {out.status(i)}", flush=True)
            os.makedirs("gpurun_out/soak_bad", exist_ok=True)
            open(f"gpurun_out/soak_bad/s{seed}_{done + i}.jpg", "wb").write(f)
    done += len(files)
    print(f"{done} streams, {failed_streams} undecodable by design, {bad} mismatches, retries {out.retries}, {time.time() - t0:.0f} s", flush=True)
    out.close()
sys.exit(1 if bad else 0)
