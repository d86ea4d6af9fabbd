#!/usr/bin/env python3
"""Development aid: sustained PCIe-INCLUSIVE rate of a loader-style caller.  P producer threads
parse headers (jdgpu_parse_files) and create batches (plan, gather into the pinned staging
buffer, upload) while the main thread decodes the previous batch; every batch has its own stream,
so uploads and kernels overlap.  usage: pipeline_demo.py [batches] [producers] [batch size]"""
import os, queue, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package()
import bench

NB = int(sys.argv[1]) if len(sys.argv) > 1 else 12
P = min(2, int(sys.argv[2]) if len(sys.argv) > 2 else 2)   # each batch in flight holds ~56 GB of HBM at 4096 x 1080p
BS = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
files = bench.make_files(pkg, "c3", 64)
dec = pkg.Decoder(0)
batch_files = [files[i % 64] for i in range(BS)]
q = queue.Queue(maxsize=2)
todo = list(range(NB + 2))
lock = threading.Lock()

def producer():
    while True:
        with lock:
            if not todo:
                return
            todo.pop()
        t = time.perf_counter()
        try:
            descs = pkg.parse_many(batch_files)
            b = pkg.Batch(descs)
        except Exception as e:      # e.g. out of device memory with too many batches in flight
            q.put((e, 0.0))
            return
        q.put((b, time.perf_counter() - t))

threads = [threading.Thread(target=producer, daemon=True) for _ in range(P)]
for t in threads: t.start()
done = pixels = 0
t0 = None
create_s = []
while done < NB + 2:
    b, tc = q.get()
    if isinstance(b, Exception):
        raise SystemExit(f"a producer failed: {b}")
    if done == 2:            # the first two batches warm up (pinned staging allocation, code load)
        t0 = time.perf_counter(); pixels = 0
    b.decode()
    bad = sum(1 for i in (0, b.n // 2, b.n - 1) if b.status(i))
    assert not bad
    pixels += b.total_pixels
    create_s.append(tc)
    b.close()
    done += 1
el = time.perf_counter() - t0
print(f"{NB} batches of {BS} x 1080p, {P} producer thread(s): {el / NB * 1e3:.1f} ms per batch sustained, "
      f"{pixels / el / 1e9:.1f} Gpixel/s PCIe-inclusive (parse + create per batch {sum(create_s[2:]) / NB * 1e3:.0f} ms on its thread)")
