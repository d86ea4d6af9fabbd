#!/usr/bin/env python3
"""bench.py -- decoded Mpixels/s of the MI355X JPEG decode path (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c3|c2|c4|c5] [--batch B] [--lean]

A "step" is one full decode (unstuff -> Huffman sync/scan/write -> fused pixel kernel) of one
batch of synthetic baseline JPEGs whose scan bytes, tables and descriptors are already
resident in HBM.  N > 1: one process per GPU (torch.distributed / RCCL only for the barrier and
the reductions of time and pixel counts -- the path has no data-path collective: images are
independent).  c3 / c4 / c2: every rank decodes its own batch of the same shape; c5: every rank
derives the same global list of 8192 x N images (parameters only) and takes its shard with
shard_greedy.  Per-GPU work is fixed as N grows: "weak" scaling.

Prints ONE JSON line (rank 0).  `roofline` is for the stage that takes the most time, measured
with HIP events on the decode stream inside the timed steps; `pixel_stage` repeats the same
figures for the fused pixel kernel, which BASELINE.json's metric names explicitly.
`cpu_baseline` is the UNMODIFIED reference (oracle/_ref/libjpegref.so, built from
/root/reference by oracle/Makefile) -- or, where that file is absent, this repo's CPU
restatement -- timed on ONE pinned host core over a bounded sample of the same workload, plus
the same on all the cores this process may use (`all_cores`).  At N = 1 the default run adds
`pcie_inclusive` (host files -> parse -> create -> decode, pipelined and not) and
`other_workloads` (short runs of the other BASELINE configs); --lean leaves them out.

`python bench.py --gpus N` with no launcher around it (WORLD_SIZE unset) starts the N ranks itself:
N fresh child processes of the same command line with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set,
before this process has touched torch or HIP; it relays rank 0's JSON line and exits non-zero if a
child does.  Under `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N` the
ranks are the launcher's; --gpus must then equal WORLD_SIZE.

`verified`: after the timed steps (outside them) every copy of every distinct stream is compared on
the device with the first copy of that stream, and one copy each of up to 8 distinct streams with
the CPU oracle's coefficients and RGB (oracle/ used as the checker, like in the cpu_baseline leg);
a mismatch ends the run with a non-zero exit status.
"""
import argparse
import itertools
import json
import os
import queue
import socket
import subprocess
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)

WORKLOADS = {
    # name: (description, default batch, distinct images)
    "c3": ("synthetic 1920x1080 4:2:0 q85, no restart markers (BASELINE.json configs[2])", 4096, 64),
    "c2": ("examples/no_subsampling 4:4:4 files x1024 (BASELINE.json configs[1])", 1024, 4),
    "c4": ("synthetic 3840x2160 4:2:0 q85, DRI = 1 MCU row (BASELINE.json configs[3])", 512, 16),
    "c5": ("synthetic mixed-size 4:2:0/4:4:4, global list of 8192 x N sharded by shard_greedy "
           "(BASELINE.json configs[4]: 65536 over 8 GPUs)", 8192, 96),
}


def make_files(pkg, workload, distinct):
    wl = pkg.workloads
    if workload == "c2":
        return wl.c2_files()
    params = {"c3": wl.c3_params, "c4": wl.c4_params, "c5": wl.c5_params}[workload](distinct)
    return wl.encode_all(params)


def batch_indices(pkg, workload, descs, batch_n, world, rank):
    """which distinct stream each image of THIS rank's batch is"""
    if workload == "c5":
        mine, _total = pkg.workloads.c5_shard(descs, world, rank, per_gpu=batch_n)
        return mine
    return [i % len(descs) for i in range(batch_n)]


# ---- CPU baseline -----------------------------------------------------------------------------

def _cpu_decode_loop(files, budget_s):
    """jpeg_read_info + jpeg_read_image + image_ycbcr_to_rgb of the unmodified reference
    (src/test.c:23,34,36) through fmemopen, over and over for budget_s seconds"""
    from oracle.oracle import Oracle, Reference
    kind = "reference" if Reference.available() else "port"
    dec = Reference() if kind == "reference" else Oracle()
    pix, t0, n, best = 0, time.perf_counter(), 0, None
    for f in itertools.cycle(files):
        t1 = time.perf_counter()
        if kind == "reference":
            rc, img = dec.decode(f, rgb=True)
        else:
            rc, img = dec.rgb(f, honour_dri=True)
        dt = time.perf_counter() - t1
        if rc == 0:
            pix += img.shape[0] * img.shape[1]
            n += 1
            r = img.shape[0] * img.shape[1] / dt / 1e6
            best = r if best is None else max(best, r)
        if time.perf_counter() - t0 > budget_s or (rc != 0 and n == 0 and time.perf_counter() - t0 > 2):
            break
    return kind, pix, n, time.perf_counter() - t0, best


def _cpu_worker(core, paths, budget_s, q):
    """one process of the all-cores leg: pinned to `core`, its own copy of the reference library"""
    try:
        os.sched_setaffinity(0, {core})
    except OSError:
        pass
    files = [open(p, "rb").read() for p in paths]
    kind, pix, n, el, _ = _cpu_decode_loop(files, budget_s)
    q.put((pix, n, el))


def cpu_baseline(files, budget_s=10.0, all_cores_budget_s=8.0):
    """ONE core, pinned (the headline denominator), then N independent processes, one
    per core this process is allowed, N stated"""
    allowed = sorted(os.sched_getaffinity(0))
    core = allowed[-1]
    os.sched_setaffinity(0, {core})
    try:
        kind, pix, n, el, best = _cpu_decode_loop(files, budget_s)
    finally:
        os.sched_setaffinity(0, set(allowed))
    out = {"value": round(pix / el / 1e6, 3) if n else 0.0, "unit": "Mpixel/s", "cores": 1, "kind": kind,
           "pinned_to_core": core,
           "sample": f"{n} images of the same workload, {el:.1f} s wall on one pinned host core "
                     f"(best single image {best:.2f} Mpixel/s)" if n else "no image decodable by the CPU path"}
    if n and all_cores_budget_s > 0:
        import multiprocessing as mp
        import tempfile
        cores = allowed[:16]     # the GPU boxes give one GPU's process 16 cores
        ctx = mp.get_context("spawn")   # this process has initialised the GPU: no fork
        with tempfile.TemporaryDirectory() as td:
            paths = []
            for i, f in enumerate(files[:8]):
                paths.append(os.path.join(td, f"{i}.jpg"))
                open(paths[-1], "wb").write(f)
            q = ctx.Queue()
            procs = [ctx.Process(target=_cpu_worker, args=(c, paths, all_cores_budget_s, q)) for c in cores]
            t0 = time.perf_counter()
            for p in procs:
                p.start()
            res = [q.get(timeout=all_cores_budget_s * 4 + 120) for _ in procs]
            for p in procs:
                p.join(30)
        tot = sum(r[0] / r[2] for r in res if r[2] > 0)   # every process ran its own budget; rates add
        out["all_cores"] = {"value": round(tot / 1e6, 2), "unit": "Mpixel/s", "cores": len(cores),
                            "sample": f"{sum(r[1] for r in res)} images, {len(cores)} pinned processes x "
                                      f"{all_cores_budget_s:.0f} s ({time.perf_counter() - t0:.0f} s wall with start-up)"}
    return out


# ---- timed decode of one workload ----------------------------------------------------------------

def verify_outputs(pkg, torch, batch, idx, files, n_oracle=8):
    """What the timed steps left in HBM, checked after them (outside the timed region): every copy of
    distinct stream k against the first copy of k on the device (torch.equal over the output views),
    and the LAST copy of up to n_oracle distinct streams against the CPU oracle: quantised coefficients
    bit for bit, RGB to the north star's +-1 (observed: 0).  This covers the plan choices only the
    full-size batches trigger (spans, 8-wave write groups at 4096 x 1080p, config 4 at 512)."""
    from oracle.oracle import Oracle
    torch.cuda.synchronize()
    first, last, bad_copies, compared = {}, {}, 0, 0
    for i, k in enumerate(idx):
        last[k] = i
        if k not in first:
            first[k] = (i, batch.tensor(i))
            continue
        compared += 1
        if not torch.equal(batch.tensor(i), first[k][1]):
            bad_copies += 1
    o = Oracle()
    ks = sorted(first)[:n_oracle]
    bad_oracle, max_d = 0, 0
    for k in ks:
        i = last[k]
        rc, _info, q = o.qcoef(files[k], honour_dri=True)
        rc2, want = o.rgb(files[k], honour_dri=True)
        if rc or rc2:
            bad_oracle += 1
            continue
        d = int(np.abs(batch.pixels(i).astype(np.int16) - want.astype(np.int16)).max())
        max_d = max(max_d, d)
        if d > 1 or not np.array_equal(batch.qcoef(i), q.astype(np.int16)):
            bad_oracle += 1
    return {"images": len(idx), "copies_equal_on_device": compared - bad_copies, "copies_compared": compared,
            "vs_oracle": len(ks), "vs_oracle_mismatches": bad_oracle, "max_abs_rgb_diff": max_d,
            "ok": bad_copies == 0 and bad_oracle == 0}


def reduce_ranks(torch, dist, elapsed, pixels_rank, world, rank, on_cpu):
    """max-over-ranks time and every rank's pixel count (the only cross-rank traffic of the path)"""
    per_rank = [pixels_rank]
    if dist is not None:
        dev = "cpu" if on_cpu else "cuda"
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        p = torch.zeros(world, dtype=torch.float64, device=dev)
        p[rank] = float(pixels_rank)
        dist.all_reduce(p, op=dist.ReduceOp.SUM)
        per_rank = [int(x) for x in p.tolist()]
    return elapsed, per_rank


def run_workload(pkg, torch, dist, name, batch_n, distinct, steps, warmup, layout, world, rank, rehearsal,
                 verify=True, dry_run=False):
    files = make_files(pkg, name, distinct)
    descs = pkg.parse_many(files)
    idx = batch_indices(pkg, name, descs, batch_n, world, rank)
    if dry_run:
        # JD_BENCH_DRYRUN=1 (CPU test of the launcher and the N > 1 control flow): everything but the GPU --
        # files, header parse, this rank's shard, the rendezvous and the reductions
        pixels_rank = sum(descs[i].width * descs[i].height for i in idx)
        if dist is not None:
            dist.barrier()
        elapsed, per_rank = reduce_ranks(torch, dist, 1.0 + rank, pixels_rank, world, rank, True)
        return {"files": files, "elapsed": elapsed, "pixels_rank": pixels_rank, "pixels_per_rank": per_rank,
                "stage_ms": {k: 0.0 for k in pkg.STAGES}, "retries": 0, "blocks": 0,
                "scan_bytes": sum(descs[i].scan_len for i in idx), "n": len(idx), "distinct": len(files),
                "verified": None}
    layout_flag = {"rgb": 0, "rgba": pkg.OUT_4BYTE, "planar": pkg.OUT_PLANAR, "rgb-pitched": pkg.OUT_PITCHED,
                   "rgba-pitched": pkg.OUT_4BYTE | pkg.OUT_PITCHED, "planar-pitched": pkg.OUT_PLANAR | pkg.OUT_PITCHED}[layout]
    batch = pkg.Batch([descs[i] for i in idx], flags=pkg.OUT_RGB | pkg.PROFILE | layout_flag)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    ablation = bool(os.environ.get("JD_BENCH_ABLATION"))  # ablation builds decode garbage on purpose
    for _ in range(warmup):
        batch.decode()
    for i in range(batch.n):
        if batch.status(i) and not ablation:
            raise SystemExit(f"{name}: image {i} failed to decode: status {batch.status(i)}")
    stage_sum = {}
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        batch.decode(sync=True)
        for k, v in batch.stage_ms().items():
            stage_sum[k] = stage_sum.get(k, 0.0) + v
    barrier()
    elapsed = time.perf_counter() - t0
    pixels_rank = batch.total_pixels
    elapsed, per_rank = reduce_ranks(torch, dist, elapsed, pixels_rank, world, rank, rehearsal)
    verified = None
    if verify and layout == "rgb" and not ablation:
        verified = verify_outputs(pkg, torch, batch, idx, files)
        if not verified["ok"]:
            raise SystemExit(f"{name}: the decoded batch FAILED verification: {json.dumps(verified)}")
    res = {"files": files, "elapsed": elapsed, "pixels_rank": pixels_rank, "pixels_per_rank": per_rank,
           "stage_ms": {k: v / steps for k, v in stage_sum.items()}, "retries": batch.retries,
           "blocks": batch.total_blocks, "scan_bytes": batch.total_scan_bytes, "n": batch.n, "distinct": len(files),
           "verified": verified}
    batch.close()
    return res


def stage_rooflines(res, px_bytes):
    coef_b = res["blocks"] * 128
    alg = {  # ALGORITHMIC bytes per launch: per-image figure x images of the batch
        "unstuff": 2 * res["scan_bytes"],
        "sync": res["scan_bytes"],
        "scan": 0,
        "write": res["scan_bytes"] + coef_b,
        "pixel": coef_b + res["pixels_rank"] * px_bytes,
    }

    def roof(stage):
        ms = res["stage_ms"][stage]
        gbs = alg[stage] / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
        return {"kernel": stage, "bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4), "traffic": None,
                "algorithmic_bytes": alg[stage], "avg_ms": round(ms, 4)}
    return alg, roof


def traffic_file():
    """the newest committed PMC traffic summary (profiles/rNN_pmc_traffic.json, tools/pmc_passes.sh)"""
    import glob
    cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))
    return cands[-1] if cands else None


def with_traffic(r, workload, layout, pixels):
    """HBM bytes per launch from the committed PMC passes: measured on the same workload shape at the
    benched launch size (4096 images per launch since round 3; the file says), scaled per pixel to this
    launch; null for workloads the passes did not cover.  `traffic_source` names the file and the commit of the kernels it was measured on, so a
    figure older than the kernels is visible."""
    try:
        path = traffic_file()
        t = json.load(open(path))
        k = t["kernels"].get(r["kernel"])
        if workload == "c3" and layout == "rgb" and k:
            per_px = (2 * k["FETCH_SIZE_KiB"] + k["WRITE_SIZE_KiB"]) * 1024 / (t["images"] * t["pixels_per_image"])
            r["traffic"] = int(per_px * pixels)
            # how busy HBM really is: the bytes the kernel moves (narrow images keep 64-byte units, so
            # fewer than SURVEY's algorithmic bytes) over the same live-measured duration
            if r["avg_ms"] > 0:
                r["achieved_measured"] = round(r["traffic"] / (r["avg_ms"] * 1e-3) / 1e9, 1)
                r["frac_measured"] = round(r["achieved_measured"] / HBM_PEAK_GBS, 4)
            r["traffic_source"] = (f"{os.path.relpath(path, ROOT)} (kernels at commit {t.get('kernel_commit', 'unrecorded')}): "
                                   f"PMC FETCH_SIZE x2 + WRITE_SIZE (KiB) of a {t['images']}-image launch of this workload, "
                                   "scaled per pixel to this launch; not a counter read of this run")
    except (OSError, ValueError, KeyError, TypeError):
        pass
    return r


# ---- PCIe-inclusive rates -------------------------------------------------------------------------

def h2d_rate(src_ptr, nbytes):
    """what the link gives one upload of nbytes on this box, bytes / s (best of 4): the ceiling of any loader.
    Timed the way the library uploads a batch's files -- one synchronous hipMemcpy from pinned memory
    (src_ptr: the arena), through the HIP runtime this process already has loaded; the runtime spreads such a
    copy over its DMA engines, a stream-ordered copy of the same bytes is ~3 % slower.  55 - 59 GB/s on the
    boxes seen."""
    import ctypes
    import re
    import torch
    dst = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    try:
        path = re.search(r"(/\S*libamdhip64\.so\S*)", open("/proc/self/maps").read()).group(1)
        hip = ctypes.CDLL(path)
        hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
        hip.hipMemcpy.restype = ctypes.c_int

        def copy():
            if hip.hipMemcpy(dst.data_ptr(), src_ptr, nbytes, 1):   # hipMemcpyHostToDevice
                raise RuntimeError("hipMemcpy failed")
    except Exception:   # the runtime was not found by name: a stream-ordered copy from torch's own pinned memory
        src = torch.empty(nbytes, dtype=torch.uint8, pin_memory=True)
        src.zero_()

        def copy():
            dst.copy_(src, non_blocking=True)
            torch.cuda.synchronize()
    best = float("inf")
    for _ in range(4):
        t0 = time.perf_counter()
        copy()
        best = min(best, time.perf_counter() - t0)
    del dst
    return nbytes / best


def pcie_inclusive(pkg, files, batch_n, n_batches=6, producers=2, parse_threads=4):
    """host files -> header parse -> jdgpu_batch_create (plan, upload) -> decode.  Unpipelined: one after
    the other.  Pipelined: `producers` threads create batch k+1 (own stream each) while the main thread
    decodes batch k -- what a loader does.  The files of a batch are batch_n SEPARATE copies in host memory
    (1.49 GB for C3), in one region of pinned memory (pkg.Arena = jdgpu_pinned_alloc: a loader reads its
    files into it; filling it is outside the timed region, like the disk read it stands for), so a batch's
    bytes go up in one DMA from where they lie.  `pageable`: the same files as ordinary byte strings, which
    the library gathers into its pinned staging buffer first (16 gather threads)."""
    batch_files = [files[i % len(files)] for i in range(batch_n)]
    arena = pkg.Arena(batch_files)
    total_bytes = sum(len(f) for f in batch_files)
    px = None
    # not overlapped (the first round warms the pinned staging buffer and the buffer cache)
    times = []
    for _ in range(3):
        t0 = time.perf_counter()
        b = pkg.Batch(pkg.parse_many(arena, threads=parse_threads))
        b.decode()
        times.append(time.perf_counter() - t0)
        px = b.total_pixels
        b.close()
    unpipelined = px / min(times[1:]) / 1e6

    def pipelined(make_batch):
        q = queue.Queue()
        alive = threading.Semaphore(4)     # batches in existence (~56 GB of HBM each): two being created, one ready, one decoding
        warm = 4                           # as many batches as may exist: their device buffers come out of hipMalloc
        stop = threading.Event()           # (up to 0.5 s while another batch decodes), every later one out of the cache

        def producer():
            while True:
                alive.acquire()
                if stop.is_set():
                    alive.release()
                    return
                try:
                    q.put(make_batch())
                except Exception as e:  # e.g. out of device memory with too many batches in flight
                    q.put(e)
                    return

        threads = [threading.Thread(target=producer, daemon=True) for _ in range(producers)]
        for t in threads:
            t.start()
        done, timed, pixels, t0 = 0, 0, 0, None
        parts = {k: 0.0 for k in pkg.CREATE_PARTS}
        while timed < n_batches:
            # the clock starts once the warm-up is over AND the consumer has caught up with the producers (nothing
            # ready in the queue): batches created before the clock started would otherwise count as free
            if t0 is None and done >= warm and (q.empty() or done >= warm + 8):
                t0 = time.perf_counter()
            b = q.get()
            if isinstance(b, Exception):
                raise SystemExit(f"pcie_inclusive: a producer failed: {b}")
            b.decode()
            assert not any(b.status(i) for i in (0, b.n // 2, b.n - 1))
            if t0 is not None:
                timed += 1
                pixels += b.total_pixels
                for k, v in b.create_ms().items():
                    parts[k] += v
            b.close()
            alive.release()
            done += 1
        el = time.perf_counter() - t0
        stop.set()
        for _ in threads:
            alive.release()
        for t in threads:
            t.join()
        while not q.empty():               # what the producers had begun when told to stop
            b = q.get()
            if not isinstance(b, Exception):
                b.close()
        return {"value": round(pixels / el / 1e6, 1), "ms_per_batch": round(el / n_batches * 1e3, 2),
                "create_ms": {k: round(v / n_batches, 2) for k, v in parts.items()}}

    host = pipelined(lambda: pkg.Batch(pkg.parse_many(arena, threads=parse_threads)))
    # the same with the headers parsed and the scans picked out on the device (row f1): no host parse at all
    dev = pipelined(lambda: pkg.Batch.from_files(arena))
    pageable = pipelined(lambda: pkg.Batch(pkg.parse_many(batch_files)))
    # the link's rate WHILE the GPU decodes, as in the pipelines above (an idle GPU's clocks make the same copy ~3 % slower)
    busy, stop = pkg.Batch.from_files(arena), threading.Event()

    def keep_decoding():
        while not stop.is_set():
            busy.decode()

    th = threading.Thread(target=keep_decoding, daemon=True)
    th.start()
    time.sleep(0.1)
    rate = h2d_rate(arena.ptr, total_bytes)
    stop.set()
    th.join()
    busy.close()
    arena.close()
    ceiling = px / (total_bytes / rate) / 1e6
    return {"unit": "Mpixel/s", "unpipelined": round(unpipelined, 1), "pipelined": host["value"],
            "pipelined_device_headers": dev["value"], "pipelined_pageable_files": pageable["value"],
            "pcie_ceiling": round(ceiling, 1),
            "pcie_ceiling_note": f"{total_bytes / 1e9:.2f} GB of files per batch at {rate / 1e9:.1f} GB/s H2D (one hipMemcpy of "
                                 f"the arena beside a running decode, timed here) = {total_bytes / rate * 1e3:.1f} ms per batch: no loader can be faster than this",
            "producer_threads": producers, "host_parse_threads": parse_threads, "batches": n_batches, "batch": batch_n,
            "ms_per_batch": {"pipelined": host["ms_per_batch"], "pipelined_device_headers": dev["ms_per_batch"],
                             "pipelined_pageable_files": pageable["ms_per_batch"],
                             "unpipelined": round(min(times[1:]) * 1e3, 2)},
            "create_ms": {"pipelined": host["create_ms"], "pipelined_device_headers": dev["create_ms"],
                          "pipelined_pageable_files": pageable["create_ms"]},
            "includes": "header parse (4 host threads, or on the device), plan, H2D of the files' bytes from pinned host "
                        "memory, scan gather on the device, decode; outputs stay in HBM"}


def git_head():
    try:
        return subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True,
                              timeout=10).stdout.strip() or None
    except (OSError, subprocess.SubprocessError):
        return None


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: start the N ranks as fresh child processes of this
    same command line (this process has not imported torch or touched HIP and never does), one per GPU,
    rendezvous on 127.0.0.1.  Rank 0's stdout (the JSON line) is relayed; the others' goes to stderr.
    Returns the exit status: non-zero if any rank's is."""
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, os.path.abspath(__file__)] + sys.argv[1:]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE if r == 0 else sys.stderr))
    # rank 0's pipe is drained by a thread; the ranks are polled, so that one that dies does not leave the
    # others waiting in a rendezvous or a barrier for their time-outs: they are ended (by PID) at once
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    while True:
        rcs = [p.poll() for p in procs]
        if all(c is not None for c in rcs):
            break
        if any(c for c in rcs):
            for p in procs:
                if p.poll() is None:
                    p.terminate()
            for p in procs:
                try:
                    p.wait(20)
                except subprocess.TimeoutExpired:
                    p.kill()
            rcs = [p.wait() for p in procs]
            break
        time.sleep(0.05)
    reader.join(30)
    sys.stdout.write(b"".join(chunks).decode(errors="replace"))
    sys.stdout.flush()
    bad = [(r, c) for r, c in enumerate(rcs) if c]
    if bad:
        print(f"bench.py: ranks failed (rank, exit status): {bad}", file=sys.stderr)
    return 1 if bad else 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="c3", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=0)
    ap.add_argument("--distinct", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-verify", action="store_true", help="skip the check of the decoded batch after the timed steps")
    ap.add_argument("--lean", action="store_true", help="headline workload only: no pcie_inclusive, no other_workloads")
    ap.add_argument("--layout", default="rgb", choices=["rgb", "rgba", "planar", "rgb-pitched", "rgba-pitched", "planar-pitched"],
                    help="output layout (default: the reference's 3-byte pixel[]; the others are SURVEY 8 row f3)")
    args = ap.parse_args()

    if args.gpus < 1:
        raise SystemExit("--gpus must be at least 1")
    if "WORLD_SIZE" not in os.environ:
        if args.gpus > 1:
            sys.exit(spawn_ranks(args.gpus))
    elif int(os.environ["WORLD_SIZE"]) != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started WORLD_SIZE={os.environ['WORLD_SIZE']} ranks: "
                         "give both the same number (or leave the launcher out: bench.py starts its own ranks)")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if os.environ.get("JD_BENCH_FAIL_RANK") == str(rank):   # test hook: a rank that dies before the rendezvous
        raise SystemExit(f"rank {rank}: JD_BENCH_FAIL_RANK")
    import torch
    dist = None
    # JD_BENCH_REHEARSAL=1: several ranks on ONE GPU over gloo -- exercises the N > 1 control flow
    # on a one-GPU box (the real N > 1 run is one rank per GPU over RCCL).  JD_BENCH_DRYRUN=1: no GPU
    # at all (tests/test_multi_rank.py: launcher, rendezvous, sharding, reductions, JSON line)
    dry_run = bool(os.environ.get("JD_BENCH_DRYRUN"))
    rehearsal = bool(os.environ.get("JD_BENCH_REHEARSAL")) or dry_run
    if rehearsal:
        local_rank = 0
    if world > 1:
        import torch.distributed as dist
        if not dry_run:
            torch.cuda.set_device(local_rank)
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    pkg = ge.load_package()
    if not dry_run:
        pkg.Decoder(local_rank)
    desc, batch_n, distinct = WORKLOADS[args.workload]
    batch_n = args.batch or batch_n
    distinct = min(args.distinct or distinct, batch_n)
    px_bytes = 4 if args.layout.startswith("rgba") else 3

    res = run_workload(pkg, torch, dist, args.workload, batch_n, distinct, args.steps, args.warmup, args.layout,
                       world, rank, rehearsal, verify=not args.no_verify, dry_run=dry_run)

    if rank == 0:
        K = args.steps
        pixels_all = sum(res["pixels_per_rank"])
        ms_step = res["elapsed"] / K * 1e3
        value = pixels_all / (res["elapsed"] / K) / 1e6
        alg, roof = stage_rooflines(res, px_bytes)
        stage_ms = res["stage_ms"]
        dominant = max((s for s in alg if s != "scan"), key=lambda s: stage_ms.get(s, 0.0))
        out = {
            "metric": "decoded Mpixels/s", "value": None if dry_run else round(value, 1), "unit": "Mpixel/s",
            "n_gpus": world, "steps": K, "warmup": args.warmup, "ms_per_step": round(ms_step, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8/i16 entropy decode, f64 IDCT", "data": "synthetic",
            "config": {"workload": desc, "batch_per_gpu": res["n"], "distinct_images": res["distinct"],
                       "pixels_per_step_per_gpu": res["pixels_rank"], "scan_bytes_per_step_per_gpu": res["scan_bytes"],
                       "pixels_per_step_per_rank": res["pixels_per_rank"],
                       "output_layout": args.layout,
                       "parallelism": f"image-sharded x{world}, no collective" +
                                      (", shard_greedy over one global list" if args.workload == "c5" else "")},
            "roofline": with_traffic(roof(dominant), args.workload, args.layout, res["pixels_rank"]),
            "pixel_stage": with_traffic(roof("pixel"), args.workload, args.layout, res["pixels_rank"]),
            "stage_ms": {k: round(v, 4) for k, v in stage_ms.items()},
            # every stage against the HBM roofline by its ALGORITHMIC bytes (DESIGN.md section 4
            # says what each stage is really bound by: only the pixel stage is a bandwidth kernel)
            "stage_rooflines": {s: {"achieved_GBps": roof(s)["achieved"], "frac": roof(s)["frac"]}
                                for s in ("unstuff", "sync", "write", "pixel")},
            "stitch_retries": res["retries"],
            "verified": res["verified"],
        }
        if dry_run:
            out["dry_run"] = True
        if git_head():
            out["commit"] = git_head()
        extras = world == 1 and not args.lean and args.workload == "c3" and not args.batch and not dry_run
        if extras:
            pkg.trim()
            out["pcie_inclusive"] = pcie_inclusive(pkg, res["files"], batch_n)
            pkg.trim()
            others = {}
            for name in ("c2", "c4", "c5"):
                d2, b2, n2 = WORKLOADS[name]
                r2 = run_workload(pkg, torch, None, name, b2, n2, 5, 2, "rgb", 1, 0, False, verify=not args.no_verify)
                _, roof2 = stage_rooflines(r2, 3)
                others[name] = {"workload": d2, "batch": r2["n"], "value": round(r2["pixels_rank"] / (r2["elapsed"] / 5) / 1e6, 1),
                                "unit": "Mpixel/s", "ms_per_step": round(r2["elapsed"] / 5 * 1e3, 3), "steps": 5, "warmup": 2,
                                "stage_ms": {k: round(v, 4) for k, v in r2["stage_ms"].items()},
                                "pixel_stage_frac_of_hbm_peak": roof2("pixel")["frac"], "stitch_retries": r2["retries"],
                                "verified": r2["verified"]}
                pkg.trim()
            out["other_workloads"] = others
        if world == 1 and not args.no_cpu_baseline and not dry_run:
            cpu_files = res["files"]
            if args.workload == "c4":  # the reference aborts on DRI (src/jpeg.c:50): same pixels, no DRI
                cpu_files = [pkg.synth.make_jpeg(3840, 2160, seed=4321 + i, quality=85, subsamp=1) for i in range(2)]
            out["cpu_baseline"] = cpu_baseline(cpu_files)
            if out["cpu_baseline"]["value"]:
                out["gpu_over_cpu_core"] = round(value / out["cpu_baseline"]["value"], 1)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
