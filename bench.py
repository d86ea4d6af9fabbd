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
"""
import argparse
import itertools
import json
import os
import queue
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

def run_workload(pkg, torch, dist, name, batch_n, distinct, steps, warmup, layout, world, rank, rehearsal):
    files = make_files(pkg, name, distinct)
    descs = pkg.parse_many(files)
    idx = batch_indices(pkg, name, descs, batch_n, world, rank)
    layout_flag = {"rgb": 0, "rgba": pkg.OUT_4BYTE, "planar": pkg.OUT_PLANAR, "rgb-pitched": pkg.OUT_PITCHED,
                   "rgba-pitched": pkg.OUT_4BYTE | pkg.OUT_PITCHED, "planar-pitched": pkg.OUT_PLANAR | pkg.OUT_PITCHED}[layout]
    batch = pkg.Batch([descs[i] for i in idx], flags=pkg.OUT_RGB | pkg.PROFILE | layout_flag)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(warmup):
        batch.decode()
    for i in range(batch.n):
        if batch.status(i) and not os.environ.get("JD_BENCH_ABLATION"):  # ablation builds decode garbage on purpose
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
    per_rank = [pixels_rank]
    if dist is not None:
        dev = "cpu" if rehearsal else "cuda"
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        p = torch.zeros(world, dtype=torch.float64, device=dev)
        p[rank] = float(pixels_rank)
        dist.all_reduce(p, op=dist.ReduceOp.SUM)
        per_rank = [int(x) for x in p.tolist()]
    res = {"files": files, "elapsed": elapsed, "pixels_rank": pixels_rank, "pixels_per_rank": per_rank,
           "stage_ms": {k: v / steps for k, v in stage_sum.items()}, "retries": batch.retries,
           "blocks": batch.total_blocks, "scan_bytes": batch.total_scan_bytes, "n": batch.n, "distinct": len(files)}
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
    """HBM bytes per launch from the committed PMC passes: measured on the same workload shape at
    256 images per launch, scaled per pixel to this launch; null for workloads the passes did not
    cover.  `traffic_source` names the file and the commit of the kernels it was measured on, so a
    figure older than the kernels is visible."""
    try:
        path = traffic_file()
        t = json.load(open(path))
        k = t["kernels"].get(r["kernel"])
        if workload == "c3" and layout == "rgb" and k:
            per_px = (2 * k["FETCH_SIZE_KiB"] + k["WRITE_SIZE_KiB"]) * 1024 / (t["images"] * t["pixels_per_image"])
            r["traffic"] = int(per_px * pixels)
            r["traffic_source"] = (f"{os.path.relpath(path, ROOT)} (kernels at commit {t.get('kernel_commit', 'unrecorded')}): "
                                   "PMC FETCH_SIZE x2 + WRITE_SIZE (KiB), 256-image launch, scaled per pixel; "
                                   "not a counter read of this run")
    except (OSError, ValueError, KeyError, TypeError):
        pass
    return r


# ---- PCIe-inclusive rates -------------------------------------------------------------------------

def pcie_inclusive(pkg, files, batch_n, n_batches=6, producers=2):
    """host files -> jdgpu_parse_files -> jdgpu_batch_create (plan, gather into pinned staging,
    upload) -> decode.  Unpipelined: one after the other.  Pipelined: `producers` threads parse and
    create batch k+1 (own stream each) while the main thread decodes batch k -- what a loader does."""
    batch_files = [files[i % len(files)] for i in range(batch_n)]
    px = None
    # not overlapped (the first round warms the pinned staging buffer and the buffer cache)
    times = []
    for _ in range(3):
        t0 = time.perf_counter()
        b = pkg.Batch(pkg.parse_many(batch_files))
        b.decode()
        times.append(time.perf_counter() - t0)
        px = b.total_pixels
        b.close()
    unpipelined = px / min(times[1:]) / 1e6
    def pipelined(make_batch):
        q = queue.Queue(maxsize=2)
        todo = list(range(n_batches + 2))
        lock = threading.Lock()

        def producer():
            while True:
                with lock:
                    if not todo:
                        return
                    todo.pop()
                try:
                    q.put(make_batch())
                except Exception as e:  # e.g. out of device memory with too many batches in flight
                    q.put(e)
                    return

        threads = [threading.Thread(target=producer, daemon=True) for _ in range(producers)]
        for t in threads:
            t.start()
        done, pixels, t0 = 0, 0, None
        while done < n_batches + 2:
            b = q.get()
            if isinstance(b, Exception):
                raise SystemExit(f"pcie_inclusive: a producer failed: {b}")
            if done == 2:  # two batches of warm-up
                t0, pixels = time.perf_counter(), 0
            b.decode()
            assert not any(b.status(i) for i in (0, b.n // 2, b.n - 1))
            pixels += b.total_pixels
            b.close()
            done += 1
        el = time.perf_counter() - t0
        for t in threads:
            t.join()
        return pixels, el

    pixels, el = pipelined(lambda: pkg.Batch(pkg.parse_many(batch_files)))
    # the same with the headers parsed and the scans gathered on the device (row f1): whole files uploaded
    pixels_d, el_d = pipelined(lambda: pkg.Batch.from_files(batch_files))
    return {"unit": "Mpixel/s", "unpipelined": round(unpipelined, 1), "pipelined": round(pixels / el / 1e6, 1),
            "pipelined_device_headers": round(pixels_d / el_d / 1e6, 1),
            "producer_threads": producers, "batches": n_batches, "batch": batch_n,
            "ms_per_batch_pipelined": round(el / n_batches * 1e3, 2),
            "ms_per_batch_unpipelined": round(min(times[1:]) * 1e3, 2),
            "includes": "header parse (threads), plan, gather to pinned memory, H2D of the scans, decode; outputs stay in HBM"}


def git_head():
    try:
        return subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True,
                              timeout=10).stdout.strip() or None
    except (OSError, subprocess.SubprocessError):
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="c3", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=0)
    ap.add_argument("--distinct", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--lean", action="store_true", help="headline workload only: no pcie_inclusive, no other_workloads")
    ap.add_argument("--layout", default="rgb", choices=["rgb", "rgba", "planar", "rgb-pitched", "rgba-pitched", "planar-pitched"],
                    help="output layout (default: the reference's 3-byte pixel[]; the others are SURVEY 8 row f3)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    dist = None
    # JD_BENCH_REHEARSAL=1: several ranks on ONE GPU over gloo -- exercises the N > 1 control flow
    # on a one-GPU box (the real N > 1 run is one rank per GPU over RCCL)
    rehearsal = bool(os.environ.get("JD_BENCH_REHEARSAL"))
    if rehearsal:
        local_rank = 0
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    pkg = ge.load_package()
    pkg.Decoder(local_rank)
    desc, batch_n, distinct = WORKLOADS[args.workload]
    batch_n = args.batch or batch_n
    distinct = min(args.distinct or distinct, batch_n)
    px_bytes = 4 if args.layout.startswith("rgba") else 3

    res = run_workload(pkg, torch, dist, args.workload, batch_n, distinct, args.steps, args.warmup, args.layout,
                       world, rank, rehearsal)

    if rank == 0:
        K = args.steps
        pixels_all = sum(res["pixels_per_rank"])
        ms_step = res["elapsed"] / K * 1e3
        value = pixels_all / (res["elapsed"] / K) / 1e6
        alg, roof = stage_rooflines(res, px_bytes)
        stage_ms = res["stage_ms"]
        dominant = max((s for s in alg if s != "scan"), key=lambda s: stage_ms.get(s, 0.0))
        out = {
            "metric": "decoded Mpixels/s", "value": round(value, 1), "unit": "Mpixel/s",
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
        }
        if git_head():
            out["commit"] = git_head()
        extras = world == 1 and not args.lean and args.workload == "c3" and not args.batch
        if extras:
            pkg.trim()
            out["pcie_inclusive"] = pcie_inclusive(pkg, res["files"], batch_n)
            pkg.trim()
            others = {}
            for name in ("c2", "c4", "c5"):
                d2, b2, n2 = WORKLOADS[name]
                r2 = run_workload(pkg, torch, None, name, b2, n2, 5, 2, "rgb", 1, 0, False)
                _, roof2 = stage_rooflines(r2, 3)
                others[name] = {"workload": d2, "batch": r2["n"], "value": round(r2["pixels_rank"] / (r2["elapsed"] / 5) / 1e6, 1),
                                "unit": "Mpixel/s", "ms_per_step": round(r2["elapsed"] / 5 * 1e3, 3), "steps": 5, "warmup": 2,
                                "stage_ms": {k: round(v, 4) for k, v in r2["stage_ms"].items()},
                                "pixel_stage_frac_of_hbm_peak": roof2("pixel")["frac"], "stitch_retries": r2["retries"]}
                pkg.trim()
            out["other_workloads"] = others
        if world == 1 and not args.no_cpu_baseline:
            cpu_files = res["files"]
            if args.workload == "c4":  # the reference aborts on DRI (src/jpeg.c:50): same pixels, no DRI
                cpu_files = [pkg.synth.make_jpeg(3840, 2160, seed=4321 + i, quality=85, subsamp=1) for i in range(2)]
            out["cpu_baseline"] = cpu_baseline(cpu_files)
            if out["cpu_baseline"]["value"]:
                out["gpu_over_cpu_core"] = round(value / out["cpu_baseline"]["value"], 1)
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
