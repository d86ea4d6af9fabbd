#!/usr/bin/env python3
"""bench.py -- decoded Mpixels/s of the MI355X JPEG decode path (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c3|c2|c4|c5] [--batch B]

A "step" is one full decode (unstuff -> Huffman sync/scan/write -> fused pixel kernel) of one
batch of synthetic baseline JPEGs whose scan bytes, tables and descriptors are already
resident in HBM.  N > 1: one process per GPU (torch.distributed / RCCL only for the barrier and
the max-over-ranks reduction -- the path has no data-path collective: images are independent
and every rank decodes its own batch of the same shape, "weak" scaling).

Prints ONE JSON line (rank 0).  `roofline` is for the stage that takes the most time, measured
with HIP events on the decode stream inside the timed steps; `pixel_stage` repeats the same
figures for the fused pixel kernel, which BASELINE.json's metric names explicitly.
`cpu_baseline` is the UNMODIFIED reference (oracle/_ref/libjpegref.so, built from
/root/reference by oracle/Makefile) -- or, where that file is absent, this repo's CPU
restatement -- timed on one host core over a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

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
    "c5": ("synthetic mixed-size 4:2:0/4:4:4 (BASELINE.json configs[4], per-GPU shard)", 8192, 96),
}


def make_files(pkg, workload, distinct, rank):
    synth = pkg.synth
    jobs = []
    if workload == "c3":
        jobs = [(1920, 1080, 1234 + i, 85, 1, 0) for i in range(distinct)]
    elif workload == "c4":
        jobs = [(3840, 2160, 4321 + i, 85, 1, 240) for i in range(distinct)]
    elif workload == "c5":
        rng = np.random.default_rng(65536 + rank)
        sizes = [(320, 240), (640, 480), (1280, 720), (1920, 1080), (200, 200), (240, 320)]
        for i in range(distinct):
            w, h = sizes[int(rng.integers(len(sizes)))]
            jobs.append((w, h, 777 + 1000 * rank + i, int(rng.integers(50, 96)), int(rng.integers(2)), 0))
    elif workload == "c2":
        gold = os.path.join(ROOT, "tests", "golden", "streams")
        names = ["ex_no_subsampling_ferris_wheel", "ex_no_subsampling_lake",
                 "ex_no_subsampling_pineapple", "ex_no_subsampling_skyline"]
        return [open(os.path.join(gold, n + ".jpg"), "rb").read() for n in names]
    with ThreadPoolExecutor(max_workers=min(16, len(jobs))) as ex:
        return list(ex.map(lambda j: synth.make_jpeg(j[0], j[1], seed=j[2], quality=j[3], subsamp=j[4],
                                                     restart_mcus=j[5]), jobs))


def cpu_baseline(files, budget_s=12.0):
    """single-core decode of a bounded sample: jpeg_read_info + jpeg_read_image +
    image_ycbcr_to_rgb of the unmodified reference (src/test.c:23,34,36) through fmemopen"""
    from oracle.oracle import Oracle, Reference
    kind = "reference" if Reference.available() else "port"
    dec = Reference() if kind == "reference" else Oracle()
    pix, t0, n = 0, time.perf_counter(), 0
    best = None
    import itertools
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
    el = time.perf_counter() - t0
    return {"value": round(pix / el / 1e6, 3), "unit": "Mpixel/s", "cores": 1, "kind": kind,
            "sample": f"{n} images of the same workload, {el:.1f} s wall on one host core "
                      f"(best single image {best:.2f} Mpixel/s)" if n else "no image decodable by the CPU path"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="c3", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=0)
    ap.add_argument("--distinct", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--layout", default="rgb", choices=["rgb", "rgba", "planar"],
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
    dec = pkg.Decoder(local_rank)
    desc, batch_n, distinct = WORKLOADS[args.workload]
    batch_n = args.batch or batch_n
    distinct = min(args.distinct or distinct, batch_n)

    # c4 with DRI cannot be decoded by the reference; its CPU baseline uses the non-DRI twin shape
    files = make_files(pkg, args.workload, distinct, rank)
    descs = [pkg.parse(f) for f in files]
    layout_flag = {"rgb": 0, "rgba": pkg.OUT_4BYTE, "planar": pkg.OUT_PLANAR}[args.layout]
    px_bytes = 4 if args.layout == "rgba" else 3
    batch = pkg.Batch([descs[i % len(descs)] for i in range(batch_n)], flags=pkg.OUT_RGB | pkg.PROFILE | layout_flag)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        batch.decode()
    for i in range(batch.n):
        if batch.status(i) and not os.environ.get("JD_BENCH_ABLATION"):  # ablation builds decode garbage on purpose
            raise SystemExit(f"image {i} failed to decode: status {batch.status(i)}")
    stage_sum = {}
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        batch.decode(sync=True)
        for k, v in batch.stage_ms().items():
            stage_sum[k] = stage_sum.get(k, 0.0) + v
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if rehearsal else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    retries = batch.retries

    if rank == 0:
        K = args.steps
        pixels = batch.total_pixels
        ms_step = elapsed / K * 1e3
        value = pixels * world / (elapsed / K) / 1e6
        stage_ms = {k: v / K for k, v in stage_sum.items()}
        coef_b = batch.total_blocks * 128
        alg = {  # ALGORITHMIC bytes per launch: per-image figure x images of the batch
            "unstuff": 2 * batch.total_scan_bytes,
            "sync": batch.total_scan_bytes,
            "scan": 0,
            "write": batch.total_scan_bytes + coef_b,
            "pixel": coef_b + pixels * px_bytes,
        }

        def roof(stage):
            gbs = alg[stage] / (stage_ms[stage] * 1e-3) / 1e9 if stage_ms[stage] > 0 else 0.0
            return {"kernel": stage, "bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4), "traffic": None,
                    "algorithmic_bytes": alg[stage], "avg_ms": round(stage_ms[stage], 4)}

        def with_traffic(r):
            """HBM bytes per launch from the committed PMC passes (profiles/r01_pmc_traffic.json):
            measured on the same workload shape at 256 images per launch, scaled per pixel to this
            launch; null for workloads the passes did not cover"""
            try:
                t = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")))
                k = t["kernels"].get(r["kernel"])
                if args.workload == "c3" and args.layout == "rgb" and k:
                    per_px = (2 * k["FETCH_SIZE_KiB"] + k["WRITE_SIZE_KiB"]) * 1024 / (t["images"] * t["pixels_per_image"])
                    r["traffic"] = int(per_px * pixels)
                    r["traffic_source"] = "PMC FETCH_SIZE x2 + WRITE_SIZE (KiB), 256-image launch, scaled per pixel"
            except (OSError, ValueError, KeyError):
                pass
            return r

        dominant = max((s for s in alg if s != "scan"), key=lambda s: stage_ms.get(s, 0.0))
        out = {
            "metric": "decoded Mpixels/s", "value": round(value, 1), "unit": "Mpixel/s",
            "n_gpus": world, "steps": K, "warmup": args.warmup, "ms_per_step": round(ms_step, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8/i16 entropy decode, f64 IDCT", "data": "synthetic",
            "config": {"workload": desc, "batch_per_gpu": batch_n, "distinct_images": len(files),
                       "pixels_per_step_per_gpu": pixels, "scan_bytes_per_step_per_gpu": batch.total_scan_bytes,
                       "output_layout": args.layout,
                       "parallelism": f"image-sharded x{world}, no collective"},
            "roofline": with_traffic(roof(dominant)),
            "pixel_stage": with_traffic(roof("pixel")),
            "stage_ms": {k: round(v, 4) for k, v in stage_ms.items()},
            "stitch_retries": retries,
        }
        if world == 1 and not args.no_cpu_baseline:
            cpu_files = files
            if args.workload == "c4":  # the reference aborts on DRI (src/jpeg.c:50): same pixels, no DRI
                cpu_files = [pkg.synth.make_jpeg(3840, 2160, seed=4321 + i, quality=85, subsamp=1) for i in range(2)]
            out["cpu_baseline"] = cpu_baseline(cpu_files)
            if out["cpu_baseline"]["value"]:
                out["gpu_over_cpu_core"] = round(value / out["cpu_baseline"]["value"], 1)
        print(json.dumps(out))
    batch.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
