"""The named workloads of BASELINE.json as parameter lists: what bench.py times and
what the GPU tests decode against the oracle.  Streams come from this repo's own encoder
(synth.py); nothing here is on the decode path.

A workload is (distinct streams, index list): image k of the batch is distinct[index[k]].  Only
parameters (sizes, seeds, qualities) are derived here, so every rank of a multi-GPU run builds the
same global list without exchanging bytes, and takes its shard with shard.shard_greedy.
"""
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np

from . import shard, synth

_HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN_STREAMS = os.path.join(os.path.dirname(_HERE), "tests", "golden", "streams")

C2_NAMES = ["ex_no_subsampling_ferris_wheel", "ex_no_subsampling_lake",
            "ex_no_subsampling_pineapple", "ex_no_subsampling_skyline"]
C5_SIZES = [(320, 240), (640, 480), (1280, 720), (1920, 1080), (200, 200), (240, 320)]
C5_PER_GPU = 8192          # 65536 images over 8 GPUs (BASELINE.json configs[4])


def encode_all(params, workers=16):
    """params: (w, h, seed, quality, subsamp, restart_mcus) -> JPEG bytes, on a few threads"""
    if not params:
        return []
    with ThreadPoolExecutor(max_workers=min(workers, len(params))) as ex:
        return list(ex.map(lambda j: synth.make_jpeg(j[0], j[1], seed=j[2], quality=j[3], subsamp=j[4],
                                                     restart_mcus=j[5]), params))


def c2_files():
    """the reference's four examples/no_subsampling files (copies under tests/golden/streams)"""
    return [open(os.path.join(GOLDEN_STREAMS, n + ".jpg"), "rb").read() for n in C2_NAMES]


def c3_params(distinct=64):
    return [(1920, 1080, 1234 + i, 85, 1, 0) for i in range(distinct)]


def c4_params(distinct=16):
    return [(3840, 2160, 4321 + i, 85, 1, 240) for i in range(distinct)]


def c5_params(distinct=96, seed=65536):
    """distinct streams of config 5: every size of C5_SIZES in both layouts (4:2:0 = 1, 4:4:4 = 0),
    qualities 50..95 seeded; stream i has size i % 6 and layout (i // 6) % 2"""
    rng = np.random.default_rng(seed)
    out = []
    for i in range(distinct):
        w, h = C5_SIZES[i % len(C5_SIZES)]
        out.append((w, h, 777 + i, int(rng.integers(50, 96)), 1 - (i // len(C5_SIZES)) % 2, 0))
    return out


def c5_global_index(total, distinct=96, seed=65536):
    """which distinct stream each of the `total` images of the GLOBAL config-5 list is: sizes and
    layouts drawn uniformly (so 50/50 4:2:0 / 4:4:4), the same on every rank"""
    rng = np.random.default_rng(seed + 1)
    return rng.integers(0, distinct, size=total)


def c5_shard(descs, world, rank, per_gpu=C5_PER_GPU, seed=65536):
    """this rank's part of the global list of per_gpu * world images: indices into `descs` (the
    parsed distinct streams, identical on every rank), balanced by shard.shard_greedy over
    shard.image_cost(scan bytes, pixels).  Returns (index list of the shard, global list length)."""
    total = per_gpu * world
    idx = c5_global_index(total, len(descs), seed)
    unit = [shard.image_cost(d.scan_len, d.width * d.height) for d in descs]
    mine = shard.shard_greedy([unit[i] for i in idx], world, rank)
    return [int(idx[k]) for k in mine], total
