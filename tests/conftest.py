import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_pkg():
    """import jpeg-decoder_amd/ (the directory name is not an identifier) as jpeg_decoder_amd."""
    import __graft_entry__ as ge
    return ge.load_package()


@pytest.fixture(scope="session")
def manifest():
    return json.load(open(os.path.join(GOLD, "manifest.json")))["entries"]


@pytest.fixture(scope="session")
def oracle():
    from oracle.oracle import Oracle
    return Oracle()


def golden_stream(name):
    return open(os.path.join(GOLD, "streams", name + ".jpg"), "rb").read()


def golden_arrays(name):
    z = np.load(os.path.join(GOLD, name + ".npz"))
    return z["coef"], z["ycbcr"], z["rgb"]


def all_golden_names():
    m = json.load(open(os.path.join(GOLD, "manifest.json")))["entries"]
    return sorted(m)


def plain_names():
    m = json.load(open(os.path.join(GOLD, "manifest.json")))["entries"]
    return sorted(n for n in m if not m[n]["dri"])


def dri_names():
    m = json.load(open(os.path.join(GOLD, "manifest.json")))["entries"]
    return sorted(n for n in m if m[n]["dri"])


def fuzzed_streams(oracle, count=120, seed=2024):
    """the four intact fixtures followed by `count` seeded corruptions of their scans: bit flips,
    random bursts, truncations, inserted 0xFF bytes, stray RSTn markers"""
    rng = np.random.default_rng(seed)
    bases = [golden_stream(n) for n in ("syn_200x136_420_q50", "syn_200x136_444_q95",
                                        "ex_no_subsampling_lake", "syn_96x40_422_q80")]
    files = list(bases)
    for t in range(count):
        g = bytearray(bases[t % len(bases)])
        off = oracle.parse(bytes(g))[1].scan_offset
        n = len(g) - off - 2
        kind = t % 5
        pos = off + int(rng.integers(0, max(1, n)))
        if kind == 0:
            g[pos] ^= 1 << int(rng.integers(8))
        elif kind == 1:
            ln = int(rng.integers(2, 64))
            g[pos:pos + ln] = bytes(rng.integers(0, 256, ln, dtype=np.uint8))[:max(0, len(g) - pos)]
        elif kind == 2:
            g = g[:pos]
        elif kind == 3:
            g[pos:pos] = b"\xff" * int(rng.integers(1, 4))
        else:
            g[pos:pos + 2] = bytes([0xFF, 0xD0 + int(rng.integers(8))])
        files.append(bytes(g))
    return files


def with_component_ids(data, ids):
    """the same stream with its three components numbered `ids` (a permutation of 1, 2, 3) in
    SOF0 and SOS: component k's samples then land in channel ids[k] - 1 (src/decoding.c:107)"""
    b = bytearray(data)
    f = data.index(b"\xff\xc0")
    assert b[f + 9] == 3
    for k in range(3):
        b[f + 10 + 3 * k] = ids[k]
    sos = data.index(b"\xff\xda")
    assert b[sos + 4] == 3
    for k in range(3):
        b[sos + 5 + 2 * k] = ids[k]
    return bytes(b)


def with_high_table_ids(data):
    """the same synthetic stream with its quantisation tables under ids 2 and 3 and its Huffman
    tables under ids 2 (luma) and 3 (chroma) instead of 0 and 1 -- every id the headers allow
    (src/parsing.c:42-49, 92-96, 125-126)"""
    b = bytearray(data)
    q = data.index(b"\xff\xdb")
    assert b[q + 4] == 0 and b[q + 4 + 65] == 1
    b[q + 4], b[q + 4 + 65] = 2, 3
    f = data.index(b"\xff\xc0")
    for k in range(b[f + 9]):
        b[f + 12 + 3 * k] += 2
    at = 0
    for _ in range(4):
        at = data.index(b"\xff\xc4", at)
        b[at + 4] += 2
        at += 2
    sos = data.index(b"\xff\xda")
    for k in range(b[sos + 4]):
        b[sos + 6 + 2 * k] += 0x22
    return bytes(b)
