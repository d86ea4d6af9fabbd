"""CPU: the drop-in boundary (include/*.h, libjpegamd.so) and the host-side logic, no GPU
compute.  Mirrors what the reference's callers rely on (src/test.c:21-45) and SURVEY.md 8(b)."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, all_golden_names, golden_stream, load_pkg

INC = os.path.join(ROOT, "include")


@pytest.fixture(scope="module")
def pkg():
    return load_pkg()


def declared_functions():
    names = set()
    for h in ("jpeg.h", "image.h", "jpeg_gpu.h"):
        src = open(os.path.join(INC, h)).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        names |= set(re.findall(r"\b((?:jdgpu|jpeg|image)_[a-z_0-9]+)\s*\(", src))
    names.discard("image_size")  # macro
    return names


def test_library_exports_every_declared_symbol(pkg):
    L = C.CDLL(pkg.LIB_PATH)
    decl = declared_functions()
    assert decl == set(pkg.EXPORTS), decl ^ set(pkg.EXPORTS)
    for name in decl:
        assert hasattr(L, name), f"{name} declared in include/ but not exported"


def test_struct_layout_matches_reference(tmp_path):
    """sizes/offsets recorded in SURVEY.md 8(b) for the reference's jpeg.h (measured there)."""
    prog = tmp_path / "layout.c"
    prog.write_text('#include <stddef.h>\n#include <stdio.h>\n#include "jpeg.h"\n'
                    'int main(void){printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu\\n", sizeof(jpeg_info),'
                    'offsetof(jpeg_info, quantizationTables), offsetof(jpeg_info, huffmanTables),'
                    'offsetof(jpeg_info, width), offsetof(jpeg_info, height), offsetof(jpeg_info, precision),'
                    'sizeof(scan_info), sizeof(huffnode), sizeof(pixel));return 0;}\n')
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-std=c99", "-I", INC, str(prog), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split()
    assert list(map(int, out)) == [1224, 768, 1024, 1216, 1218, 1220, 15, 24, 3]


@pytest.mark.parametrize("name", all_golden_names())
def test_header_parse_matches_oracle(pkg, oracle, name):
    data = golden_stream(name)
    d = pkg.parse(data)
    rc, info = oracle.parse(data)
    assert rc == 0
    assert (d.width, d.height, d.ncomp) == (info.width, info.height, info.ncomp)
    assert d.restart_interval == info.restart_interval
    assert d.scan_len == len(data) - info.scan_offset
    for i in range(d.ncomp):
        cid = info.scomp[i].id
        assert (d.comp[i].id, d.comp[i].Td, d.comp[i].Ta) == (cid, info.scomp[i].Td, info.scomp[i].Ta)
        assert (d.comp[i].H, d.comp[i].V, d.comp[i].Tq) == (info.comp[cid].H, info.comp[cid].V, info.comp[cid].Tq)
    for t in range(4):
        assert bytes(d.quant[t]) == bytes(info.quant[t])
    for c in range(2):
        for t in range(4):
            assert bool(d.huff_defined[c][t]) == bool(info.huff[c][t].defined)
            if info.huff[c][t].defined:
                n = info.huff[c][t].nsymbols
                assert bytes(d.huff[c][t].counts) == bytes(info.huff[c][t].counts)
                assert bytes(d.huff[c][t].symbols)[:n] == bytes(info.huff[c][t].symbols)[:n]


def test_header_rejections(pkg, oracle):
    """same -1 cases as the reference's parsers (src/jpeg.c:25-54, src/parsing.c)"""
    good = golden_stream("syn_8x8_444_q85")
    cases = []
    b = bytearray(good); b[1] = 0xD9; cases.append(bytes(b))                 # not SOI
    cases.append(good[:40])                                                   # short payload read
    i = good.index(b"\xff\xc0")
    b = bytearray(good); b[i + 4] = 12; cases.append(bytes(b))               # precision 12
    b = bytearray(good); b[i + 5] = b[i + 6] = 0; cases.append(bytes(b))     # height 0 (DNL)
    b = bytearray(good); b[i + 11] = 0x51; cases.append(bytes(b))            # H sampling 5
    b = bytearray(good); b[i + 12] = 4; cases.append(bytes(b))               # Tq 4
    j = good.index(b"\xff\xdb")
    b = bytearray(good); b[j + 4] = 0x10; cases.append(bytes(b))             # 16-bit DQT
    k = good.index(b"\xff\xc4")
    b = bytearray(good); b[k + 4] = 0x24; cases.append(bytes(b))             # DHT class 2
    s = good.index(b"\xff\xda")
    b = bytearray(good); b[s + 6] = 0x40; cases.append(bytes(b))             # Td 4
    b = bytearray(good); b[s + 3] = 11; cases.append(bytes(b))               # SOS length mismatch
    for c in cases:
        assert oracle.parse(c)[0] == -1
        with pytest.raises(pkg.JpegError):
            pkg.parse(c)


@pytest.mark.parametrize("threads", [0, 1, 5])
def test_threaded_parse_equals_single(pkg, threads):
    """jdgpu_parse_files over all fixtures (repeated: 300+ files, so several threads run) ==
    jdgpu_parse_file one at a time, descriptor bytes and status; rejected files in between leave a
    zeroed descriptor and are counted"""
    names = all_golden_names()
    good = [golden_stream(n) for n in names] * 20
    bad = bytearray(good[0]); bad[1] = 0xD9
    files = good[:7] + [bytes(bad)] + good[7:100] + [b""] + good[100:] + [good[3][:40]]
    got = pkg.parse_many(files, threads=threads, strict=False)
    assert len(got) == len(files)
    rejected = [i for i, st in enumerate(got.status) if st]
    assert rejected == [7, 101, len(files) - 1]
    assert all(got.status[i] == pkg.ERR_HEADER for i in rejected)
    size = C.sizeof(pkg.ImageDesc)
    for i, f in enumerate(files):
        raw = C.string_at(C.addressof(got[i]), size)
        if i in rejected:
            assert raw == bytes(size)
            continue
        one = pkg.parse(f)
        assert one.scan_len == got[i].scan_len and got[i].scan_len > 0
        # everything but the scan pointer (first 8 bytes: each call aliases its own copy)
        assert raw[8:] == C.string_at(C.addressof(one), size)[8:]
        off = len(f) - got[i].scan_len
        assert C.string_at(got[i].scan, 16) == f[off:off + 16]
    with pytest.raises(pkg.JpegError):
        pkg.parse_many(files, threads=threads)
    assert len(pkg.parse_many([], threads=threads)) == 0


def test_table_redefinitions_last_one_wins(pkg, oracle):
    """a DQT / DHT id defined twice before the scan.  The reference rebuilds the tree
    (src/hufftree.c:44 memsets the root, leaking the old children) and overwrites the quantisation
    table, so the last definition is the one in force; here: descriptor identical to the stream
    without the extra segments, and the unmodified reference decodes both to the same pixels."""
    from oracle.oracle import Reference
    good = golden_stream("syn_64x48_420_q85")
    q = good.index(b"\xff\xdb")
    qlen = int.from_bytes(good[q + 2:q + 4], "big")
    dqt = bytearray(good[q:q + 2 + qlen])
    for k in range(5, 5 + 64):
        dqt[k] = 255 - dqt[k] % 200            # another luma table, same id
    h = good.index(b"\xff\xc4")
    # the second DHT segment (AC luma, 0x10) re-labelled as DC luma (0x00) and put in front
    h2 = good.index(b"\xff\xc4", h + 2)
    h2len = int.from_bytes(good[h2 + 2:h2 + 4], "big")
    dht = bytearray(good[h2:h2 + 2 + h2len])
    assert dht[4] == 0x10
    dht[4] = 0x00
    twice = good[:q] + bytes(dqt) + good[q:h] + bytes(dht) + good[h:]
    a, b = pkg.parse(good), pkg.parse(twice)
    size = C.sizeof(pkg.ImageDesc)
    assert C.string_at(C.addressof(a), size)[8:] == C.string_at(C.addressof(b), size)[8:]
    assert oracle.parse(twice)[0] == 0
    rc1, rgb1 = oracle.rgb(good)
    rc2, rgb2 = oracle.rgb(twice)
    assert rc1 == 0 and rc2 == 0 and np.array_equal(rgb1, rgb2)
    if Reference.available():
        ref = Reference()
        assert np.array_equal(ref.decode(twice)[1], ref.decode(good)[1])


def _canonical_decode(counts, symbols, v16):
    """bit-by-bit canonical walk (the tree of src/hufftree.c:46-59 as first/count per length)"""
    code, first, k = 0, 0, 0
    for L in range(1, 17):
        code = (code << 1) | ((v16 >> (16 - L)) & 1)
        n = counts[L - 1]
        if code - first < n:
            return L, symbols[k + code - first]
        k += n
        first = (first + n) << 1
    return None


def test_flattened_tables_equal_tree_walk(pkg):
    """every 16-bit window of every distinct table in the fixtures: the flattened lookup the
    kernels use == the bitwise code walk (test pyramid item 1)"""
    L = pkg.hooks()
    seen = {}
    for name in all_golden_names():
        d = pkg.parse(golden_stream(name))
        for c in range(2):
            for t in range(4):
                if d.huff_defined[c][t]:
                    key = (c, bytes(d.huff[c][t].counts), bytes(d.huff[c][t].symbols))
                    seen[key] = d.huff[c][t]
    assert len(seen) >= 8
    ln, run, size = C.c_uint(), C.c_uint(), C.c_uint()
    maxlen = 0
    for (cls, counts, symbols), spec in seen.items():
        step = 1 if max(i + 1 for i in range(16) if counts[i]) > 12 else 3
        for v in range(0, 65536, step):
            rc = L.jdgpu_host_lut_lookup(C.byref(spec), int(cls == 0), v, C.byref(ln), C.byref(run), C.byref(size))
            want = _canonical_decode(counts, symbols, v)
            if want is None:
                assert rc == -1
                continue
            wl, sym = want
            maxlen = max(maxlen, wl)
            assert rc == 0 and ln.value == wl
            if cls == 0:
                assert (run.value, size.value) == (0, sym)
            else:
                assert (run.value, size.value) == (sym >> 4, sym & 15)
    assert maxlen == 16  # the fixtures exercise the long-code path


def _random_table(rng, n_sym, complete):
    """random prefix code with n_sym symbols: split leaves of a random tree, lengths <= 16"""
    lens = [1, 1]
    while len(lens) < n_sym:
        i = int(rng.integers(len(lens)))
        if lens[i] >= 16:
            if all(x >= 16 for x in lens):
                break
            continue
        lens[i] += 1
        lens.append(lens[i])
    if not complete:
        lens.remove(max(lens))          # leave part of the code space unused ("no such code")
    counts = [lens.count(L) for L in range(1, 17)]
    return counts


@pytest.mark.parametrize("seed", range(12))
def test_two_level_tables_random_codes(pkg, seed):
    """random prefix codes (deep, shallow, complete and incomplete): the two-level lookup ==
    the bitwise walk on every 16-bit window, including the consumed length of failing walks
    being reported as failures"""
    rng = np.random.default_rng(seed)
    n_sym = [2, 12, 40, 162, 200, 256][seed % 6]
    counts = _random_table(rng, n_sym, complete=seed % 2 == 0)
    total = sum(counts)
    spec = pkg.HuffSpec()
    syms = list(rng.permutation(256)[:total])
    for i in range(16):
        spec.counts[i] = counts[i]
    for i, sy in enumerate(syms):
        spec.symbols[i] = int(sy)
    L = pkg.hooks()
    ln, run, size = C.c_uint(), C.c_uint(), C.c_uint()
    for v in range(65536):
        rc = L.jdgpu_host_lut_lookup(C.byref(spec), 0, v, C.byref(ln), C.byref(run), C.byref(size))
        want = _canonical_decode(counts, syms, v)
        if want is None:
            assert rc == -1
        else:
            assert rc == 0 and ln.value == want[0]
            assert (run.value, size.value) == (want[1] >> 4, want[1] & 15)


def _ref_reader_bytes(scan):
    """the reference's byte reader (src/bitstream.c:7-18): data bytes in the order they are used"""
    out, i = bytearray(), 0
    while i < len(scan):
        out.append(scan[i])
        i += 2 if scan[i] == 0xFF else 1
    return bytes(out)


def _unstuff(pkg, scan, use_rst=0, n_intervals=1):
    L = pkg.hooks()
    SB = pkg.segment_bytes()
    cap = len(scan) // SB + n_intervals + 2
    clean = np.zeros(cap * SB, np.uint8)
    seg = np.zeros(cap, np.uint32)
    src = np.frombuffer(bytes(scan) + b"\0", np.uint8)
    n = L.jdgpu_host_unstuff(src.ctypes.data_as(C.POINTER(C.c_uint8)), len(scan), use_rst, n_intervals,
                             clean.ctypes.data_as(C.POINTER(C.c_uint8)), clean.size,
                             seg.ctypes.data_as(C.POINTER(C.c_uint32)), seg.size)
    assert n >= 1
    return clean[:n * SB], seg[:n]


def test_unstuff_matches_reference_reader(pkg, oracle):
    for name in all_golden_names():
        data = golden_stream(name)
        off = oracle.parse(data)[1].scan_offset
        scan = data[off:]
        clean, seg = _unstuff(pkg, scan)
        want = _ref_reader_bytes(scan)
        nbits = int((seg & 0xFFFF).sum())
        assert nbits == 8 * len(want)
        assert bytes(clean[:len(want)]) == want
        assert np.all(clean[len(want):] == 0xFF)           # padding
        assert seg[0] >> 31 == 1 and np.all(seg[1:] >> 31 == 0)


def test_unstuff_tail_and_ff_runs(pkg):
    # birb.jpg ends "... FF 00 FF D9": the stuffed byte is the last data byte
    clean, seg = _unstuff(pkg, bytes([0x12, 0xFF, 0x00, 0xFF, 0xD9]))
    assert bytes(clean[:3]) == bytes([0x12, 0xFF, 0xFF]) and (seg[0] & 0xFFFF) == 24
    # FF FF FF 41: first FF data, second dropped, third data, 0x41 dropped
    clean, seg = _unstuff(pkg, bytes([0xFF, 0xFF, 0xFF, 0x41, 0x07]))
    assert bytes(clean[:3]) == bytes([0xFF, 0xFF, 0x07]) and (seg[0] & 0xFFFF) == 24
    # trailing lone FF
    clean, seg = _unstuff(pkg, bytes([0x01, 0xFF]))
    assert bytes(clean[:2]) == bytes([0x01, 0xFF]) and (seg[0] & 0xFFFF) == 16


def test_unstuff_restart_split(pkg):
    SB = pkg.segment_bytes()
    a, b, c = bytes(i % 251 + 1 for i in range(SB + 71)), bytes([0x55, 0xFF, 0x00, 0x66]), b""
    scan = a + b"\xff\xd0" + b + b"\xff\xd1" + c + b"\xff\xd2" + bytes([9, 9])
    clean, seg = _unstuff(pkg, scan, use_rst=1, n_intervals=4)
    firsts = np.nonzero(seg >> 31)[0]
    assert list(firsts) == [0, 2, 3, 4] and len(seg) == 5
    assert list(seg & 0xFFFF) == [SB * 8, 71 * 8, 3 * 8, 0, 16]
    assert bytes(clean[:len(a)]) == a and bytes(clean[2 * SB:2 * SB + 3]) == bytes([0x55, 0xFF, 0x66])
    assert bytes(clean[4 * SB:4 * SB + 2]) == bytes([9, 9])
    # without a restart interval FF D0 is "data FF + dropped byte" like any other (src/bitstream.c:11-14)
    clean2, seg2 = _unstuff(pkg, scan, use_rst=0, n_intervals=1)
    assert bytes(clean2[:int((seg2 & 0xFFFF).sum()) // 8]) == _ref_reader_bytes(scan)
    # more markers than intervals: the marker after the last interval ends the data
    clean3, seg3 = _unstuff(pkg, scan, use_rst=1, n_intervals=2)
    assert int((seg3 >> 31).sum()) == 2 and int((seg3 & 0xFFFF).sum()) == (len(a) + 3) * 8


class _HuffNode(C.Structure):
    pass


_HuffNode._fields_ = [("left", C.POINTER(_HuffNode)), ("right", C.POINTER(_HuffNode)), ("symbol", C.c_uint8),
                      ("hasLeft", C.c_bool), ("hasRight", C.c_bool), ("hasSymbol", C.c_bool)]


class _JpegInfo(C.Structure):
    _fields_ = [("componentInfo", (C.c_uint8 * 3) * 256), ("quantizationTables", (C.c_uint8 * 64) * 4),
                ("huffmanTables", (_HuffNode * 4) * 2), ("width", C.c_ushort), ("height", C.c_ushort),
                ("precision", C.c_ubyte), ("restartInterval", C.c_ushort)]


class _ScanInfo(C.Structure):
    _fields_ = [("componentInfo", (C.c_uint8 * 3) * 4), ("maxH", C.c_ubyte), ("maxV", C.c_ubyte), ("n", C.c_ubyte)]


def _walk(node, bits):
    """src/hufftree.c:64-78 over the public tree"""
    cur = node
    for b in bits:
        if b:
            assert cur.hasRight
            cur = cur.right.contents
        else:
            assert cur.hasLeft
            cur = cur.left.contents
        if cur.hasSymbol:
            return cur.symbol
    return None


def test_jpeg_read_info_through_file_pointer(pkg, oracle, tmp_path):
    """the reference's own entry point signature: FILE* in, jpeg_info/scan_info out, fp left at
    the first entropy-coded byte (src/jpeg.c:55), trees walkable, jpeg_free safe twice."""
    assert C.sizeof(_JpegInfo) == 1224 and C.sizeof(_ScanInfo) == 15
    L = pkg.lib()
    libc = C.CDLL(None)
    libc.fopen.restype = C.c_void_p
    libc.fopen.argtypes = [C.c_char_p, C.c_char_p]
    libc.ftell.argtypes = [C.c_void_p]
    libc.ftell.restype = C.c_long
    libc.fclose.argtypes = [C.c_void_p]
    L.jpeg_read_info.argtypes = [C.POINTER(_JpegInfo), C.POINTER(_ScanInfo), C.c_void_p]
    L.jpeg_free.argtypes = [C.POINTER(_JpegInfo)]
    L.jpeg_free.restype = None
    for name in ("ex_birb", "ex_no_subsampling_lake", "syn_64x48_420_q85_dri4"):
        data = golden_stream(name)
        p = tmp_path / (name + ".jpg")
        p.write_bytes(data)
        fp = libc.fopen(str(p).encode(), b"rb")
        ji, si = _JpegInfo(), _ScanInfo()
        C.memset(C.byref(ji), 0xAB, C.sizeof(ji))      # src/test.c:21 passes an uninitialised struct
        assert L.jpeg_read_info(C.byref(ji), C.byref(si), fp) == 0
        rc, info = oracle.parse(data)
        assert libc.ftell(fp) == info.scan_offset
        assert (ji.width, ji.height, ji.precision, si.n) == (info.width, info.height, 8, info.ncomp)
        assert (si.maxH, si.maxV) == (info.maxH, info.maxV)
        assert ji.restartInterval == info.restart_interval
        # first code of the DC table 0: walk the tree with its bits
        t = info.huff[0][info.scomp[0].Td]
        L0 = next(i + 1 for i in range(16) if t.counts[i])
        assert _walk(ji.huffmanTables[0][info.scomp[0].Td], [0] * L0) == t.symbols[0]
        L.jpeg_free(C.byref(ji))
        L.jpeg_free(C.byref(ji))
        libc.fclose(fp)
    # failure path: jpeg_free after a rejected header must be safe (src/test.c:44-45)
    p = tmp_path / "bad.jpg"
    p.write_bytes(b"\xff\xd8\xff\xc0\x00\x05abc")
    fp = libc.fopen(str(p).encode(), b"rb")
    ji, si = _JpegInfo(), _ScanInfo()
    C.memset(C.byref(ji), 0xAB, C.sizeof(ji))
    assert L.jpeg_read_info(C.byref(ji), C.byref(si), fp) == -1
    L.jpeg_free(C.byref(ji))
    libc.fclose(fp)


def test_rgb_to_ycbcr_host_matches_oracle(pkg, oracle):
    """image_rgb_to_ycbcr (src/image.c:54-66) stays host C; same bytes as the oracle"""
    rng = np.random.default_rng(5)
    img = rng.integers(0, 256, (37, 53, 3), dtype=np.uint8)
    got = img.copy()
    L = pkg.lib()
    L.image_rgb_to_ycbcr.argtypes = [C.c_void_p, C.c_uint, C.c_uint]
    L.image_rgb_to_ycbcr.restype = None
    L.image_rgb_to_ycbcr(got.ctypes.data, 53, 37)
    assert np.array_equal(got, oracle.to_ycbcr(img))


def _have_gpu(pkg):
    return pkg.device_count() > 0


def test_no_gpu_fails_loudly(pkg):
    """the product has no CPU fallback: without a device every decode entry point errors"""
    if _have_gpu(pkg):
        pytest.skip("a GPU is present")
    with pytest.raises(pkg.JpegError, match="no HIP device"):
        pkg.Decoder(0)
    with pytest.raises(pkg.JpegError):
        pkg.Batch([pkg.parse(golden_stream("syn_8x8_444_q85"))])
    with pytest.raises(pkg.JpegError):
        pkg.ycbcr_to_rgb(np.zeros((2, 2, 3), np.uint8))
    # the entry points added in round 2 alike: header parsing on the device, batches from files,
    # the print formatter
    with pytest.raises(pkg.JpegError, match="no HIP device"):
        pkg.Batch.from_files([golden_stream("syn_8x8_444_q85")])
    with pytest.raises(pkg.JpegError, match="no HIP device"):
        pkg.parse_device(4096, [0, 10])
    with pytest.raises(pkg.JpegError, match="no HIP device"):
        pkg.format_image(np.zeros((2, 2, 3), np.uint8), 10, 10, 2)
    exe = os.path.join(ROOT, "jpeg-decoder_amd", "bin", "test")
    r = subprocess.run([exe, os.path.join(ROOT, "tests", "golden", "streams", "ex_birb.jpg")], capture_output=True)
    assert r.returncode != 0 and b"no HIP device" in r.stderr


def test_cli_messages():
    exe = os.path.join(ROOT, "jpeg-decoder_amd", "bin", "test")
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 1 and r.stdout == "No path given.\n"                 # src/test.c:11
    r = subprocess.run([exe, "/nonexistent.jpg"], capture_output=True, text=True)
    assert r.returncode == 1 and r.stdout.startswith("Could not open file: ")   # src/test.c:17
    r = subprocess.run([exe, os.path.join(ROOT, "BASELINE.json")], capture_output=True, text=True)
    assert r.returncode == 1 and r.stdout == "Could not parse file.\n"          # src/test.c:24


def _insert_before(data, marker, extra):
    i = data.index(marker)
    return data[:i] + extra + data[i:]


def with_fill_bytes(data):
    """the same stream with 0xFF fill bytes in front of several markers and two stand-alone
    markers (TEM, RST3) between segments: legal per T.81 B.1.1.2, unreadable for the reference
    (src/jpeg.c:32-37 takes the bytes after any marker word as a length)"""
    out = _insert_before(data, b"\xff\xdb", b"\xff")                 # one fill byte before DQT
    out = _insert_before(out, b"\xff\xc0", b"\xff\xff\xff")          # three before SOF0
    out = _insert_before(out, b"\xff\xc4", b"\xff\x01")              # TEM, no length
    out = _insert_before(out, b"\xff\xda", b"\xff\xd3\xff\xff")      # RST3 + two fill bytes before SOS
    return out


def test_fill_bytes_and_standalone_markers(pkg, oracle):
    """SURVEY 8 row f2, last item: fill bytes / length-less markers in the header loop.  Same
    descriptor as the plain stream; the oracle's twin extension agrees (scan offset moved by the
    inserted bytes); the committed fixture is this construction"""
    for name in ("syn_64x48_420_q85", "ex_no_subsampling_lake", "syn_64x48_420_q85_dri4"):
        plain = golden_stream(name)
        filled = with_fill_bytes(plain)
        assert len(filled) == len(plain) + 10
        a, b = pkg.parse(plain), pkg.parse(filled)
        size = C.sizeof(pkg.ImageDesc)
        assert C.string_at(C.addressof(a), size)[8:] == C.string_at(C.addressof(b), size)[8:]
        assert C.string_at(b.scan, 16) == C.string_at(a.scan, 16)
        rc, info = oracle.parse(filled)
        assert rc == 0 and info.scan_offset == oracle.parse(plain)[1].scan_offset + 10
        rc1, rgb1 = oracle.rgb(plain)
        rc2, rgb2 = oracle.rgb(filled)
        assert rc1 == 0 and rc2 == 0 and np.array_equal(rgb1, rgb2)
    fixture = open(os.path.join(ROOT, "tests", "golden", "streams_ext", "fill_bytes_420.jpg"), "rb").read()
    assert fixture == with_fill_bytes(golden_stream("syn_64x48_420_q85"))
    # fill bytes that run into the end of the file, and a marker word that is not one, still fail
    with pytest.raises(pkg.JpegError):
        pkg.parse(golden_stream("syn_8x8_444_q85")[:20] + b"\xff" * 9)
    assert oracle.parse(golden_stream("syn_8x8_444_q85")[:20] + b"\xff" * 9)[0] == -1


def test_truncated_dht_is_rejected(pkg):
    """a DHT table header near the end of its segment used to make the 16 counts and
    the symbols be read from beyond the segment (up to ~4 KB past the 64 KiB payload buffer for a
    segment of length 0xFFFF).  The counts and the symbols they announce must fit the segment."""
    good = golden_stream("syn_8x8_444_q85")
    k = good.index(b"\xff\xc4")
    ln = int.from_bytes(good[k + 2:k + 4], "big")
    table = good[k + 4:k + 2 + ln]                       # class/id byte + 16 counts + symbols
    # a maximal segment: the first table, then filler tables, then a table header in its last byte
    body = bytearray(table)
    filler = bytes([0x00]) + bytes(16)                   # an empty DC table 0: 17 bytes, no symbols
    while len(body) + len(filler) <= 0xFFFF - 2 - 1:
        body += filler
    body += b"\x00" * (0xFFFF - 2 - 1 - len(body)) if len(body) < 0xFFFF - 3 else b""
    assert pkg.parse(good)                                # sanity
    for tail in (b"\x10", b"\x10" + bytes([1] * 10)):    # header only / counts cut short
        seg = bytes(body[:0xFFFF - 2 - len(tail)]) + tail
        assert len(seg) == 0xFFFF - 2
        bad = good[:k] + b"\xff\xc4\xff\xff" + seg + good[k + 2 + ln:]
        with pytest.raises(pkg.JpegError):
            pkg.parse(bad)
    # counts that announce more symbols than the segment holds
    b = bytearray(good)
    b[k + 5 + 15] = 200                                   # 200 codes of length 16 that are not there
    with pytest.raises(pkg.JpegError):
        pkg.parse(bytes(b))


def test_reference_types_and_tree_functions(pkg, tmp_path):
    """what the reference's jpeg.h:4 drags in (src/hufftree.h:10-27, src/bitstream.h:7-17): the
    types compile, hufftree_parse / hufftree_destroy are exported and build the reference's tree;
    the serial decode primitives are declared but deliberately NOT exported (no CPU decode path)"""
    prog = tmp_path / "tree.c"
    prog.write_text(r'''
#include "jpeg.h"
#include <string.h>
int main(void) {
    bitstream bs; memset(&bs, 0, sizeof bs); bs._bitsLeft = 3; bs._c = 0xA5; bs._fp = NULL;   /* the type is visible */
    huffnode root;                                  /* uninitialised, as a caller of the reference may pass it */
    unsigned char dht[16 + 3] = {0, 1, 2};          /* one code of length 2, two of length 3 */
    dht[16] = 7; dht[17] = 8; dht[18] = 9;
    size_t used = hufftree_parse(&root, dht);
    /* codes: 00 -> 7, 010 -> 8, 011 -> 9 */
    int ok = used == 19 && root.hasLeft && !root.hasRight && root.left->hasLeft && root.left->left->hasSymbol &&
             root.left->left->symbol == 7 && root.left->right->left->symbol == 8 && root.left->right->right->symbol == 9;
    hufftree_destroy(&root);
    printf("%d %zu %d\n", ok, sizeof(bitstream), (int)CLASS_AC);
    return ok ? 0 : 1;
}
''')
    exe = tmp_path / "tree"
    libdir = os.path.dirname(pkg.LIB_PATH)
    subprocess.run(["gcc", "-std=c99", "-Wall", "-I", INC, str(prog), "-L", libdir, "-ljpegamd",
                    f"-Wl,-rpath,{libdir}", "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split()
    assert out == ["1", "16", "1"]
    syms = subprocess.run(["nm", "-D", "--defined-only", pkg.LIB_PATH], capture_output=True, text=True, check=True).stdout
    for name in ("hufftree_parse", "hufftree_destroy"):
        assert f" T {name}" in syms
    for name in ("bitstream_create", "bitstream_get_bit", "bitstream_get_bits", "hufftree_decode_next_symbol",
                 "jdgpu_host_unstuff", "jdgpu_host_lut_lookup"):
        assert name not in syms, f"{name}: a CPU decode primitive / test hook inside the product library"


def test_reference_cli_source_builds_unchanged_against_include(pkg, tmp_path):
    """drop-in check: the reference's own src/test.c, unmodified and read where it lies, compiles
    against include/ and links against libjpegamd.so (this container only: the GPU box has no
    /root/reference).  On a box without a GPU the program then fails loudly."""
    src = "/root/reference/src/test.c"
    if not os.path.exists(src):
        pytest.skip("/root/reference is not present")
    exe = tmp_path / "reftest_on_amd"
    libdir = os.path.dirname(pkg.LIB_PATH)
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-I", INC, src, "-L", libdir, "-ljpegamd",
                    f"-Wl,-rpath,{libdir}", "-o", str(exe)], check=True)
    r = subprocess.run([str(exe)], capture_output=True, text=True)
    assert r.returncode == 1 and r.stdout == "No path given.\n"
    if not _have_gpu(pkg):
        r = subprocess.run([str(exe), os.path.join(ROOT, "tests", "golden", "streams", "ex_birb.jpg")], capture_output=True)
        assert r.returncode != 0 and b"no HIP device" in r.stderr


def test_factor_four_layouts_are_planned(pkg):
    """the reference accepts sampling factors 1..4 (src/parsing.c:42-46); 4:1:1 and its
    relatives parse, and the descriptor carries the factors (decode parity: the GPU suite, against
    the reference's own output in tests/golden)"""
    for name, hv in (("syn_64x48_411_q85", (4, 1)), ("syn_160x64_410_q80", (4, 2)), ("syn_48x64_v4_q85", (1, 4)),
                     ("syn_80x96_2x4_q90", (2, 4))):
        d = pkg.parse(golden_stream(name))
        assert (d.comp[0].H, d.comp[0].V) == hv and (d.comp[1].H, d.comp[1].V) == (1, 1)


def test_first_pass_work_list_and_span_rule(pkg):
    """k_huff_first's work list (jd_plan::fwg): a lane takes `span` consecutive segments, a wave 64 *
    span.  The plan picks the span per batch -- 1 while the batch does not fill the chip's 8192 wave
    slots at one segment per lane, 2 from there, 3 from twice that -- and only for images of 384
    segments and more; whatever it picks, every segment of every image is on the list exactly
    once (checked in C by the test-only hook, csrc/host_hooks.c).  JDGPU_FIRST_SPAN forces a span."""
    import ctypes as C
    H = pkg.hooks()
    big = pkg.synth.make_jpeg(1920, 1080, seed=5, quality=85, subsamp=1)
    rst = pkg.synth.make_jpeg(1920, 1080, seed=6, quality=85, subsamp=1, restart_mcus=120)
    small = pkg.synth.make_jpeg(320, 240, seed=7, quality=85, subsamp=0)
    d3 = pkg.parse_many([big, rst, small])
    sb = pkg.segment_bytes()
    waves_big = -(-(-(-d3[0].scan_len // sb)) // 64)   # an upper bound of the stream's segments, in waves

    def plan(descs):
        n = len(descs)
        arr = (pkg.ImageDesc * n)(*descs)
        span = (C.c_uint32 * n)()
        waves = C.c_uint32()
        rc = H.jdgpu_host_plan_first_list(arr, n, span, C.byref(waves))
        assert rc == 0, rc
        return list(span), waves.value

    for copies, want in ((8, 1), (8192 // waves_big + 8, 2), (2 * 8192 // waves_big + 16, 3)):
        span, waves = plan([d3[i % 2] for i in range(copies)] + [d3[2]])
        assert set(span[0:-1:2]) == {want}, (copies, span[:4])
        assert set(span[1:-1:2]) == {min(want, 2)}, (copies, span[:4])   # restart intervals: a span of 2 at most (round 3)
        assert span[-1] == 1                      # the small image keeps one segment per lane
        assert waves >= copies * (waves_big // want) // 2
    for forced in ("1", "2", "5", "3all"):
        os.environ["JDGPU_FIRST_SPAN"] = forced
        try:
            span, _ = plan([d3[0], d3[1], d3[2]] * 3)
        finally:
            del os.environ["JDGPU_FIRST_SPAN"]
        f = int(forced.rstrip("al"))
        assert span[0] == span[1] == f and span[2] == (f if forced.endswith("all") else 1)
