"""ctypes bindings for the TEST-ONLY checkers under oracle/.

* ``Oracle``    -> oracle/liboracle.so   (this repo's CPU restatement, jpeg_oracle.c)
* ``Reference`` -> oracle/_ref/libjpegref.so (the unmodified reference, built by oracle/Makefile
  from /root/reference/src; prebuilt file travels to the GPU box)

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


class HuffTable(C.Structure):
    _fields_ = [("counts", C.c_uint8 * 16), ("symbols", C.c_uint8 * 256),
                ("nsymbols", C.c_int), ("defined", C.c_int)]


class _Comp(C.Structure):
    _fields_ = [("H", C.c_uint8), ("V", C.c_uint8), ("Tq", C.c_uint8)]


class _SComp(C.Structure):
    _fields_ = [("id", C.c_uint8), ("Td", C.c_uint8), ("Ta", C.c_uint8)]


class Info(C.Structure):
    _fields_ = [("width", C.c_uint16), ("height", C.c_uint16), ("precision", C.c_uint8),
                ("comp", _Comp * 256), ("quant", (C.c_uint8 * 64) * 4),
                ("huff", (HuffTable * 4) * 2), ("scomp", _SComp * 4),
                ("maxH", C.c_uint8), ("maxV", C.c_uint8), ("ncomp", C.c_uint8),
                ("restart_interval", C.c_uint32), ("scan_offset", C.c_size_t)]


def build(ref=True):
    """make liboracle.so (and _ref when the reference sources are present)."""
    subprocess.run(["make", "-s", "-C", HERE, "liboracle.so"] + (["ref"] if ref else []),
                   check=True)


class Oracle:
    def __init__(self):
        path = os.path.join(HERE, "liboracle.so")
        if not os.path.exists(path):
            build(ref=False)
        L = self.lib = C.CDLL(path)
        u8p = C.POINTER(C.c_uint8)
        L.ojpeg_parse.argtypes = [u8p, C.c_size_t, C.POINTER(Info)]
        L.ojpeg_block_count.argtypes = [C.POINTER(Info)]
        L.ojpeg_block_count.restype = C.c_size_t
        L.ojpeg_decode_qcoef.argtypes = [C.POINTER(Info), u8p, C.c_size_t,
                                         C.POINTER(C.c_int32), C.c_int]
        L.ojpeg_decode_ycbcr.argtypes = [C.POINTER(Info), u8p, C.c_size_t, u8p, C.c_int]
        L.ojpeg_pixels_from_qcoef.argtypes = [C.POINTER(Info), C.POINTER(C.c_int32), u8p]
        L.ojpeg_pixels_from_qcoef.restype = None
        L.ojpeg_ycbcr_to_rgb.argtypes = [u8p, C.c_uint, C.c_uint]
        L.ojpeg_ycbcr_to_rgb.restype = None
        L.ojpeg_rgb_to_ycbcr.argtypes = [u8p, C.c_uint, C.c_uint]
        L.ojpeg_rgb_to_ycbcr.restype = None
        self.zigzag = np.array((C.c_uint * 64).in_dll(L, "ojpeg_zigzag"), dtype=np.int64).reshape(8, 8)

    @staticmethod
    def _buf(data):
        a = np.frombuffer(bytes(data), dtype=np.uint8)
        return a, a.ctypes.data_as(C.POINTER(C.c_uint8))

    def overruns(self):
        """coefficient positions past 63 seen so far: inputs on which the reference itself has
        undefined behaviour (src/decoding.c:65 has no bound check)"""
        return C.c_ulong.in_dll(self.lib, "ojpeg_overruns").value

    def parse(self, data):
        a, p = self._buf(data)
        info = Info()
        rc = self.lib.ojpeg_parse(p, len(a), C.byref(info))
        return rc, info

    def qcoef(self, data, honour_dri=True):
        """-> (rc, info, int32 [nblocks, 64] quantised zigzag-order coefficients)"""
        a, p = self._buf(data)
        info = Info()
        rc = self.lib.ojpeg_parse(p, len(a), C.byref(info))
        if rc:
            return rc, info, None
        nb = self.lib.ojpeg_block_count(C.byref(info))
        q = np.zeros((nb, 64), dtype=np.int32)
        off = info.scan_offset
        sp = C.cast(C.addressof(p.contents) + off, C.POINTER(C.c_uint8))
        rc = self.lib.ojpeg_decode_qcoef(C.byref(info), sp, len(a) - off,
                                         q.ctypes.data_as(C.POINTER(C.c_int32)), int(honour_dri))
        return rc, info, q

    def dequant_natural(self, info, q):
        """quantised zigzag [nb,64] -> dequantised natural [nb,8,8] int32 in decode order,
        the layout oracle/_ref/refprobe dumps (src/decoding.c:46,65,68-72)."""
        nb = q.shape[0]
        tq = self.block_tq(info, nb)
        quant = np.array([list(info.quant[i]) for i in range(4)], dtype=np.int64)
        deq = q.astype(np.int64) * quant[tq]
        return deq[:, self.zigzag.reshape(-1)].reshape(nb, 8, 8).astype(np.int32)

    @staticmethod
    def block_tq(info, nb):
        per_mcu = []
        for i in range(info.ncomp):
            cid = info.scomp[i].id
            per_mcu += [info.comp[cid].Tq] * (info.comp[cid].H * info.comp[cid].V)
        per_mcu = np.array(per_mcu, dtype=np.int64)
        return np.tile(per_mcu, nb // len(per_mcu))

    def pixels_from_qcoef(self, info, q):
        img = np.zeros((info.height, info.width, 3), dtype=np.uint8)
        q = np.ascontiguousarray(q, dtype=np.int32)
        self.lib.ojpeg_pixels_from_qcoef(C.byref(info), q.ctypes.data_as(C.POINTER(C.c_int32)),
                                         img.ctypes.data_as(C.POINTER(C.c_uint8)))
        return img

    def ycbcr(self, data, honour_dri=True):
        rc, info, q = self.qcoef(data, honour_dri)
        if rc:
            return rc, None
        return 0, self.pixels_from_qcoef(info, q)

    def to_rgb(self, ycc):
        out = np.ascontiguousarray(ycc).copy()
        h, w = out.shape[:2]
        self.lib.ojpeg_ycbcr_to_rgb(out.ctypes.data_as(C.POINTER(C.c_uint8)), w, h)
        return out

    def to_ycbcr(self, rgb):
        out = np.ascontiguousarray(rgb).copy()
        h, w = out.shape[:2]
        self.lib.ojpeg_rgb_to_ycbcr(out.ctypes.data_as(C.POINTER(C.c_uint8)), w, h)
        return out

    def rgb(self, data, honour_dri=True):
        rc, ycc = self.ycbcr(data, honour_dri)
        if rc:
            return rc, None
        return 0, self.to_rgb(ycc)


class Reference:
    """The unmodified reference behind fmemopen (oracle/refbench.c)."""

    def __init__(self):
        path = os.path.join(HERE, "_ref", "libjpegref.so")
        if not os.path.exists(path):
            raise FileNotFoundError(path + " (run make -C oracle ref where /root/reference exists)")
        L = self.lib = C.CDLL(path)
        L.refbench_decode.argtypes = [C.POINTER(C.c_uint8), C.c_size_t, C.POINTER(C.c_uint8),
                                      C.POINTER(C.c_uint), C.POINTER(C.c_uint), C.c_int]

    @staticmethod
    def available():
        return os.path.exists(os.path.join(HERE, "_ref", "libjpegref.so"))

    def decode(self, data, rgb=True, slack_rows=0):
        """NOTE: on a stream the reference cannot decode it calls exit(1) (src/macros.h:8).
        slack_rows: extra rows allocated behind the image -- on odd sizes with subsampling the
        reference stores up to one row + one pixel past width*height (src/decoding.c:104-107)."""
        a = np.frombuffer(bytes(data), dtype=np.uint8)
        p = a.ctypes.data_as(C.POINTER(C.c_uint8))
        w, h = C.c_uint(), C.c_uint()
        rc = self.lib.refbench_decode(p, len(a), None, C.byref(w), C.byref(h), 0)
        if rc:
            return rc, None
        out = np.zeros((h.value + slack_rows, w.value, 3), dtype=np.uint8)
        rc = self.lib.refbench_decode(p, len(a), out.ctypes.data_as(C.POINTER(C.c_uint8)),
                                      C.byref(w), C.byref(h), int(rgb))
        return rc, out[:h.value]
