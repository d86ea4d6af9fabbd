"""Synthetic baseline-JPEG workload generator (ctypes over csrc/synth_encoder.c).

Workload tooling for tests and bench.py -- not part of the decode path.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def _lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libjpegsynth.so")
        if not os.path.exists(path):
            subprocess.run(["make", "-s", "-C", _HERE, "libjpegsynth.so"], check=True)
        L = C.CDLL(path)
        L.jsynth_make.restype = C.c_size_t
        L.jsynth_make.argtypes = [C.c_int, C.c_int, C.c_uint64, C.c_int, C.c_int, C.c_int,
                                  C.c_void_p, C.c_size_t]
        L.jsynth_encode_rgb.restype = C.c_size_t
        L.jsynth_encode_rgb.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                        C.c_void_p, C.c_size_t]
        L.jsynth_pattern.restype = None
        L.jsynth_pattern.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_uint64, C.c_double]
        _LIB = L
    return _LIB


def make_jpeg(w, h, seed=1234, quality=85, subsamp=1, restart_mcus=0):
    """Deterministic test-pattern JPEG.  subsamp: 0 = 4:4:4, 1 = 4:2:0, 2 = 4:2:2,
    3 = 4:4:0, 4 = grayscale (one component), 5 = 4:1:1 (luma 4x1), 6 = 4:1:0 (luma 4x2),
    7 = luma 1x4, 8 = luma 2x4; 0x100 | H << 4 | V = luma H x V (factors 1..4), chroma 1x1."""
    cap = w * h * 6 + 65536   # white noise at quality 100 codes to more than the raw size
    buf = np.empty(cap, np.uint8)
    n = _lib().jsynth_make(w, h, seed, quality, subsamp, restart_mcus, buf.ctypes.data, cap)
    assert 0 < n <= cap
    return buf[:n].tobytes()


def encode_rgb(rgb, quality=85, subsamp=1, restart_mcus=0):
    rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
    h, w = rgb.shape[:2]
    cap = w * h * 6 + 65536   # white noise at quality 100 codes to more than the raw size
    buf = np.empty(cap, np.uint8)
    n = _lib().jsynth_encode_rgb(rgb.ctypes.data, w, h, quality, subsamp, restart_mcus,
                                 buf.ctypes.data, cap)
    assert 0 < n <= cap
    return buf[:n].tobytes()


def pattern(w, h, seed=1234, sigma=6.0):
    rgb = np.empty((h, w, 3), np.uint8)
    _lib().jsynth_pattern(rgb.ctypes.data, w, h, seed, sigma)
    return rgb
