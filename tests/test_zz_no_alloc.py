"""GPU (-m gpu), collected last: jdgpu_ycbcr_to_rgb_noalloc -- the one entry point that allocates
nothing (static device scratch) -- against jdgpu_ycbcr_to_rgb, through the C ABI.

This compares two HIP entry points with each other.  What pins the one it compares against:
tests/test_gpu_parity.py::test_exhaustive_colour runs jdgpu_ycbcr_to_rgb (the allocating entry point, same
kernel) over all 2^24 (Y, Cb, Cr) triples against the oracle, which tests/test_oracle_golden.py pins to the
unmodified reference over the same 2^24 inputs.  jdgpu_ycbcr_to_rgb keeps its own hipMalloc'd staging
buffer on purpose, so that this comparison stays one between two different staging paths."""
import ctypes as C

import numpy as np
import pytest

from conftest import load_pkg

pytestmark = pytest.mark.gpu


def test_noalloc_colour_conversion_equals_the_allocating_entry_point():
    pkg = load_pkg()
    L = pkg.lib()
    L.jdgpu_ycbcr_to_rgb_noalloc.argtypes = [C.c_void_p, C.c_uint, C.c_uint]
    rng = np.random.default_rng(77)
    # two calls in a row on the same scratch, the second with a smaller input; then a ragged one
    # (pixel count not a multiple of 4), an empty one, and the largest the scratch holds
    for (w, h) in ((1920, 1080), (200, 137), (3, 1), (0, 5), (4096, 2730)):
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        if w * h:
            special = np.array([(128, 28, 228), (94, 28, 228), (165, 28, 228), (255, 0, 255)], np.uint8)  # the exceptional chroma pair
            img.reshape(-1, 3)[:min(4, w * h)] = special[:min(4, w * h)]
        want = img.copy()
        assert L.jdgpu_ycbcr_to_rgb(want.ctypes.data, w, h) == 0
        got = img.copy()
        assert L.jdgpu_ycbcr_to_rgb_noalloc(got.ctypes.data, w, h) == 0, L.jdgpu_last_error()
        assert np.array_equal(got, want), (w, h)          # integer work: bit-exact
    # the property the entry point exists for: no device allocation.  Free device memory (hipMemGetInfo,
    # through torch, which shares the process's HIP runtime) does not move across warm calls (the library's
    # buffer cache is emptied first, so nothing it hands back hides an allocation)
    import torch
    pkg.trim()
    img = rng.integers(0, 256, (1080, 1920, 3), dtype=np.uint8)
    assert L.jdgpu_ycbcr_to_rgb_noalloc(img.ctypes.data, 1920, 1080) == 0
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    for (w, h) in ((1920, 1080), (640, 480), (4096, 2730)):
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        assert L.jdgpu_ycbcr_to_rgb_noalloc(img.ctypes.data, w, h) == 0
        assert torch.cuda.mem_get_info()[0] == free0, (w, h)
    assert L.jdgpu_noalloc_max_bytes() == 32 << 20
    assert 4096 * 2730 * 3 <= 32 << 20 < 4096 * 2731 * 3
    big = np.zeros((2731, 4096, 3), np.uint8)
    keep = big.copy()
    assert L.jdgpu_ycbcr_to_rgb_noalloc(big.ctypes.data, 4096, 2731) == -12      # JDGPU_ERR_ARG: beyond the scratch
    assert b"static scratch" in L.jdgpu_last_error() and np.array_equal(big, keep)
    assert L.jdgpu_ycbcr_to_rgb_noalloc(None, 8, 8) == -12
