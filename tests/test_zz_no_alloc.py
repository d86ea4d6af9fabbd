"""GPU (-m gpu), collected last: jdgpu_ycbcr_to_rgb_noalloc -- the one entry point that allocates
nothing (static device scratch) -- against jdgpu_ycbcr_to_rgb, through the C ABI."""
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
    assert 4096 * 2730 * 3 <= 32 << 20 < 4096 * 2731 * 3
    big = np.zeros((2731, 4096, 3), np.uint8)
    keep = big.copy()
    assert L.jdgpu_ycbcr_to_rgb_noalloc(big.ctypes.data, 4096, 2731) == -12      # JDGPU_ERR_ARG: beyond the scratch
    assert b"static scratch" in L.jdgpu_last_error() and np.array_equal(big, keep)
    assert L.jdgpu_ycbcr_to_rgb_noalloc(None, 8, 8) == -12
