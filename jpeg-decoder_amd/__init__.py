"""jpeg-decoder_amd: MI355X-native baseline-DCT JPEG decode behind the C API of cefqrn/jpeg-decoder.

This Python layer is plumbing over the C ABI (include/jpeg_gpu.h, libjpegamd.so): it loads the
in-tree shared library with ctypes and mirrors the reference's call sequence
(jpeg_read_info -> jpeg_read_image -> image_ycbcr_to_rgb, src/test.c:23-36) for batches.
There is no CPU decode path: without the HIP library or a GPU every decode call raises.
"""
import ctypes as C
import os

import numpy as np

from . import shard, synth, workloads  # noqa: F401  (sharding policy, workload generator, named workloads)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("JPEGAMD_LIB") or os.path.join(_HERE, "libjpegamd.so")  # override: A/B builds only

# status / flag values of include/jpeg_gpu.h
OK = 0
ERR_HEADER = -1
ERR_NO_DEVICE = -10
IMG_EOF, IMG_NOSYMBOL, IMG_UNSUPPORTED, IMG_UNSYNCED, IMG_RANGE = 1, 2, 4, 8, 16
OUT_RGB, OUT_YCBCR, PROFILE, OUT_4BYTE, OUT_PLANAR, OUT_PITCHED, SCAN_ON_DEVICE = 0, 1, 2, 8, 16, 32, 64
STAGES = ("unstuff", "sync", "scan", "write", "pixel", "total")
CREATE_PARTS = ("plan", "stage", "upload", "other")

EXPORTS = [
    "jdgpu_device_count", "jdgpu_init", "jdgpu_last_error", "jdgpu_trim", "jdgpu_parse_file", "jdgpu_parse_files",
    "jdgpu_parse_files_device", "jdgpu_batch_create_from_files", "jdgpu_batch_image_size",
    "jdgpu_batch_create", "jdgpu_batch_decode", "jdgpu_batch_sync", "jdgpu_batch_destroy",
    "jdgpu_batch_size", "jdgpu_batch_status", "jdgpu_batch_total_pixels",
    "jdgpu_batch_total_blocks", "jdgpu_batch_total_scan_bytes", "jdgpu_batch_image_blocks",
    "jdgpu_batch_read_pixels", "jdgpu_batch_pixel_bytes", "jdgpu_batch_pixel_pitch", "jdgpu_batch_device_pixels", "jdgpu_batch_read_coefficients",
    "jdgpu_batch_stage_ms", "jdgpu_batch_retries", "jdgpu_batch_repair_counts", "jdgpu_ycbcr_to_rgb", "jdgpu_ycbcr_to_rgb_noalloc", "jdgpu_noalloc_max_bytes", "jdgpu_rgb_to_ycbcr", "jdgpu_image_format",
    "jdgpu_batch_debug_segments", "jdgpu_batch_debug_states", "jdgpu_segment_bytes", "jdgpu_pinned_alloc", "jdgpu_pinned_free", "jdgpu_batch_create_ms",
    # include/jpeg.h, include/image.h
    "jpeg_read_info", "jpeg_read_image", "jpeg_free",
    "image_print", "image_ycbcr_to_rgb", "image_rgb_to_ycbcr",
]


class HuffSpec(C.Structure):
    _fields_ = [("counts", C.c_uint8 * 16), ("symbols", C.c_uint8 * 256)]


class _Comp(C.Structure):
    _fields_ = [("id", C.c_uint8), ("H", C.c_uint8), ("V", C.c_uint8), ("Tq", C.c_uint8),
                ("Td", C.c_uint8), ("Ta", C.c_uint8)]


class ImageDesc(C.Structure):
    _fields_ = [("scan", C.c_void_p), ("scan_len", C.c_size_t),
                ("width", C.c_uint16), ("height", C.c_uint16), ("ncomp", C.c_uint8),
                ("comp", _Comp * 4), ("quant", (C.c_uint8 * 64) * 4),
                ("huff", (HuffSpec * 4) * 2), ("huff_defined", (C.c_uint8 * 4) * 2),
                ("restart_interval", C.c_uint16)]


class JpegError(RuntimeError):
    pass


_lib = None


def lib():
    """Load libjpegamd.so (built in-tree by `make -C jpeg-decoder_amd`); fail loudly if absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise JpegError(f"{LIB_PATH} is missing: run __graft_entry__.build() "
                        "(make -C jpeg-decoder_amd); there is no CPU fallback")
    # One HIP runtime per process.  The torch wheel carries its own libamdhip64.so.7 + HSA runtime;
    # whichever copy is loaded first serves every later user of that soname.  Loaded after
    # /opt/rocm's (which this library would pull in on its own), torch finds no GPU -- so where
    # torch is installed it goes first and the library binds to its runtime (measured: same
    # kernel times either way).  Without torch the library loads /opt/rocm's runtime as linked.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    vp, sz, u8p = C.c_void_p, C.c_size_t, C.POINTER(C.c_uint8)
    L.jdgpu_last_error.restype = C.c_char_p
    L.jdgpu_trim.restype = None
    L.jdgpu_parse_file.argtypes = [vp, sz, C.POINTER(ImageDesc)]
    L.jdgpu_parse_files.argtypes = [C.POINTER(vp), C.POINTER(sz), sz, C.POINTER(ImageDesc),
                                    C.POINTER(C.c_int), C.c_uint]
    L.jdgpu_parse_files.restype = C.c_long
    L.jdgpu_batch_create.argtypes = [C.POINTER(ImageDesc), sz, C.c_uint, C.POINTER(vp)]
    L.jdgpu_parse_files_device.argtypes = [vp, C.POINTER(C.c_uint64), sz, C.POINTER(ImageDesc), C.POINTER(C.c_int)]
    L.jdgpu_parse_files_device.restype = C.c_long
    L.jdgpu_batch_create_from_files.argtypes = [C.POINTER(vp), C.POINTER(sz), sz, C.c_uint, C.POINTER(vp), C.POINTER(C.c_int)]
    L.jdgpu_batch_image_size.argtypes = [vp, sz, C.POINTER(C.c_uint), C.POINTER(C.c_uint)]
    for f in ("jdgpu_batch_decode", "jdgpu_batch_sync"):
        getattr(L, f).argtypes = [vp]
    L.jdgpu_batch_destroy.argtypes = [vp]
    L.jdgpu_batch_destroy.restype = None
    L.jdgpu_batch_size.argtypes = [vp]
    L.jdgpu_batch_size.restype = sz
    L.jdgpu_batch_status.argtypes = [vp, sz]
    L.jdgpu_batch_status.restype = C.c_uint32
    for f in ("jdgpu_batch_total_pixels", "jdgpu_batch_total_blocks", "jdgpu_batch_total_scan_bytes"):
        getattr(L, f).argtypes = [vp]
        getattr(L, f).restype = C.c_uint64
    L.jdgpu_batch_image_blocks.argtypes = [vp, sz]
    L.jdgpu_batch_image_blocks.restype = sz
    L.jdgpu_batch_read_pixels.argtypes = [vp, sz, vp]
    L.jdgpu_batch_pixel_bytes.argtypes = [vp, sz]
    L.jdgpu_batch_pixel_bytes.restype = sz
    L.jdgpu_batch_pixel_pitch.argtypes = [vp, sz]
    L.jdgpu_batch_pixel_pitch.restype = sz
    L.jdgpu_batch_device_pixels.argtypes = [vp, sz]
    L.jdgpu_batch_device_pixels.restype = vp
    L.jdgpu_batch_read_coefficients.argtypes = [vp, sz, vp]
    L.jdgpu_batch_stage_ms.argtypes = [vp, C.POINTER(C.c_float)]
    L.jdgpu_batch_retries.argtypes = [vp]
    L.jdgpu_batch_repair_counts.argtypes = [vp, C.POINTER(C.c_uint32), sz]
    L.jdgpu_ycbcr_to_rgb.argtypes = [vp, C.c_uint, C.c_uint]
    L.jdgpu_ycbcr_to_rgb_noalloc.argtypes = [vp, C.c_uint, C.c_uint]
    L.jdgpu_noalloc_max_bytes.restype = sz
    L.jdgpu_rgb_to_ycbcr.argtypes = [vp, C.c_uint, C.c_uint]
    L.jdgpu_segment_bytes.restype = C.c_uint
    L.jdgpu_image_format.argtypes = [vp, C.c_uint, C.c_uint, C.c_uint, C.c_uint, C.c_uint, vp, sz, C.POINTER(sz)]
    L.jdgpu_batch_debug_segments.argtypes = [vp, sz, vp, sz, vp, sz]
    L.jdgpu_batch_debug_segments.restype = C.c_long
    L.jdgpu_batch_debug_states.argtypes = [vp, sz, vp, vp, sz]
    L.jdgpu_batch_debug_states.restype = C.c_long
    L.jdgpu_pinned_alloc.argtypes = [sz]
    L.jdgpu_pinned_alloc.restype = vp
    L.jdgpu_pinned_free.argtypes = [vp]
    L.jdgpu_pinned_free.restype = None
    L.jdgpu_batch_create_ms.argtypes = [vp, C.POINTER(C.c_float)]
    _lib = L
    return L


_hooks = None


def hooks():
    """libjdhooks.so: TEST-ONLY plain-C statements of the table lookup and the unstuff stage
    (csrc/host_hooks.c); not part of the product library."""
    global _hooks
    if _hooks is None:
        H = C.CDLL(os.path.join(_HERE, "libjdhooks.so"))
        u8p, sz = C.POINTER(C.c_uint8), C.c_size_t
        H.jdgpu_host_lut_lookup.argtypes = [C.POINTER(HuffSpec), C.c_int, C.c_uint, C.POINTER(C.c_uint),
                                            C.POINTER(C.c_uint), C.POINTER(C.c_uint)]
        H.jdgpu_host_unstuff.argtypes = [u8p, sz, C.c_uint, C.c_uint, u8p, sz, C.POINTER(C.c_uint32), sz]
        H.jdgpu_host_unstuff.restype = C.c_long
        H.jdgpu_host_plan_first_list.argtypes = [C.POINTER(ImageDesc), sz, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        _hooks = H
    return _hooks


def _err(L, what, rc):
    return JpegError(f"{what} failed ({rc}): {L.jdgpu_last_error().decode(errors='replace')}")


def parse(data):
    """jpeg_read_info over bytes -> ImageDesc (its scan pointer aliases a kept numpy copy)."""
    L = lib()
    buf = np.frombuffer(bytes(data), dtype=np.uint8)
    d = ImageDesc()
    rc = L.jdgpu_parse_file(buf.ctypes.data, buf.size, C.byref(d))
    if rc:
        raise JpegError(f"header rejected ({rc})")
    d._keep = buf
    return d


class DescArray:
    """n descriptors in one ctypes array (what jdgpu_batch_create takes) + the file buffers their
    scan pointers alias + the per-file parse status."""

    def __init__(self, arr, n, keep, status):
        self.arr, self.n, self._keep, self.status = arr, n, keep, status

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        if not 0 <= i < self.n:
            raise IndexError(i)
        return self.arr[i]

    def __iter__(self):
        return (self.arr[i] for i in range(self.n))


class Arena:
    """Files copied back to back (16-byte aligned starts) into ONE region of pinned host memory
    (jdgpu_pinned_alloc): what a loader that reads its files straight into pinned memory holds.  parse_many
    and Batch.from_files take it in place of a list of byte strings; batch creation then uploads from
    where the files lie instead of copying them into the library's staging buffer first."""

    def __init__(self, files):
        L = self._L = lib()
        sizes = [len(f) for f in files]
        total = sum((s + 15) & ~15 for s in sizes) + 16
        self.ptr = L.jdgpu_pinned_alloc(total)
        if not self.ptr:
            raise _err(L, "jdgpu_pinned_alloc", -11)
        self.buf = np.ctypeslib.as_array((C.c_uint8 * total).from_address(self.ptr))
        self.views, at = [], 0
        for f, s in zip(files, sizes):
            self.buf[at:at + s] = np.frombuffer(f, np.uint8)
            self.views.append(self.buf[at:at + s])
            at += (s + 15) & ~15
        n = len(sizes)
        # the (pointer, length) arrays the C calls take, made once: a loader's per-batch Python work is then a few calls
        self.ptrs = (C.c_void_p * max(1, n))(*[v.ctypes.data if v.size else None for v in self.views])
        self.lens = (C.c_size_t * max(1, n))(*sizes)

    def __len__(self):
        return len(self.views)

    def close(self):
        if getattr(self, "ptr", None):
            self.views, self.buf = [], None
            self._L.jdgpu_pinned_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def parse_many(files, threads=0, strict=True):
    """jdgpu_parse_files: the headers of many files on `threads` host threads (0 = per core).
    strict: raise on the first rejected header, as parse() does; otherwise rejected files keep a
    zeroed descriptor (decoded as IMG_UNSUPPORTED) and .status holds the codes.  files: byte strings, or
    an Arena (parsed where they lie: the descriptors' scan pointers then point into pinned memory)."""
    L = lib()
    n = len(files)
    if isinstance(files, Arena):
        keep, ptrs, lens = files, files.ptrs, files.lens
    else:
        keep = [np.frombuffer(bytes(f), dtype=np.uint8) for f in files]
        ptrs = (C.c_void_p * max(1, n))(*[k.ctypes.data if k.size else None for k in keep])
        lens = (C.c_size_t * max(1, n))(*[k.size for k in keep])
    arr = (ImageDesc * max(1, n))()
    status = (C.c_int * max(1, n))()
    bad = L.jdgpu_parse_files(ptrs, lens, n, arr, status, threads)
    if bad < 0:
        raise _err(L, "jdgpu_parse_files", bad)
    if bad and strict:
        i = next(i for i in range(n) if status[i])
        raise JpegError(f"header of file {i} rejected ({status[i]}); {bad} of {n} rejected")
    return DescArray(arr, n, keep, list(status[:n]))


def parse_device(dev_ptr, offsets, keep=None):
    """jdgpu_parse_files_device: headers of len(offsets) - 1 files that lie in device memory from
    dev_ptr (file i = bytes offsets[i] .. offsets[i+1]), parsed by a kernel.  Returns a DescArray
    whose scan pointers are device addresses: Batch(descs, flags | SCAN_ON_DEVICE)."""
    L = lib()
    n = len(offsets) - 1
    offs = (C.c_uint64 * (n + 1))(*[int(o) for o in offsets])
    arr = (ImageDesc * max(1, n))()
    status = (C.c_int * max(1, n))()
    rc = L.jdgpu_parse_files_device(int(dev_ptr), offs, n, arr, status)
    if rc < 0:
        raise _err(L, "jdgpu_parse_files_device", rc)
    return DescArray(arr, n, keep, list(status)[:n])


class Batch:
    """A decode batch resident on one GPU (jdgpu_batch)."""

    @classmethod
    def from_files(cls, files, flags=OUT_RGB):
        """jdgpu_batch_create_from_files: whole files in, headers parsed on the GPU, scans gathered
        on the GPU.  self.file_status: per-file 0 / ERR_HEADER."""
        L = lib()
        n = len(files)
        if isinstance(files, Arena):
            ptrs, lens = files.ptrs, files.lens
        else:
            bufs = [np.frombuffer(f, np.uint8) for f in files]
            ptrs = (C.c_void_p * max(1, n))(*[b.ctypes.data for b in bufs])
            lens = (C.c_size_t * max(1, n))(*[b.size for b in bufs])
        status = (C.c_int * max(1, n))()
        h = C.c_void_p()
        rc = L.jdgpu_batch_create_from_files(ptrs, lens, n, flags, C.byref(h), status)
        if rc:
            raise _err(L, "jdgpu_batch_create_from_files", rc)
        self = cls.__new__(cls)
        self._L, self._h, self.flags, self.n, self.descs = L, h, flags, n, None
        self.file_status = list(status)[:n]
        self._dims = None
        return self

    @property
    def dims(self):
        """(width, height) per image; a batch made from files asks the library on first use"""
        if self._dims is None and self.descs is not None:
            self._dims = [(d.width, d.height) for d in self.descs]
        if self._dims is None:
            self._dims = []
            for i in range(self.n):
                w, hh = C.c_uint(), C.c_uint()
                self._L.jdgpu_batch_image_size(self._h, i, C.byref(w), C.byref(hh))
                self._dims.append((w.value, hh.value))
        return self._dims

    def __init__(self, descs, flags=OUT_RGB):
        L = self._L = lib()
        if isinstance(descs, DescArray):
            self.descs, arr = descs, descs.arr
        else:
            self.descs = list(descs)
            arr = (ImageDesc * max(1, len(self.descs)))(*self.descs)
        h = C.c_void_p()
        rc = L.jdgpu_batch_create(arr, len(self.descs), flags, C.byref(h))
        if rc:
            raise _err(L, "jdgpu_batch_create", rc)
        self._h = h
        self.flags = flags
        self.n = len(self.descs)
        self._dims = None if isinstance(descs, DescArray) else [(d.width, d.height) for d in self.descs]
        self._arr = arr

    def decode(self, sync=True):
        rc = self._L.jdgpu_batch_decode(self._h)
        if rc:
            raise _err(self._L, "jdgpu_batch_decode", rc)
        if sync:
            self.sync()
        return self

    def sync(self):
        rc = self._L.jdgpu_batch_sync(self._h)
        if rc:
            raise _err(self._L, "jdgpu_batch_sync", rc)

    def status(self, i):
        return int(self._L.jdgpu_batch_status(self._h, i))

    def shape(self, i):
        """array shape of image i in the batch's output layout"""
        w, h = self.dims[i]
        if self.flags & OUT_PLANAR:
            return (3, h, w)
        return (h, w, 4 if self.flags & OUT_4BYTE else 3)

    def pitch(self, i):
        """bytes from one output row to the next (of a plane, in the planar layout)"""
        return int(self._L.jdgpu_batch_pixel_pitch(self._h, i))

    def strides(self, i):
        """byte strides that go with shape(i): dense unless the batch was created with OUT_PITCHED"""
        w, h = self.dims[i]
        p = self.pitch(i)
        if self.flags & OUT_PLANAR:
            return (p * h, p, 1)
        return (p, 4 if self.flags & OUT_4BYTE else 3, 1)

    def padded(self, i):
        """the output buffer of image i as the device holds it, row padding included (flat bytes)"""
        out = np.empty(self._L.jdgpu_batch_pixel_bytes(self._h, i), np.uint8)
        rc = self._L.jdgpu_batch_read_pixels(self._h, i, out.ctypes.data)
        if rc:
            raise _err(self._L, "jdgpu_batch_read_pixels", rc)
        return out

    def pixels(self, i):
        buf = self.padded(i)
        if not self.flags & OUT_PITCHED:
            return buf.reshape(self.shape(i))
        return np.lib.stride_tricks.as_strided(buf, self.shape(i), self.strides(i)).copy()

    rgb = pixels

    def qcoef(self, i):
        """quantised coefficients [nblocks, 64] int16, zigzag order, decode order"""
        nb = self._L.jdgpu_batch_image_blocks(self._h, i)
        out = np.empty((nb, 64), np.int16)
        rc = self._L.jdgpu_batch_read_coefficients(self._h, i, out.ctypes.data)
        if rc:
            raise _err(self._L, "jdgpu_batch_read_coefficients", rc)
        return out

    def segments(self, i):
        """(clean stream bytes, per-segment words) the unstuff stage produced for image i"""
        sb = self._L.jdgpu_segment_bytes()
        # capacity: the scan's segments plus one per restart interval (at most 65535 + 1 MCU rows x columns
        # is not known here: a batch made from files keeps no descriptors, so size by the whole batch's scans)
        scan_len = self.descs[i].scan_len if self.descs is not None else self.total_scan_bytes
        cap = scan_len // sb + 70000
        clean = np.empty(cap * sb, np.uint8)
        seg = np.empty(cap, np.uint32)
        n = self._L.jdgpu_batch_debug_segments(self._h, i, clean.ctypes.data, clean.size, seg.ctypes.data, seg.size)
        if n < 0:
            raise _err(self._L, "jdgpu_batch_debug_segments", n)
        return clean[:n * sb].copy(), seg[:n].copy()

    def states(self, i, cap=1 << 20):
        """(start states, exit states) of image i's segments as last decoded (development aid)"""
        a, b = np.empty(cap, np.uint32), np.empty(cap, np.uint32)
        n = self._L.jdgpu_batch_debug_states(self._h, i, a.ctypes.data, b.ctypes.data, cap)
        if n < 0:
            raise _err(self._L, "jdgpu_batch_debug_states", n)
        return a[:n].copy(), b[:n].copy()

    def stage_ms(self):
        ms = (C.c_float * len(STAGES))()
        rc = self._L.jdgpu_batch_stage_ms(self._h, ms)
        if rc:
            raise _err(self._L, "jdgpu_batch_stage_ms", rc)
        return dict(zip(STAGES, [float(x) for x in ms]))

    def create_ms(self):
        """where the creation of this batch spent its host time: plan / stage / upload / other (ms)"""
        ms = (C.c_float * len(CREATE_PARTS))()
        rc = self._L.jdgpu_batch_create_ms(self._h, ms)
        if rc:
            raise _err(self._L, "jdgpu_batch_create_ms", rc)
        return dict(zip(CREATE_PARTS, [float(x) for x in ms]))

    @property
    def total_pixels(self):
        return int(self._L.jdgpu_batch_total_pixels(self._h))

    @property
    def total_blocks(self):
        return int(self._L.jdgpu_batch_total_blocks(self._h))

    @property
    def total_scan_bytes(self):
        return int(self._L.jdgpu_batch_total_scan_bytes(self._h))

    def repair_counts(self, rounds=8):
        """segments listed for a start-state repair in each round of the last pass"""
        out = (C.c_uint32 * rounds)()
        rc = self._L.jdgpu_batch_repair_counts(self._h, out, rounds)
        if rc:
            raise _err(self._L, "jdgpu_batch_repair_counts", rc)
        return list(out)

    @property
    def retries(self):
        return int(self._L.jdgpu_batch_retries(self._h))

    def device_pixels(self, i):
        return int(self._L.jdgpu_batch_device_pixels(self._h, i) or 0)

    def tensor(self, i):
        """image i as a torch uint8 tensor on the GPU that ALIASES the batch's output buffer (no
        copy; valid until the batch is closed or decoded again; synchronise first)."""
        return _device_view(self, self.device_pixels(i), self.shape(i),
                            self.strides(i) if self.flags & OUT_PITCHED else None)

    def dlpack(self, i):
        """image i as a DLPack capsule over the batch's output buffer (row f3 of SURVEY 8: any
        DLPack consumer takes it without a copy; pitched rows come through as strides)"""
        import torch.utils.dlpack
        return torch.utils.dlpack.to_dlpack(self.tensor(i))

    def tensor_all(self):
        """the whole batch as one [n, ...] tensor aliasing the output buffer; only for batches of
        equal-sized images whose byte size is a multiple of 256 (images start on 256 bytes)"""
        if self.n == 0 or len(set(self.dims)) != 1 or self.flags & OUT_PITCHED:
            raise JpegError("tensor_all needs a non-empty batch of equal-sized images with dense rows")
        nbytes = int(np.prod(self.shape(0)))
        if self.n > 1 and self.device_pixels(1) - self.device_pixels(0) != nbytes:
            raise JpegError(f"images of {nbytes} bytes are padded to 256-byte starts: not one dense array")
        return _device_view(self, self.device_pixels(0), (self.n,) + self.shape(0))

    def close(self):
        if getattr(self, "_h", None):
            self._L.jdgpu_batch_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class _DeviceArray:
    """__cuda_array_interface__ over a region of a batch's output buffer (keeps the batch alive)"""

    def __init__(self, owner, ptr, shape, strides=None):
        self._owner = owner
        self.__cuda_array_interface__ = {"shape": tuple(int(x) for x in shape), "typestr": "|u1",
                                         "data": (int(ptr), False), "version": 2,
                                         "strides": None if strides is None else tuple(int(x) for x in strides)}


def _device_view(owner, ptr, shape, strides=None):
    import torch
    if not ptr:
        raise JpegError("the batch has no device output buffer")
    dev = torch.device("cuda", torch.cuda.current_device())
    return torch.as_tensor(_DeviceArray(owner, ptr, shape, strides), device=dev)


class Decoder:
    """Binds the process to one MI355X and decodes lists of JPEG byte strings."""

    def __init__(self, device=0):
        L = self._L = lib()
        rc = L.jdgpu_init(device)
        if rc:
            raise _err(L, "jdgpu_init", rc)
        self.device = device

    def batch(self, files, flags=OUT_RGB):
        return Batch(parse_many(files), flags)

    def decode(self, files, flags=OUT_RGB):
        return self.batch(files, flags).decode()


def ycbcr_to_rgb(img):
    """image_ycbcr_to_rgb (src/image.c:39-51) on the GPU over an (h, w, 3) uint8 array."""
    L = lib()
    out = np.ascontiguousarray(img, dtype=np.uint8).copy()
    h, w = out.shape[:2]
    rc = L.jdgpu_ycbcr_to_rgb(out.ctypes.data, w, h)
    if rc:
        raise _err(L, "jdgpu_ycbcr_to_rgb", rc)
    return out


def rgb_to_ycbcr(img):
    """image_rgb_to_ycbcr (src/image.c:54-66) on the GPU over an (h, w, 3) uint8 array."""
    L = lib()
    out = np.ascontiguousarray(img, dtype=np.uint8).copy()
    h, w = out.shape[:2]
    rc = L.jdgpu_rgb_to_ycbcr(out.ctypes.data, w, h)
    if rc:
        raise _err(L, "jdgpu_rgb_to_ycbcr", rc)
    return out


def format_image(img, max_width, max_height, pixel_width):
    """jdgpu_image_format: the bytes image_print(img, w, h, max_width, max_height, pixel_width)
    writes to stdout, formatted on the GPU.  img: (h, w, 3) uint8."""
    L = lib()
    img = np.ascontiguousarray(img, dtype=np.uint8)
    h, w = img.shape[:2]
    n = C.c_size_t()
    rc = L.jdgpu_image_format(img.ctypes.data, w, h, max_width, max_height, pixel_width, None, 0, C.byref(n))
    if rc:
        raise _err(L, "jdgpu_image_format", rc)
    out = np.empty(max(1, n.value), np.uint8)
    rc = L.jdgpu_image_format(img.ctypes.data, w, h, max_width, max_height, pixel_width, out.ctypes.data, n.value, C.byref(n))
    if rc:
        raise _err(L, "jdgpu_image_format", rc)
    return out[:n.value].tobytes()


def segment_bytes():
    return int(lib().jdgpu_segment_bytes())


def trim():
    """release the device buffers cached from destroyed batches"""
    lib().jdgpu_trim()


def device_count():
    return int(lib().jdgpu_device_count())
