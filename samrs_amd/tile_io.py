"""ctypes binding of ``libsamrs_io.so`` (include/samrs_io.h): PNG read / write for the generation driver's reader and writer
threads.  A ctypes call drops the interpreter lock for its whole duration, so N threads decode / encode N images at once
(a PIL encode re-takes the lock per block; see DESIGN.md section 6 for what that cost).

Replaces, around the hot path of ``Generate Dataset/main_sam_hbox_semantic.py``:
  :114      ``np.array(Image.open(path))``                         -> :func:`read_rgb`
  :212-215  ``Image.fromarray(seg_mask).save(...)`` / ``seg_color`` -> :func:`write_gray` / :func:`write_lut_rgb`

Formats the native decoder does not cover (JPEG, TIFF, 16-bit or interlaced PNG) are decoded by PIL -- a different file
format, not a different result: both produce the file's RGB pixels.  The library itself is required: a missing
``libsamrs_io.so`` is an error, not a silent switch to PIL.
"""
from __future__ import annotations

import ctypes
import os
from typing import Optional, Tuple

import numpy as np

ABI_VERSION = 3
LEVEL_RUNS = -1            # SAMRS_IO_LEVEL_RUNS: zlib run-length strategy
LEVEL_LABELS = -2          # SAMRS_IO_LEVEL_LABELS: the label-aware encoder (class maps and their palette images): the preset of generate
OK, EOPEN, UNSUPPORTED, ECORRUPT, ESIZE, EWRITE, ENOMEM = 0, -1, -2, -3, -4, -5, -6
_NAMES = {EOPEN: "cannot open", UNSUPPORTED: "unsupported PNG variant", ECORRUPT: "corrupt PNG", ESIZE: "bad size",
          EWRITE: "write failed", ENOMEM: "out of memory"}

_lib: Optional[ctypes.CDLL] = None


class TileIOError(OSError):
    def __init__(self, code: int, path: str):
        super().__init__(f"{_NAMES.get(code, 'error %d' % code)}: {path}")
        self.code = code


def library_path() -> str:
    return os.environ.get("SAMRS_IO_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "libsamrs_io.so")


def load_library() -> ctypes.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    path = library_path()
    if not os.path.exists(path):
        raise RuntimeError(f"{path} not found: build it with `python __graft_entry__.py` (make -C samrs_amd/csrc)")
    lib = ctypes.CDLL(path)
    u8p, szt, ip, cp = ctypes.c_void_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_int), ctypes.c_char_p
    lib.samrs_io_abi_version.restype = ctypes.c_int
    for name, args in (("samrs_io_png_info", [cp, ip, ip]),
                       ("samrs_io_png_read_rgb", [cp, u8p, szt, ip, ip]),
                       ("samrs_io_png_decode_rgb", [u8p, szt, u8p, szt, ip, ip]),
                       ("samrs_io_png_write_gray", [cp, u8p, ctypes.c_int, ctypes.c_int, szt, ctypes.c_int]),
                       ("samrs_io_png_write_lut_rgb", [cp, u8p, ctypes.c_int, ctypes.c_int, szt, u8p, ctypes.c_int]),
                       ("samrs_io_png_write_label_pair", [cp, cp, u8p, ctypes.c_int, ctypes.c_int, szt, u8p]),
                       ("samrs_io_png_write_rgb", [cp, u8p, ctypes.c_int, ctypes.c_int, szt, ctypes.c_int])):
        fn = getattr(lib, name)
        fn.argtypes, fn.restype = args, ctypes.c_int
    if lib.samrs_io_abi_version() != ABI_VERSION:
        raise RuntimeError(f"{path}: ABI {lib.samrs_io_abi_version()}, this package expects {ABI_VERSION}; rebuild")
    _lib = lib
    return lib


def _ptr(a: np.ndarray) -> int:
    return a.ctypes.data


def png_size(path: str) -> Tuple[int, int]:
    h, w = ctypes.c_int(), ctypes.c_int()
    rc = load_library().samrs_io_png_info(os.fsencode(path), ctypes.byref(h), ctypes.byref(w))
    if rc != OK:
        raise TileIOError(rc, path)
    return h.value, w.value


def read_rgb(path: str, out: Optional[np.ndarray] = None) -> np.ndarray:
    """The image at `path` as uint8 [H, W, 3] RGB.  PNG (8-bit, non-interlaced) goes through the native decoder, everything
    else through PIL's ``.convert("RGB")``.  `out`: an optional C-contiguous uint8 destination of at least H * W * 3
    elements (e.g. a slice of a pinned staging buffer); the result is a view of it."""
    lib = load_library()
    bpath = os.fsencode(path)
    h, w = ctypes.c_int(), ctypes.c_int()
    rc = lib.samrs_io_png_info(bpath, ctypes.byref(h), ctypes.byref(w))
    if rc == OK:
        n = h.value * w.value * 3
        if out is None:
            dst = np.empty(n, dtype=np.uint8)
        else:
            if out.dtype != np.uint8 or not out.flags.c_contiguous or out.size < n:
                raise ValueError("out must be C-contiguous uint8 with room for H*W*3 bytes")
            dst = out.reshape(-1)
        rc = lib.samrs_io_png_read_rgb(bpath, _ptr(dst), dst.size, ctypes.byref(h), ctypes.byref(w))
        if rc == OK:
            return dst[:n].reshape(h.value, w.value, 3)
    if rc == EOPEN:
        raise FileNotFoundError(path)
    if rc != UNSUPPORTED:
        raise TileIOError(rc, path)
    from PIL import Image
    img = np.asarray(Image.open(path).convert("RGB"))
    if out is not None:
        dst = out.reshape(-1)[:img.size].reshape(img.shape)
        dst[...] = img
        return dst
    return img


def _check2d(a: np.ndarray) -> np.ndarray:
    if a.dtype != np.uint8 or a.ndim != 2:
        raise ValueError("expected a uint8 [H, W] array")
    if a.strides[1] != 1:
        a = np.ascontiguousarray(a)
    return a


def write_gray(path: str, seg: np.ndarray, level: int = LEVEL_RUNS) -> None:
    """8-bit gray PNG of a class map (atomic: written to `path + '.tmp.<pid>'`, then renamed)."""
    seg = _check2d(seg)
    rc = load_library().samrs_io_png_write_gray(os.fsencode(path), _ptr(seg), seg.shape[0], seg.shape[1], seg.strides[0], level)
    if rc != OK:
        raise TileIOError(rc, path)


def write_lut_rgb(path: str, seg: np.ndarray, lut: np.ndarray, level: int = 6) -> None:
    """Truecolour PNG with pixel = lut[seg[y, x]]; `lut` is uint8 [256, 3]."""
    seg = _check2d(seg)
    lut = np.ascontiguousarray(lut, dtype=np.uint8)
    if lut.shape != (256, 3):
        raise ValueError("lut must be [256, 3]")
    rc = load_library().samrs_io_png_write_lut_rgb(os.fsencode(path), _ptr(seg), seg.shape[0], seg.shape[1], seg.strides[0],
                                                   _ptr(lut), level)
    if rc != OK:
        raise TileIOError(rc, path)


def write_label_pair(gray_path: str, color_path: str, seg: np.ndarray, lut: np.ndarray) -> None:
    """``gray/<stem>.png`` and ``color/<stem>.png`` of one class map (main_sam_hbox_semantic.py:212-215) from ONE parse of the label
    runs (the LEVEL_LABELS encoder); same pixels as :func:`write_gray` + :func:`write_lut_rgb`."""
    seg = _check2d(seg)
    lut = np.ascontiguousarray(lut, dtype=np.uint8)
    if lut.shape != (256, 3):
        raise ValueError("lut must be [256, 3]")
    rc = load_library().samrs_io_png_write_label_pair(os.fsencode(gray_path), os.fsencode(color_path), _ptr(seg), seg.shape[0],
                                                      seg.shape[1], seg.strides[0], _ptr(lut))
    if rc != OK:
        raise TileIOError(rc, gray_path + " / " + color_path)


def write_rgb(path: str, img: np.ndarray, level: int = 6) -> None:
    if img.dtype != np.uint8 or img.ndim != 3 or img.shape[2] != 3:
        raise ValueError("expected a uint8 [H, W, 3] array")
    if img.strides[2] != 1 or img.strides[1] != 3:
        img = np.ascontiguousarray(img)
    rc = load_library().samrs_io_png_write_rgb(os.fsencode(path), _ptr(img), img.shape[0], img.shape[1], img.strides[0], level)
    if rc != OK:
        raise TileIOError(rc, path)


def class_lut(palette: np.ndarray) -> np.ndarray:
    """[256, 3] lookup table: class id -> palette colour, everything else (255 = unlabeled) white
    (main_sam_hbox_semantic.py:163 initialises the colour image to 255)."""
    palette = np.asarray(palette, dtype=np.uint8).reshape(-1, 3)
    if len(palette) > 255:
        raise ValueError(f"{len(palette)} palette rows: class ids must stay below 255 (255 = unlabeled, painted white)")
    lut = np.full((256, 3), 255, dtype=np.uint8)
    lut[:len(palette)] = palette
    return lut
