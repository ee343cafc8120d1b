"""Oracle: numpy restatement of the byte/integer pieces either side of the hot path.

TEST INFRASTRUCTURE (see oracle/__init__.py).

Reference lines followed (relative to /root/reference/src/clip_feature_codec/):
  .clp container   io/bitstream.py:14-33   b"CLPF" + "<I" compressed length + one zstd frame
  z decode         cli/eval.py:57-60       q*scale+zero, then x / max(||x||2, 1e-9)
  uint8 image      eval/metrics.py:16-19   ((img+1)*127.5).clip(0,255).astype(uint8)  (truncation)
  PSNR             eval/metrics.py:22-29   on the uint8 images, inf when identical

``zstandard`` (pinned ``>=0.22`` in the reference's pyproject.toml:18) is not
installed in this image, so the frame codec is the system ``libzstd.so.1``
(1.4.8) through ctypes: the zstd frame format is version-stable, so frames
written by either decode with the other.
"""
from __future__ import annotations

import ctypes
import ctypes.util
import struct
from pathlib import Path

import numpy as np

MAGIC = b"CLPF"
_lib = None


def _zstd():
    global _lib
    if _lib is None:
        name = ctypes.util.find_library("zstd") or "libzstd.so.1"
        lib = ctypes.CDLL(name)
        lib.ZSTD_compressBound.restype = ctypes.c_size_t
        lib.ZSTD_compressBound.argtypes = [ctypes.c_size_t]
        lib.ZSTD_compress.restype = ctypes.c_size_t
        lib.ZSTD_compress.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
        lib.ZSTD_decompress.restype = ctypes.c_size_t
        lib.ZSTD_decompress.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t]
        lib.ZSTD_getFrameContentSize.restype = ctypes.c_ulonglong
        lib.ZSTD_getFrameContentSize.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
        lib.ZSTD_isError.restype = ctypes.c_uint
        lib.ZSTD_isError.argtypes = [ctypes.c_size_t]
        _lib = lib
    return _lib


def zstd_compress(raw: bytes, level: int = 22) -> bytes:
    lib = _zstd()
    cap = lib.ZSTD_compressBound(len(raw))
    dst = ctypes.create_string_buffer(cap)
    n = lib.ZSTD_compress(dst, cap, raw, len(raw), level)
    if lib.ZSTD_isError(n):
        raise RuntimeError("zstd compress failed")
    return dst.raw[:n]


def zstd_decompress(comp: bytes) -> bytes:
    lib = _zstd()
    size = lib.ZSTD_getFrameContentSize(comp, len(comp))
    if size >= (1 << 62):
        raise RuntimeError("zstd frame without a content size")
    dst = ctypes.create_string_buffer(max(int(size), 1))
    n = lib.ZSTD_decompress(dst, int(size), comp, len(comp))
    if lib.ZSTD_isError(n):
        raise RuntimeError("zstd decompress failed")
    return dst.raw[:n]


def write_bitstream(q_bytes: bytes, dim: int, out_path) -> None:
    comp = zstd_compress(bytes(q_bytes), 22)
    Path(out_path).write_bytes(MAGIC + struct.pack("<I", len(comp)) + comp)


def read_bitstream(in_path) -> np.ndarray:
    blob = Path(in_path).read_bytes()
    assert blob[:4] == MAGIC, "Bad magic"
    (ln,) = struct.unpack("<I", blob[4:8])
    return np.frombuffer(zstd_decompress(blob[8:8 + ln]), dtype=np.uint8)


def decode_z(q: np.ndarray, scale: np.ndarray, zero: np.ndarray) -> np.ndarray:
    z = q.astype(np.float32) * scale + zero
    z = z[None, :]
    n = np.linalg.norm(z, axis=-1, keepdims=True)
    return (z / np.maximum(n, 1e-9)).astype(np.float32)


def to_uint8(img: np.ndarray) -> np.ndarray:
    return ((img + 1.0) * 127.5).clip(0, 255).astype(np.uint8)


def psnr(img1: np.ndarray, img2: np.ndarray) -> float:
    a, b = to_uint8(img1), to_uint8(img2)
    mse = np.mean((a.astype(np.float32) - b.astype(np.float32)) ** 2)
    if mse == 0:
        return float("inf")
    return float(20.0 * np.log10(255.0 / np.sqrt(mse)))
