""".clp bitstream container: ``b"CLPF"`` + ``<I`` compressed length + one zstd frame of D uint8.

Byte-compatible with ``io/bitstream.py:14-33`` of the reference (same magic, same header, zstd
level 22, ``AssertionError("Bad magic")`` on a foreign file), so stores written by either side load
on the other.  The frame codec is the ``zstandard`` package when it is importable (what the
reference uses) and otherwise the system ``libzstd.so.1`` through ctypes -- the GPU image ships the
latter only; zstd frames are format-stable across both.
"""
from __future__ import annotations

import ctypes
import ctypes.util
import struct
from pathlib import Path

import numpy as np

MAGIC = b"CLPF"
VERSION = 1

try:                                    # the reference's dependency, if present
    import zstandard as _zstd_pkg
except Exception:                       # pragma: no cover - absent in the ROCm image
    _zstd_pkg = None

_so = None


def _libzstd():
    global _so
    if _so is None:
        so = ctypes.CDLL(ctypes.util.find_library("zstd") or "libzstd.so.1")
        so.ZSTD_compressBound.restype = ctypes.c_size_t
        so.ZSTD_compressBound.argtypes = [ctypes.c_size_t]
        so.ZSTD_compress.restype = ctypes.c_size_t
        so.ZSTD_compress.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_int]
        so.ZSTD_decompress.restype = ctypes.c_size_t
        so.ZSTD_decompress.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t]
        so.ZSTD_getFrameContentSize.restype = ctypes.c_ulonglong
        so.ZSTD_getFrameContentSize.argtypes = [ctypes.c_char_p, ctypes.c_size_t]
        so.ZSTD_isError.restype = ctypes.c_uint
        so.ZSTD_isError.argtypes = [ctypes.c_size_t]
        _so = so
    return _so


def _compress(raw: bytes, level: int) -> bytes:
    if _zstd_pkg is not None:
        return _zstd_pkg.ZstdCompressor(level=level).compress(raw)
    so = _libzstd()
    bound = so.ZSTD_compressBound(len(raw))
    buf = ctypes.create_string_buffer(bound)
    n = so.ZSTD_compress(buf, bound, raw, len(raw), level)
    if so.ZSTD_isError(n):
        raise RuntimeError("ZSTD_compress failed")
    return buf.raw[:n]


def _decompress(comp: bytes) -> bytes:
    if _zstd_pkg is not None:
        return _zstd_pkg.ZstdDecompressor().decompress(comp)
    so = _libzstd()
    size = so.ZSTD_getFrameContentSize(comp, len(comp))
    if size >= 0xFFFFFFFFFFFFFFFE:      # ZSTD_CONTENTSIZE_UNKNOWN / _ERROR
        raise RuntimeError("zstd frame carries no content size or is corrupt")
    buf = ctypes.create_string_buffer(max(int(size), 1))
    n = so.ZSTD_decompress(buf, int(size), comp, len(comp))
    if so.ZSTD_isError(n):
        raise RuntimeError("ZSTD_decompress failed")
    return buf.raw[:n]


def write_bitstream(q_bytes: bytes, dim: int, out_path: Path) -> None:
    comp = _compress(bytes(q_bytes), 22)
    with open(out_path, "wb") as f:
        f.write(MAGIC + struct.pack("<I", len(comp)) + comp)


def read_bitstream(in_path: Path) -> np.ndarray:
    blob = Path(in_path).read_bytes()
    assert blob[:4] == MAGIC, "Bad magic"
    (ln,) = struct.unpack_from("<I", blob, 4)
    return np.frombuffer(_decompress(blob[8:8 + ln]), dtype=np.uint8)


def decode_embedding(q: np.ndarray, scale: np.ndarray, zero: np.ndarray) -> np.ndarray:
    """uint8 payload -> L2-normalised (1, D) fp32 vector, exactly as cli/eval.py:57-60 does it."""
    z = q.astype(np.float32) * scale + zero
    z = z[None, :]
    return (z / np.maximum(np.linalg.norm(z, axis=-1, keepdims=True), 1e-9)).astype(np.float32)
