"""Synthetic inputs for the DDIM reconstruction path: key-seeded weights and stores.

No trained checkpoint, store or image ships with the reference, so
every parity and benchmark run uses inputs generated here.  The generator is
*reference-independent*: the parameter list is derived from the architecture
rules of ``models/unet.py:45-79`` / ``models/blocks.py:14-38`` (widths are a
running product of ``ch_mult``), each tensor is drawn from a numpy generator
seeded by the CRC32 of its state-dict key, and nothing from the reference is
imported.  ``out.weight`` / ``out.bias`` are scaled by ``gamma_out`` (default 0.1)
so the 50-step DDIM map is contractive enough to compare implementations
end-to-end (hard part 1).

This module is numpy-only and has no package-relative imports on purpose:
``tests/golden/make_golden.py`` loads it by file path next to the imported
reference package (both packages are called ``clip_feature_codec``).
"""
from __future__ import annotations

import json
import struct
import zlib
from pathlib import Path
from typing import Dict, List, Sequence, Tuple

import numpy as np

ParamSpec = List[Tuple[str, Tuple[int, ...]]]


def _resblock_spec(prefix: str, c: int, cond_dim: int) -> ParamSpec:
    return [
        (f"{prefix}.norm1.weight", (c,)), (f"{prefix}.norm1.bias", (c,)),
        (f"{prefix}.conv1.weight", (c, c, 3, 3)), (f"{prefix}.conv1.bias", (c,)),
        (f"{prefix}.norm2.weight", (c,)), (f"{prefix}.norm2.bias", (c,)),
        (f"{prefix}.conv2.weight", (c, c, 3, 3)), (f"{prefix}.conv2.bias", (c,)),
        (f"{prefix}.film.to_scale.weight", (c, cond_dim)), (f"{prefix}.film.to_scale.bias", (c,)),
        (f"{prefix}.film.to_shift.weight", (c, cond_dim)), (f"{prefix}.film.to_shift.bias", (c,)),
    ]


def unet_param_spec(z_dim: int = 512, base: int = 128, ch_mult: Sequence[int] = (1, 2, 2),
                    time_dim: int = 256, img_ch: int = 3) -> ParamSpec:
    """Ordered (key, shape) list of the CLIPCondUNet state dict.

    Follows the registration order of ``CLIPCondUNet.__init__`` (models/unet.py:45-79):
    192 entries for base=128, ch_mult=(1,2,2); 140 for base=32, ch_mult=(1,2).
    """
    spec: ParamSpec = [
        ("time_proj.0.weight", (time_dim * 4, time_dim)), ("time_proj.0.bias", (time_dim * 4,)),
        ("time_proj.2.weight", (time_dim, time_dim * 4)), ("time_proj.2.bias", (time_dim,)),
        ("z_proj.0.weight", (time_dim, z_dim)), ("z_proj.0.bias", (time_dim,)),
        ("in_conv.weight", (base, img_ch, 3, 3)), ("in_conv.bias", (base,)),
    ]
    ch = base
    for i, m in enumerate(ch_mult):
        spec += _resblock_spec(f"down.{3 * i}", ch, time_dim)
        spec += _resblock_spec(f"down.{3 * i + 1}", ch, time_dim)
        spec += [(f"down.{3 * i + 2}.weight", (ch * m, ch, 3, 3)), (f"down.{3 * i + 2}.bias", (ch * m,))]
        ch *= m
    spec += _resblock_spec("mid1", ch, time_dim)
    spec += _resblock_spec("mid2", ch, time_dim)
    for i, m in enumerate(reversed(list(ch_mult))):
        spec += _resblock_spec(f"up.{3 * i}", ch, time_dim)
        spec += _resblock_spec(f"up.{3 * i + 1}", ch, time_dim)
        # ConvTranspose2d weight layout is (C_in, C_out, 4, 4)
        spec += [(f"up.{3 * i + 2}.weight", (ch, ch // m, 4, 4)), (f"up.{3 * i + 2}.bias", (ch // m,))]
        ch //= m
    spec += [("out_norm.weight", (ch,)), ("out_norm.bias", (ch,)),
             ("out.weight", (img_ch, ch, 3, 3)), ("out.bias", (img_ch,))]
    return spec


def synth_state_dict(spec: ParamSpec, seed: int = 0, gamma_out: float = 0.1) -> Dict[str, np.ndarray]:
    """Key-seeded synthetic fp32 weights for ``spec`` (one independent stream per key)."""
    sd: Dict[str, np.ndarray] = {}
    for name, shape in spec:
        rng = np.random.default_rng((zlib.crc32(name.encode()) + 7919 * seed) & 0xFFFFFFFF)
        g = rng.standard_normal(shape).astype(np.float32)
        leaf = name.rsplit(".", 1)[-1]
        if "norm" in name:
            w = (1.0 + 0.1 * g) if leaf == "weight" else 0.1 * g
        elif leaf == "bias":
            w = 0.02 * g
        else:
            if name.startswith("up.") and len(shape) == 4 and shape[2] == 4:
                fan_in = shape[0] * 4          # each output pixel sees 2x2 of the 4x4 taps
            else:
                fan_in = int(np.prod(shape[1:]))
            # variance of PyTorch's default U(-1/sqrt(fan_in), 1/sqrt(fan_in)) init: 1/(3 fan_in)
            w = g * np.float32(1.0 / np.sqrt(3.0 * fan_in))
        if name.startswith("out."):
            w = w * np.float32(gamma_out)
        sd[name] = np.ascontiguousarray(w, dtype=np.float32)
    return sd


def synth_z(n: int, dim: int = 512, seed: int = 1234) -> np.ndarray:
    """L2-normalised pseudo CLIP vectors, one generator per record."""
    out = np.empty((n, dim), np.float32)
    for i in range(n):
        v = np.random.default_rng(seed + i).standard_normal(dim).astype(np.float32)
        out[i] = v / max(float(np.linalg.norm(v)), 1e-9)
    return out


def synth_image(i: int, size: int, seed: int = 1234) -> np.ndarray:
    """uint8 (size,size,3) image: smooth low-frequency field + N(0,8) noise."""
    rng = np.random.default_rng(seed + i)
    coarse = rng.uniform(0, 255, (8, 8, 3))
    # separable bilinear-ish upsample by interpolation on a regular grid
    src = (np.arange(size) + 0.5) / size * 8 - 0.5
    i0 = np.clip(np.floor(src).astype(int), 0, 7)
    i1 = np.clip(i0 + 1, 0, 7)
    f = np.clip(src - i0, 0, 1)[:, None, None]
    rows = coarse[i0] * (1 - f) + coarse[i1] * f                       # (size, 8, 3)
    fx = np.clip(src - i0, 0, 1)[None, :, None]
    img = rows[:, i0] * (1 - fx) + rows[:, i1] * fx                    # (size, size, 3)
    img = img + rng.normal(0, 8, img.shape)
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


def quantizer_fit(X: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """scale, zero of the per-channel affine uint8 quantiser (codecs/quantizer.py:22-27)."""
    xmin = X.min(axis=0).astype(np.float32)
    xmax = X.max(axis=0).astype(np.float32)
    scale = np.maximum(xmax - xmin, np.float32(1e-8)) / np.float32(255.0)
    return scale.astype(np.float32), xmin


def quantizer_encode(x: np.ndarray, scale: np.ndarray, zero: np.ndarray) -> np.ndarray:
    """round((x-zero)/scale).clamp(0,255) as uint8 (codecs/quantizer.py:32-33)."""
    q = np.rint((x.astype(np.float32) - zero) / scale)   # torch.round == round-half-even == rint
    return np.clip(q, 0, 255).astype(np.uint8)


def write_synth_store(store_dir: Path, n: int, size: int, dim: int = 512, seed: int = 1234,
                      write_clp=None) -> List[dict]:
    """Create a store with the layout cli/encode_images.py:77-85 produces.

    ``write_clp(q_bytes, dim, path)`` is the bitstream writer to use (the build's
    libzstd-backed writer); passed in so this module stays dependency-free.
    """
    from PIL import Image
    store_dir = Path(store_dir)
    (store_dir / "images").mkdir(parents=True, exist_ok=True)
    (store_dir / "bitstreams").mkdir(parents=True, exist_ok=True)
    Z = synth_z(n, dim, seed)
    scale, zero = quantizer_fit(Z)
    manifest = []
    for i in range(n):
        img_p = store_dir / "images" / f"{i:05d}.png"
        Image.fromarray(synth_image(i, size, seed)).save(img_p)
        clp_p = store_dir / "bitstreams" / f"{i:05d}.clp"
        write_clp(quantizer_encode(Z[i], scale, zero).tobytes(), dim, clp_p)
        manifest.append({"image": str(img_p), "bitstream": str(clp_p)})
    np.savez(store_dir / "codec_meta.npz", scale=scale, zero=zero, dim=dim)
    (store_dir / "manifest.json").write_text(json.dumps(manifest, ensure_ascii=False, indent=2), encoding="utf-8")
    return manifest


def start_noise(indices: Sequence[int], size: int, seed_base: int = 0, img_ch: int = 3) -> np.ndarray:
    """Per-image x_T from a CPU torch generator seeded ``seed_base + index``."""
    import torch
    out = []
    for i in indices:
        g = torch.Generator("cpu").manual_seed(int(seed_base) + int(i))
        out.append(torch.randn((img_ch, size, size), generator=g, dtype=torch.float32))
    return torch.stack(out, 0).numpy()
