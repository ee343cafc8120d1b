"""Shared pieces of the two hot-path CLIs: checkpoint loading, z decode, deterministic start noise."""
from __future__ import annotations

import os
from pathlib import Path
from typing import Sequence, Tuple

import numpy as np
import torch

from ..io.bitstream import read_bitstream, decode_embedding
from ..models.unet import CLIPCondUNet, infer_arch
from ..diffusion.scheduler import NoiseScheduler
from ..diffusion.ddim import DDIMSampler


def pick_device(requested: str | None) -> str:
    """'cuda' (bound to LOCAL_RANK under torchrun) -- this build has no CPU path."""
    dev = requested or ("cuda" if torch.cuda.is_available() else "cpu")
    if not str(dev).startswith("cuda"):
        raise SystemExit("this MI355X build computes on a HIP device only (no CPU fallback); "
                         "run the reference package for --device cpu")
    if dev == "cuda":
        dev = f"cuda:{int(os.environ.get('LOCAL_RANK', '0')) % max(torch.cuda.device_count(), 1)}"
    torch.cuda.set_device(torch.device(dev))
    return dev


def load_codec_meta(store_dir: Path) -> Tuple[np.ndarray, np.ndarray]:
    meta = np.load(Path(store_dir) / "codec_meta.npz")
    return meta["scale"].astype("float32"), meta["zero"].astype("float32")


def load_embedding(bitstream: Path, scale: np.ndarray, zero: np.ndarray) -> np.ndarray:
    return decode_embedding(read_bitstream(Path(bitstream)), scale, zero)


def build_model(weights: str, device: str, z_dim: int, dtype: str = "fp32") -> CLIPCondUNet:
    """Architecture from the checkpoint's shapes (reference checkpoints give base=128, ch_mult=(1,2,2))."""
    sd = torch.load(weights, map_location="cpu", weights_only=True)
    arch = infer_arch(sd)
    if arch["z_dim"] != z_dim:
        raise SystemExit(f"checkpoint z_dim {arch['z_dim']} does not match the store's embedding dim {z_dim}")
    net = CLIPCondUNet(**arch, dtype=dtype).to(device)
    net.load_state_dict(sd, strict=True)
    net.eval()
    return net


def build_sampler(eta: float, device: str) -> DDIMSampler:
    return DDIMSampler(NoiseScheduler(timesteps=1000, schedule="cosine", device=device), eta=eta)


def start_noise(indices: Sequence[int], size: int, seed: int | None, img_ch: int = 3) -> torch.Tensor | None:
    """Per-record x_T from CPU generators seeded seed+index (sharding- and batch-invariant); None = unseeded."""
    if seed is None:
        return None
    rows = [torch.randn((img_ch, size, size), generator=torch.Generator("cpu").manual_seed(int(seed) + int(i)))
            for i in indices]
    return torch.stack(rows, 0)
