"""Oracle: functional CPU restatement of the CLIPCondUNet epsilon-prediction forward.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Operates on a plain ``dict`` of
tensors keyed like the reference state dict, so reference checkpoints, the
key-seeded synthetic weights and the HIP path all share one weight source.

Reference lines followed (relative to /root/reference/src/clip_feature_codec/):
  timestep_embedding  models/unet.py:22-39   (cos first, then sin; pad if dim is odd)
  conditioning h      models/unet.py:83-86   (time_proj MLP + z_proj, summed)
  FiLM                models/blocks.py:22-25 (x*(1+s)+b)
  ResBlock            models/blocks.py:40-44 (GN->SiLU->conv, FiLM, GN->SiLU->conv, +x)
  UNet body           models/unet.py:88-106  (additive skips, no SiLU before `out`)
"""
from __future__ import annotations

import math
from typing import Callable, Dict, Optional, Sequence

import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]


def as_torch_sd(sd, dtype=torch.float32) -> SD:
    """numpy or torch state dict -> dict of CPU tensors of ``dtype``."""
    out = {}
    for k, v in sd.items():
        t = v if isinstance(v, torch.Tensor) else torch.from_numpy(v)
        out[k] = t.detach().to("cpu", dtype)
    return out


def infer_arch(sd) -> dict:
    """Recover (z_dim, base, ch_mult, time_dim, img_ch) from tensor shapes."""
    base, img_ch = sd["in_conv.weight"].shape[0], sd["in_conv.weight"].shape[1]
    time_dim = sd["time_proj.0.weight"].shape[1]
    z_dim = sd["z_proj.0.weight"].shape[1]
    mults, i = [], 0
    while f"down.{3 * i + 2}.weight" in sd:
        w = sd[f"down.{3 * i + 2}.weight"]
        mults.append(w.shape[0] // w.shape[1])
        i += 1
    return dict(z_dim=int(z_dim), base=int(base), ch_mult=tuple(int(m) for m in mults),
                time_dim=int(time_dim), img_ch=int(img_ch))


def timestep_embedding(t: torch.Tensor, dim: int, max_period: int = 10000) -> torch.Tensor:
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(0, half) / half)
    args = t.float().unsqueeze(1) * freqs.unsqueeze(0)
    emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    if dim % 2 == 1:
        emb = F.pad(emb, (0, 1))
    return emb


def cond_vector(sd: SD, z_clip: torch.Tensor, t: torch.Tensor) -> torch.Tensor:
    time_dim = sd["time_proj.0.weight"].shape[1]
    temb = timestep_embedding(t, time_dim).to(z_clip.dtype)
    temb = F.linear(F.silu(F.linear(temb, sd["time_proj.0.weight"], sd["time_proj.0.bias"])),
                    sd["time_proj.2.weight"], sd["time_proj.2.bias"])
    zemb = F.silu(F.linear(z_clip, sd["z_proj.0.weight"], sd["z_proj.0.bias"]))
    return temb + zemb


def group_norm(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor, groups: int = 8) -> torch.Tensor:
    return F.group_norm(x, min(groups, x.shape[1]), w, b, eps=1e-5)


def film(sd: SD, p: str, y: torch.Tensor, h: torch.Tensor) -> torch.Tensor:
    s = F.linear(h, sd[f"{p}.to_scale.weight"], sd[f"{p}.to_scale.bias"])[:, :, None, None]
    b = F.linear(h, sd[f"{p}.to_shift.weight"], sd[f"{p}.to_shift.bias"])[:, :, None, None]
    return y * (1 + s) + b


def resblock(sd: SD, p: str, x: torch.Tensor, h: torch.Tensor, tap: Optional[Callable] = None) -> torch.Tensor:
    y = F.conv2d(F.silu(group_norm(x, sd[f"{p}.norm1.weight"], sd[f"{p}.norm1.bias"])),
                 sd[f"{p}.conv1.weight"], sd[f"{p}.conv1.bias"], padding=1)
    y = film(sd, f"{p}.film", y, h)
    if tap:
        tap(f"{p}.film_out", y)
    y = F.conv2d(F.silu(group_norm(y, sd[f"{p}.norm2.weight"], sd[f"{p}.norm2.bias"])),
                 sd[f"{p}.conv2.weight"], sd[f"{p}.conv2.bias"], padding=1)
    out = x + y
    if tap:
        tap(f"{p}.out", out)
    return out


def unet_forward(sd: SD, x_t: torch.Tensor, z_clip: torch.Tensor, t: torch.Tensor,
                 tap: Optional[Callable] = None) -> torch.Tensor:
    """eps_hat = UNet(x_t, z, t).  ``tap(name, tensor)`` receives intermediates."""
    n_stage = 0
    while f"down.{3 * n_stage + 2}.weight" in sd:
        n_stage += 1
    h = cond_vector(sd, z_clip, t)
    if tap:
        tap("h", h)
    x = F.conv2d(x_t, sd["in_conv.weight"], sd["in_conv.bias"], padding=1)
    if tap:
        tap("in_conv", x)
    skips = []
    for i in range(n_stage):
        x = resblock(sd, f"down.{3 * i}", x, h, tap)
        x = resblock(sd, f"down.{3 * i + 1}", x, h, tap)
        skips.append(x)
        x = F.conv2d(x, sd[f"down.{3 * i + 2}.weight"], sd[f"down.{3 * i + 2}.bias"], stride=2, padding=1)
        if tap:
            tap(f"down.{3 * i + 2}", x)
    x = resblock(sd, "mid1", x, h, tap)
    x = resblock(sd, "mid2", x, h, tap)
    for i in range(n_stage):
        x = resblock(sd, f"up.{3 * i}", x, h, tap)
        x = resblock(sd, f"up.{3 * i + 1}", x, h, tap)
        x = F.conv_transpose2d(x, sd[f"up.{3 * i + 2}.weight"], sd[f"up.{3 * i + 2}.bias"], stride=2, padding=1)
        if skips:
            x = x + skips.pop()
        if tap:
            tap(f"up.{3 * i + 2}", x)
    x = F.conv2d(group_norm(x, sd["out_norm.weight"], sd["out_norm.bias"]),
                 sd["out.weight"], sd["out.bias"], padding=1)
    return x


def make_model(sd: SD) -> Callable:
    """Callable ``model(x, z, t)`` with the reference's positional signature (ddim.py:33)."""
    def model(x, z, t):
        return unet_forward(sd, x, z, t)
    return model
