"""FiLM and ResBlock with the reference's constructor / forward signatures, computed by libccn_hip.so.

Mirrors ``models/blocks.py:14-44`` of the reference: same class names, same parameter names
(``to_scale``, ``to_shift``, ``norm1``, ``conv1``, ``norm2``, ``conv2``, ``film``), so state dicts
interchange.  The ``torch.nn`` leaf modules are used as *parameter containers only* (storage,
default initialisation, ``.to()``, ``state_dict``); their ``forward`` is never called.
``AttnBlock`` / ``DWConvBlock`` are not part of CLIPCondUNet and are not provided.

Inside ``CLIPCondUNet`` the blocks are not run one by one: the whole UNet is one plan of fused
kernels (GroupNorm statistics in the producing conv's epilogue, GroupNorm-apply + SiLU in the
consuming conv's prologue, FiLM / residual in the epilogue).  The ``forward`` methods here are the
operator-level entry points the reference's unit tests exercise (``tests/test_blocks.py``).
"""
from __future__ import annotations

import torch
from torch import nn

from .. import _native


class FiLM(nn.Module):
    """Feature-wise linear modulation: ``x * (1 + to_scale(h)) + to_shift(h)``."""

    def __init__(self, c: int, cond_dim: int) -> None:
        super().__init__()
        self.to_scale = nn.Linear(cond_dim, c)
        self.to_shift = nn.Linear(cond_dim, c)

    def forward(self, x: torch.Tensor, h: torch.Tensor) -> torch.Tensor:
        return _native.film_forward(x, h, self.to_scale.weight, self.to_scale.bias,
                                    self.to_shift.weight, self.to_shift.bias)


class ResBlock(nn.Module):
    """``x + conv2(SiLU(GN(FiLM(conv1(SiLU(GN(x))), h))))`` -- models/blocks.py:40-44."""

    def __init__(self, c: int, cond_dim: int, groups: int = 8) -> None:
        super().__init__()
        self.c, self.cond_dim, self.groups = c, cond_dim, groups
        self.norm1 = nn.GroupNorm(min(groups, c), c)
        self.conv1 = nn.Conv2d(c, c, 3, padding=1)
        self.norm2 = nn.GroupNorm(min(groups, c), c)
        self.conv2 = nn.Conv2d(c, c, 3, padding=1)
        self.film = FiLM(c, cond_dim)
        self.act = nn.SiLU()
        self._native = None          # (NativeUNet, parameter version) for stand-alone use
        self.compute_dtype = "fp32"

    def _standalone_handle(self, device) -> "_native.NativeUNet":
        """A one-stage handle whose block ``down.0`` carries this block's parameters."""
        version = tuple(p._version for p in self.parameters()) + (str(device), self.compute_dtype)
        if self._native is not None and self._native[1] == version:
            return self._native[0]
        nat = _native.NativeUNet(z_dim=8, base=self.c, ch_mult=(1,), time_dim=self.cond_dim, img_ch=3,
                                 groups=self.groups, dtype=self.compute_dtype, device=device)
        mine = {f"down.0.{k}": v for k, v in self.state_dict().items()}
        sd = {name: mine[name] if name in mine else torch.zeros(shape) for name, shape in nat.param_spec()}
        nat.load_state_dict(sd)
        self._native = (nat, version)
        return nat

    def forward(self, x: torch.Tensor, h: torch.Tensor) -> torch.Tensor:
        x = _native.require_dev(x, "x")
        return self._standalone_handle(x.device).resblock("down.0", x, h)
