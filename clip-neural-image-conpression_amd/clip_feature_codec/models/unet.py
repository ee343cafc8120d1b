"""CLIPCondUNet with the reference's constructor, state-dict keys and ``forward(x_t, z_clip, t)``,
evaluated by the hand-written gfx950 kernels of libccn_hip.so.

Drop-in for ``models/unet.py:42-106`` of the reference:
  * ``CLIPCondUNet(z_dim, base, ch_mult, time_dim, img_ch)`` registers the same module tree
    (``time_proj.{0,2}``, ``z_proj.0``, ``in_conv``, ``down.{i}``, ``mid1``, ``mid2``, ``up.{i}``,
    ``out_norm``, ``out``), so ``load_state_dict(torch.load(ckpt), strict=True)`` works unchanged and
    default initialisation consumes the torch RNG in the same order as the reference.
  * ``forward`` takes NCHW fp32 ``x_t``, ``(B, z_dim)`` ``z_clip``, ``(B,)`` int64 ``t`` and returns eps
    with the shape / dtype / device of ``x_t``.
  * in ``.train()`` mode with gradients enabled, ``forward`` returns an eps that carries an autograd node whose backward is
    the library's hand-written backward pass (``train/diffusion_train.py`` of this package), so
    ``F.mse_loss(net(x_t, z, t), noise).backward()`` fills ``p.grad`` of every parameter.
  * ``sample_ddim`` is the fused fast path ``DDIMSampler.sample`` uses: the whole loop as one hipGraph.

The ``torch.nn`` leaves only hold parameters; no torch operator runs in ``forward``.  Tensors must
live on a HIP device -- there is no CPU fallback (use the reference itself for ``device='cpu'``).
"""
from __future__ import annotations

import warnings
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
from torch import nn

from .. import _native
from .blocks import ResBlock, FiLM  # noqa: F401  (FiLM re-exported like the reference module)


def timestep_embedding(t: torch.Tensor, dim: int, max_period: int = 10000) -> torch.Tensor:
    """Sinusoidal embedding ``[cos(t f_i) | sin(t f_i)]``, ``f_i = exp(-ln(max_period) i / (dim//2))``."""
    if max_period != 10000:
        raise ValueError("the HIP kernel implements max_period=10000 (the only value the reference uses)")
    return _native.timestep_embedding(t, dim)


def infer_arch(sd) -> Dict[str, object]:
    """(z_dim, base, ch_mult, time_dim, img_ch) from checkpoint tensor shapes.

    The reference hard-codes base=128, ch_mult=(1,2,2) at its three construction sites
    (cli/eval.py:50); reading the architecture off the checkpoint makes the other BASELINE
    configurations loadable through the same CLIs.
    """
    base, img_ch = sd["in_conv.weight"].shape[0], sd["in_conv.weight"].shape[1]
    mults: List[int] = []
    i = 0
    while f"down.{3 * i + 2}.weight" in sd:
        w = sd[f"down.{3 * i + 2}.weight"]
        mults.append(int(w.shape[0]) // int(w.shape[1]))
        i += 1
    return dict(z_dim=int(sd["z_proj.0.weight"].shape[1]), base=int(base), ch_mult=tuple(mults),
                time_dim=int(sd["time_proj.0.weight"].shape[1]), img_ch=int(img_ch))


class CLIPCondUNet(nn.Module):
    """FiLM-conditioned pixel-space U-Net, epsilon prediction."""

    def __init__(self, z_dim: int = 512, base: int = 128, ch_mult: Tuple[int, ...] = (1, 2, 2),
                 time_dim: int = 256, img_ch: int = 3, dtype: str = "fp32", weight_rounding: str = "phases") -> None:
        super().__init__()
        self.arch = dict(z_dim=z_dim, base=base, ch_mult=tuple(ch_mult), time_dim=time_dim, img_ch=img_ch)
        self.compute_dtype = dtype
        # bf16 mode only (ccn_set_weight_rounding): "phases" (error diffusion within every output channel AND along the DDIM steps),
        # "diffused" (within the output channel only), "nearest" (independent rounding)
        self.weight_rounding = weight_rounding
        # parameter containers, registered in the reference's order (same keys, same RNG consumption)
        self.time_proj = nn.Sequential(nn.Linear(time_dim, time_dim * 4), nn.SiLU(), nn.Linear(time_dim * 4, time_dim))
        self.z_proj = nn.Sequential(nn.Linear(z_dim, time_dim), nn.SiLU())
        self.in_conv = nn.Conv2d(img_ch, base, 3, padding=1)
        width = base
        self.down_chs: List[int] = [width]
        enc: List[nn.Module] = []
        for m in ch_mult:
            enc += [ResBlock(width, time_dim), ResBlock(width, time_dim),
                    nn.Conv2d(width, width * m, 3, stride=2, padding=1)]
            width *= m
            self.down_chs.append(width)
        self.down = nn.ModuleList(enc)
        self.mid1 = ResBlock(width, time_dim)
        self.mid2 = ResBlock(width, time_dim)
        dec: List[nn.Module] = []
        for m in reversed(ch_mult):
            dec += [ResBlock(width, time_dim), ResBlock(width, time_dim),
                    nn.ConvTranspose2d(width, width // m, 4, stride=2, padding=1)]
            width //= m
        self.up = nn.ModuleList(dec)
        self.out_norm = nn.GroupNorm(8, width)
        self.out = nn.Conv2d(width, img_ch, 3, padding=1)
        self._nat: Optional[_native.NativeUNet] = None
        self._nat_key = None
        self._warned_grad = False

    # ------------------------------------------------------------------ construction helpers
    @classmethod
    def from_state_dict(cls, sd, dtype: str = "fp32") -> "CLIPCondUNet":
        net = cls(**infer_arch(sd), dtype=dtype)
        net.load_state_dict(sd, strict=True)
        return net

    def set_compute_dtype(self, dtype: str) -> "CLIPCondUNet":
        """'fp32' (parity mode, fp32 MFMA) or 'bf16' (throughput mode, bf16 MFMA + bf16 activations)."""
        _native.dtype_code(dtype)
        self.compute_dtype = dtype
        return self

    # ------------------------------------------------------------------ native handle
    def native(self, device=None) -> _native.NativeUNet:
        """The ccn handle for the current parameters (re-committed when they change or move)."""
        params = list(self.parameters())
        dev = torch.device(device) if device is not None else params[0].device
        if dev.type != "cuda":
            raise RuntimeError(
                f"CLIPCondUNet is on {dev}: this build only computes on a HIP device (no CPU fallback); "
                "call .to('cuda') or run the reference package for device='cpu'")
        if dev.index is None:
            dev = torch.device("cuda", torch.cuda.current_device())
        # parameters re-homed into the trainer's flat buffer are written by the fused AdamW kernel through that buffer: its version
        # counter (bumped by FusedAdamW.step) is part of the key, the views' own counters do not move
        st = getattr(self, "_train_state", None)
        flat_version = st.fp.flat._version if st is not None and st.fp.intact() else None
        key = (str(dev), self.compute_dtype, self.weight_rounding, tuple(p._version for p in params), tuple(p.data_ptr() for p in params),
               flat_version)
        if self._nat is None or self._nat_key != key:
            if self._nat is not None:
                self._nat.close()
            a = self.arch
            nat = _native.NativeUNet(a["z_dim"], a["base"], a["ch_mult"], a["time_dim"], a["img_ch"], groups=8,
                                     dtype=self.compute_dtype, device=dev, weight_rounding=self.weight_rounding)
            nat.load_state_dict(self.state_dict())
            self._nat, self._nat_key = nat, key
        return self._nat

    def train_state(self, device=None):
        """Trainer handle and flat parameter buffers (parameters become views into one fp32 buffer on first use)."""
        from ..train.diffusion_train import TrainState
        params = list(self.parameters())
        dev = torch.device(device) if device is not None else params[0].device
        if dev.type != "cuda":
            raise RuntimeError(f"CLIPCondUNet is on {dev}: the training step only runs on a HIP device (no CPU fallback)")
        if dev.index is None:
            dev = torch.device("cuda", torch.cuda.current_device())
        st = getattr(self, "_train_state", None)
        if st is None or st.dtype != self.compute_dtype or st.trainer.device != dev or not st.fp.intact():
            if st is not None:
                st.trainer.close()
            st = TrainState(self, self.compute_dtype, dev)
            object.__setattr__(self, "_train_state", st)
        return st

    # ------------------------------------------------------------------ the reference interface
    def forward(self, x_t: torch.Tensor, z_clip: torch.Tensor, t: torch.Tensor) -> torch.Tensor:
        out_dtype = x_t.dtype
        if torch.is_grad_enabled() and self.training and any(p.requires_grad for p in self.parameters()):
            # training forward (train/diffusion_train.py:123): activations kept, autograd node with the library's backward
            eps = self.train_state(x_t.device).apply(x_t.float(), z_clip.float(), t)
            return eps if out_dtype == torch.float32 else eps.to(out_dtype)
        eps = self.native(x_t.device).forward(x_t, z_clip, t)
        return eps if out_dtype == torch.float32 else eps.to(out_dtype)

    # ------------------------------------------------------------------ fused sampling
    @torch.no_grad()
    def sample_ddim(self, z_clip: torch.Tensor, x_T: torch.Tensor, ts: Sequence[int], coef: np.ndarray,
                    use_graph: bool = True, slot: int = 0, sigma=None, noise: Optional[torch.Tensor] = None) -> torch.Tensor:
        """All DDIM steps (eta = 0) inside the library; see ``ccn_sample`` in include/ccn_hip.h.  ``slot`` selects an independent
        workspace / captured graph, so that consecutive batches can be in flight on different streams."""
        return self.native(x_T.device).sample(z_clip, x_T, ts, coef, use_graph=use_graph, slot=slot, sigma=sigma, noise=noise)

    def read_activation(self, name: str, shape) -> torch.Tensor:
        return self.native().read_activation(name, tuple(shape))
