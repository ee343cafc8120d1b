"""Oracle: CPU restatement of the epsilon-MSE training step (loss, parameter gradients, one AdamW update).

TEST INFRASTRUCTURE (see oracle/__init__.py): only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this.  Gradients come from torch autograd applied to the functional forward of oracle/ref_unet.py on the CPU in
fp32; the optimiser update is restated elementwise.

Reference lines followed (relative to /root/reference/src/clip_feature_codec/):
  q_sample                 diffusion/scheduler.py:46-49
  loop body                train/diffusion_train.py:119-124   (t, noise, x_t, eps_hat = net(x_t, z, t), F.mse_loss(eps_hat, noise))
  backward / step          train/diffusion_train.py:137-140   (loss.backward(); opt.step(); opt.zero_grad())
  optimiser                train/diffusion_train.py:105       (torch.optim.AdamW(net.parameters(), lr=2e-4), torch defaults:
                                                               betas (0.9, 0.999), eps 1e-8, weight_decay 1e-2)
Pinned by tests/golden/train_step.npz (made by running the reference's own modules, tests/golden/make_train_golden.py).
"""
from __future__ import annotations

from typing import Dict, Tuple

import torch
import torch.nn.functional as F

from . import ref_unet, ref_diffusion

SD = Dict[str, torch.Tensor]


def loss_and_grads(sd: SD, x_t: torch.Tensor, z: torch.Tensor, t: torch.Tensor, target: torch.Tensor) -> Tuple[torch.Tensor, SD, torch.Tensor]:
    """mse_loss(unet(x_t, z, t), target), d loss / d every entry of ``sd``, and eps_hat."""
    leaves = {k: v.detach().clone().requires_grad_(True) for k, v in sd.items()}
    eps = ref_unet.unet_forward(leaves, x_t, z, t)
    loss = F.mse_loss(eps, target)
    grads = torch.autograd.grad(loss, list(leaves.values()))
    return loss.detach(), {k: g for k, g in zip(leaves, grads)}, eps.detach()


def train_step_grads(sd: SD, tables, x0: torch.Tensor, z: torch.Tensor, t: torch.Tensor, noise: torch.Tensor):
    x_t = ref_diffusion.q_sample(tables, x0, t, noise)
    return loss_and_grads(sd, x_t, z, t, noise) + (x_t,)


def adamw_update(p: torch.Tensor, g: torch.Tensor, m: torch.Tensor, v: torch.Tensor, step: int, lr: float = 2e-4,
                 betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 1e-2):
    """One AdamW update, the formula of torch.optim.AdamW (decoupled decay first, bias-corrected moments); returns (p, m, v)."""
    b1, b2 = betas
    p = p * (1.0 - lr * weight_decay)
    m = b1 * m + (1.0 - b1) * g
    v = b2 * v + (1.0 - b2) * g * g
    bc1, bc2 = 1.0 - b1 ** step, 1.0 - b2 ** step
    denom = v.sqrt() / (bc2 ** 0.5) + eps
    p = p - (lr / bc1) * (m / denom)
    return p, m, v
