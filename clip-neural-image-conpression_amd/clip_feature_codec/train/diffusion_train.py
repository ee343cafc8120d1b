"""The epsilon-MSE training step of ``CLIPCondUNet`` on libccn_hip.so.

Reference: ``train/diffusion_train.py:119-124,137-140`` -- per batch::

    t = randint(0, T, (b,)); noise = randn_like(x0)
    x_t = sch.q_sample(x0, t, noise); eps_hat = net(x_t, z, t); loss = F.mse_loss(eps_hat, noise)
    loss.backward(); opt.step(); opt.zero_grad()

Here ``net(x_t, z, t)`` of a ``CLIPCondUNet`` in training mode returns an ``eps_hat`` that carries an autograd node
(``UNetFunction``) whose backward is the library's hand-written backward pass, so the three reference lines run
unchanged with any torch optimiser.  The parameters are re-homed as views into ONE flat fp32 buffer (what the C ABI
reads); ``FusedAdamW`` is the matching one-launch optimiser, and ``train_step`` is the loop body above with the
fused loss kernel.  The optional extras of the reference loop (L1 / TV / CLIP-alignment terms, :125-136) need models
that are not part of this path and are not provided.

With ``torch.distributed`` initialised, ``train_step(..., ddp=True)`` averages the flat gradient buffer over the ranks
with one all-reduce (RCCL over xGMI on a GPU node): the data-parallel step of BASELINE.json configs[4].
"""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import torch
from torch import nn

from .. import _native


class FlatParams:
    """Flat fp32 homes of a module's parameters and gradients, in the trainer's layout."""

    def __init__(self, net: nn.Module, trainer: "_native.NativeTrainer") -> None:
        named = dict(net.named_parameters())
        keys = [k for k, _, _ in trainer.layout]
        if set(keys) != set(named):
            raise RuntimeError(f"parameter keys differ from the library's: {sorted(set(keys) ^ set(named))[:6]}")
        dev = trainer.device
        self.flat = torch.zeros(trainer.total, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(trainer.total, dtype=torch.float32, device=dev)
        self.views: List[Tuple[nn.Parameter, int, int]] = []
        with torch.no_grad():
            for name, shape, off in trainer.layout:
                p = named[name]
                if tuple(p.shape) != tuple(shape):
                    raise RuntimeError(f"{name}: shape {tuple(p.shape)} != {tuple(shape)}")
                n = p.numel()
                view = self.flat[off:off + n].view(shape)
                view.copy_(p.detach().to(dev, torch.float32))
                p.data = view
                p.grad = self.grad[off:off + n].view(shape)
                self.views.append((p, off, n))

    def intact(self) -> bool:
        base = self.flat.data_ptr()
        return all(p.data_ptr() == base + 4 * off for p, off, _ in self.views)

    def params(self) -> List[nn.Parameter]:
        return [p for p, _, _ in self.views]

    def rebind_grads(self) -> None:
        """Make every ``p.grad`` a view of the flat gradient buffer again (after ``zero_grad(set_to_none=True)``)."""
        for p, off, n in self.views:
            if p.grad is None or p.grad.data_ptr() != self.grad.data_ptr() + 4 * off:
                p.grad = self.grad[off:off + n].view(p.shape)


class UNetFunction(torch.autograd.Function):
    """eps_hat = UNet(x_t, z, t) with the library's backward.  Inputs after ``t`` are the parameters (autograd leaves)."""

    @staticmethod
    def forward(ctx, state, x_t, z, t, *params):
        x = _native.require_dev(x_t, "x_t"); zz = _native.require_dev(z, "z_clip"); tt = _native.require_dev(t, "t", torch.int64)
        eps = state.trainer.forward(state.fp.flat, x, zz, tt)
        ctx.state = state
        ctx.save_for_backward(x, zz)
        ctx.version = state.fp.flat._version
        return eps

    @staticmethod
    def backward(ctx, d_eps):
        state = ctx.state
        x, zz = ctx.saved_tensors
        if state.fp.flat._version != ctx.version:
            raise RuntimeError("parameters were modified between the training forward and its backward")
        g = torch.zeros_like(state.fp.flat)
        state.trainer.backward(state.fp.flat, g, x, zz, _native.require_dev(d_eps, "d_eps"))
        grads = tuple(g[off:off + n].view(p.shape) for p, off, n in state.fp.views)
        return (None, None, None, None) + grads


class TrainState:
    """Trainer handle + flat parameter homes of one CLIPCondUNet (created on its first training forward)."""

    def __init__(self, net: nn.Module, dtype: str, device) -> None:
        a = net.arch
        self.trainer = _native.NativeTrainer(a["z_dim"], a["base"], a["ch_mult"], a["time_dim"], a["img_ch"], groups=8,
                                             dtype=dtype, device=device)
        self.fp = FlatParams(net, self.trainer)
        self.dtype = dtype

    def apply(self, x_t, z, t):
        return UNetFunction.apply(self, x_t, z, t, *self.fp.params())


class FusedAdamW:
    """``torch.optim.AdamW`` semantics over the flat buffers, one kernel launch per step (train/diffusion_train.py:105,138)."""

    def __init__(self, net, lr: float = 2e-4, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 1e-2) -> None:
        self.state: TrainState = net.train_state()
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        fp = self.state.fp
        self.exp_avg = torch.zeros_like(fp.flat)
        self.exp_avg_sq = torch.zeros_like(fp.flat)
        self.steps = 0

    def step(self) -> None:
        fp = self.state.fp
        self.steps += 1
        _native.adamw_step(fp.flat, fp.grad, self.exp_avg, self.exp_avg_sq, self.lr, self.betas[0], self.betas[1], self.eps,
                           self.weight_decay, self.steps)
        fp.flat._version  # noqa: B018  (in-place kernel: bump below)
        fp.flat.add_(0)    # bumps the version counter so that a stale forward is detected

    def zero_grad(self, set_to_none: bool = False) -> None:
        self.state.fp.grad.zero_()
        self.state.fp.rebind_grads()


def average_gradients(flat_grad: torch.Tensor) -> torch.Tensor:
    """Data-parallel gradient of the global batch: ONE all-reduce of the flat buffer (RCCL over xGMI on a GPU node, gloo in the
    CPU tests), then divide by the world size -- what DistributedDataParallel does bucket by bucket."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(flat_grad)
        flat_grad.div_(dist.get_world_size())
    return flat_grad


def train_step(net, sch, opt, x0: torch.Tensor, z: torch.Tensor, t: Optional[torch.Tensor] = None,
               noise: Optional[torch.Tensor] = None, ddp: bool = False) -> torch.Tensor:
    """One optimisation step; returns the (detached) loss.  ``t`` / ``noise`` default to the reference's draws."""
    b = x0.size(0)
    if t is None:
        t = torch.randint(0, sch.timesteps, (b,), device=x0.device, dtype=torch.long)
    if noise is None:
        noise = torch.randn_like(x0)
    state: TrainState = net.train_state()
    fp = state.fp
    fp.rebind_grads()
    x_t = sch.q_sample(x0, t, noise)
    xx = _native.require_dev(x_t, "x_t"); zz = _native.require_dev(z, "z"); tt = _native.require_dev(t, "t", torch.int64)
    eps = state.trainer.forward(fp.flat, xx, zz, tt)
    loss, d_eps = _native.mse_loss_grad(eps, noise)
    state.trainer.backward(fp.flat, fp.grad, xx, zz, d_eps)
    if ddp:
        average_gradients(fp.grad)
    opt.step()
    opt.zero_grad()
    return loss
