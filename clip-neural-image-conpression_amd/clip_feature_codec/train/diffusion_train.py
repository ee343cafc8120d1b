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

    def grads_bound(self) -> Optional[str]:
        """'views' if every ``p.grad`` is its view of the flat gradient buffer, 'none' if every one is None, else None (foreign)."""
        base = self.grad.data_ptr()
        n_none = sum(1 for p, _, _ in self.views if p.grad is None)
        if n_none == len(self.views):
            return "none"
        if n_none == 0 and all(p.grad.data_ptr() == base + 4 * off for p, off, _ in self.views):
            return "views"
        return None

    def rebind_grads(self) -> None:
        """Make every ``p.grad`` a view of the flat gradient buffer again (after ``zero_grad(set_to_none=True)``)."""
        for p, off, n in self.views:
            if p.grad is None or p.grad.data_ptr() != self.grad.data_ptr() + 4 * off:
                p.grad = self.grad[off:off + n].view(p.shape)


class UNetFunction(torch.autograd.Function):
    """eps_hat = UNet(x_t, z, t) with the library's backward.  Inputs after ``t`` are the parameters (autograd leaves).

    The trainer keeps the activations of ONE forward (its workspace), so every forward takes a serial number and a backward whose
    serial is no longer the trainer's latest raises instead of differentiating the wrong activations (two forwards before one
    backward: micro-batches, teacher/student double calls).  When every ``p.grad`` is the parameter's view of the flat gradient
    buffer (or None), the library accumulates straight into that buffer and the node returns no parameter gradients -- no
    per-step 130 MB scratch, no second accumulation pass by autograd; otherwise (foreign ``p.grad`` tensors) it falls back to a
    scratch buffer that autograd accumulates."""

    @staticmethod
    def forward(ctx, state, x_t, z, t, *params):
        x = _native.require_dev(x_t, "x_t"); zz = _native.require_dev(z, "z_clip"); tt = _native.require_dev(t, "t", torch.int64)
        eps = state.trainer.forward(state.fp.flat, x, zz, tt)
        ctx.state = state
        ctx.save_for_backward(x, zz)
        ctx.version = state.fp.flat._version
        ctx.serial = state.trainer.serial
        return eps

    @staticmethod
    def backward(ctx, d_eps):
        state = ctx.state
        x, zz = ctx.saved_tensors
        if state.fp.flat._version != ctx.version:
            raise RuntimeError("parameters were modified between the training forward and its backward")
        if state.trainer.serial != ctx.serial:
            raise RuntimeError("another training forward of this model ran before this backward: the library keeps the activations "
                               "of one forward at a time (run forward -> backward pairs, e.g. one micro-batch after the other)")
        fp = state.fp
        d = _native.require_dev(d_eps, "d_eps")
        mode = fp.grads_bound()
        if mode is not None:
            if mode == "none":                     # after zero_grad(set_to_none=True): the buffer holds stale sums
                fp.grad.zero_()
            fp.rebind_grads()
            if state.ddp_bucketed and state.world() > 1:
                d = d * (1.0 / state.world())      # the sum over ranks is then already the mean
                # inside `with state.no_sync():` (all micro-batches but the last) the gradients only accumulate locally; the last
                # backward's bucket all-reduces then carry the whole accumulated sum -- torch DDP's no_sync semantics.  Without
                # it every backward would re-reduce the already averaged earlier micro-batches (x world).
                state.trainer.backward(fp.flat, fp.grad, x, zz, d, bucket_cb=state.enqueue_bucket if state.sync_grads else None)
            else:
                state.trainer.backward(fp.flat, fp.grad, x, zz, d)
            return (None,) * (4 + len(fp.views))
        # foreign .grad tensors (not views of the flat gradient buffer): the gradients go back through autograd.  Under data
        # parallelism they must still be averaged -- one blocking all-reduce of THIS backward's gradient here (also inside no_sync():
        # what autograd accumulates into foreign .grad tensors afterwards is out of the library's reach), never a silent per-rank one.
        g = torch.zeros_like(fp.flat)
        state.trainer.backward(fp.flat, g, x, zz, d)
        if state.ddp_bucketed and state.world() > 1:
            average_gradients(g)
        grads = tuple(g[off:off + n].view(p.shape) for p, off, n in fp.views)
        return (None, None, None, None) + grads


class TrainState:
    """Trainer handle + flat parameter homes of one CLIPCondUNet (created on its first training forward)."""

    def __init__(self, net: nn.Module, dtype: str, device) -> None:
        a = net.arch
        self.trainer = _native.NativeTrainer(a["z_dim"], a["base"], a["ch_mult"], a["time_dim"], a["img_ch"], groups=8,
                                             dtype=dtype, device=device)
        self.fp = FlatParams(net, self.trainer)
        self.dtype = dtype
        self.ddp_bucketed = False          # UNetFunction.backward all-reduces finished gradient ranges while it still runs
        self.sync_grads = True             # False inside no_sync(): gradient accumulation over micro-batches without communication
        self._works: list = []

    def no_sync(self):
        """Context manager for gradient accumulation under data parallelism (torch DDP's ``no_sync``): backward passes inside it
        only accumulate into the local gradient buffer; run the LAST micro-batch's backward outside it, then ``wait_grad_sync()``."""
        import contextlib

        @contextlib.contextmanager
        def ctx():
            prev, self.sync_grads = self.sync_grads, False
            try:
                yield
            finally:
                self.sync_grads = prev
        return ctx()

    @staticmethod
    def world() -> int:
        import torch.distributed as dist
        return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1

    def enqueue_bucket(self, lo: int, hi: int) -> None:
        import torch.distributed as dist
        self._works.append(dist.all_reduce(self.fp.grad[lo:hi], async_op=True))

    def wait_grad_sync(self) -> None:
        """Join the bucket all-reduces the last backward enqueued (call before the optimiser step)."""
        for w in self._works:
            w.wait()
        self._works.clear()

    def apply(self, x_t, z, t):
        return UNetFunction.apply(self, x_t, z, t, *self.fp.params())

    def static_buffers(self, x0: torch.Tensor, z: torch.Tensor) -> dict:
        """Fixed-address tensors of the fused step for this batch shape (what lets the library replay captured graphs)."""
        key = (tuple(x0.shape), tuple(z.shape))
        bufs = getattr(self, "_static", None)
        if bufs is None or bufs["key"] != key:
            dev = x0.device
            bufs = dict(key=key, x_t=torch.empty_like(x0), eps=torch.empty_like(x0), d_eps=torch.empty_like(x0), noise=torch.empty_like(x0),
                        x0=torch.empty_like(x0), z=torch.empty_like(z), t=torch.empty(x0.shape[0], dtype=torch.int64, device=dev),
                        loss=torch.empty((), dtype=torch.float32, device=dev), scratch=torch.empty(1024, dtype=torch.float32, device=dev))
            self._static = bufs
        return bufs


class FusedAdamW:
    """``torch.optim.AdamW`` semantics over the flat buffers, one kernel launch per step (train/diffusion_train.py:105,138)."""

    def __init__(self, net, lr: float = 2e-4, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 1e-2) -> None:
        self.state: TrainState = net.train_state()
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        fp = self.state.fp
        self.exp_avg = torch.zeros_like(fp.flat)
        self.exp_avg_sq = torch.zeros_like(fp.flat)
        self.steps = 0

    def step(self, zero_grad: bool = False) -> None:
        """``zero_grad=True``: ``step()`` + ``zero_grad()`` as one pass over the buffers (``ccn_adamw_step_zero_grad``)."""
        fp = self.state.fp
        self.steps += 1
        _native.adamw_step(fp.flat, fp.grad, self.exp_avg, self.exp_avg_sq, self.lr, self.betas[0], self.betas[1], self.eps,
                           self.weight_decay, self.steps, zero_grad=zero_grad)
        fp.flat[:1].add_(0)    # the kernel wrote behind torch's back: bump the (shared) version counter so that a stale forward is detected
        if zero_grad:
            fp.rebind_grads()

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
               noise: Optional[torch.Tensor] = None, ddp=False, graph: bool = False) -> torch.Tensor:
    """One optimisation step; returns the (detached) loss.  ``t`` / ``noise`` default to the reference's draws.

    All tensors the library touches live at fixed addresses (``TrainState.static_buffers``), so with ``graph=True`` the forward and
    the backward are replayed as captured hipGraphs after the first step of a shape -- measured SLOWER than plain stream launches
    for this step (7.98 vs 7.33 ms; 7.53 without the side-stream branch), hence off by default.  The returned loss is a view of a
    static buffer: read it before the next step."""
    state: TrainState = net.train_state()
    fp = state.fp
    fp.rebind_grads()
    sb = state.static_buffers(_native.require_dev(x0, "x0"), _native.require_dev(z, "z"))
    sb["x0"].copy_(x0); sb["z"].copy_(z)
    if t is None:
        sb["t"].random_(0, sch.timesteps)
    else:
        sb["t"].copy_(t)
    if noise is None:
        sb["noise"].normal_()
    else:
        sb["noise"].copy_(noise)
    state.trainer.set_graph(graph)
    sch.q_sample(sb["x0"], sb["t"], sb["noise"], out=sb["x_t"])
    state.trainer.forward(fp.flat, sb["x_t"], sb["z"], sb["t"], out=sb["eps"])
    loss, d_eps = _native.mse_loss_grad(sb["eps"], sb["noise"], bufs=(sb["loss"], sb["d_eps"], sb["scratch"]))
    import torch.distributed as dist
    have_pg = bool(ddp) and dist.is_available() and dist.is_initialized()
    world = dist.get_world_size() if have_pg else 1
    # ddp="always": the bucketed route whenever a process group exists, a one-rank group included (runs the RCCL all-reduces and the
    # stream joins of the N > 1 route on a single GPU: tests/test_gpu_rccl.py)
    if world > 1 or (have_pg and ddp == "always"):
        # data parallel: the flat gradient buffer is all-reduced bucket by bucket while the backward still runs (RCCL on its own
        # stream); d_eps is pre-scaled by 1 / world so that the sum is already the mean
        d_eps.mul_(1.0 / world)
        works = []
        state.trainer.backward(fp.flat, fp.grad, sb["x_t"], sb["z"], d_eps,
                               bucket_cb=lambda lo, hi: works.append(dist.all_reduce(fp.grad[lo:hi], async_op=True)))
        for w in works:
            w.wait()
    else:
        state.trainer.backward(fp.flat, fp.grad, sb["x_t"], sb["z"], d_eps)
    if isinstance(opt, FusedAdamW):
        opt.step(zero_grad=True)
    else:
        opt.step()
        opt.zero_grad()
    return loss


# ---- the reference's entry point (train/diffusion_train.py:36-60,69-150) -----------------------------------------------------
class StoreDataset(torch.utils.data.Dataset):
    """(image in [-1, 1] as (3, S, S) fp32, L2-normalised CLIP embedding) per manifest record -- train/diffusion_train.py:36-60."""

    def __init__(self, store_dir, out_size: int = 256) -> None:
        import json
        from pathlib import Path
        import numpy as np
        self.store_dir = Path(store_dir)
        self.manifest = json.loads((self.store_dir / "manifest.json").read_text(encoding="utf-8"))
        meta = np.load(self.store_dir / "codec_meta.npz")
        self.scale = meta["scale"].astype("float32")
        self.zero = meta["zero"].astype("float32")
        self.out_size = out_size

    def __len__(self) -> int:
        return len(self.manifest)

    def __getitem__(self, i: int):
        from pathlib import Path
        import numpy as np
        from PIL import Image
        from ..io.bitstream import read_bitstream, decode_embedding
        rec = self.manifest[i]
        z = decode_embedding(read_bitstream(Path(rec["bitstream"])), self.scale, self.zero).astype(np.float32).reshape(-1)
        img = Image.open(rec["image"]).convert("RGB").resize((self.out_size, self.out_size), Image.BICUBIC)
        arr = (np.array(img).astype(np.float32) / 127.5 - 1.0).transpose(2, 0, 1)
        return torch.from_numpy(arr), torch.from_numpy(z)


def total_variation(x: torch.Tensor) -> torch.Tensor:
    return (x[:, :, 1:, :] - x[:, :, :-1, :]).abs().mean() + (x[:, :, :, 1:] - x[:, :, :, :-1]).abs().mean()


def train_diffusion(store_dir, out_size: int = 256, epochs: int = 40, batch_size: int = 8, lr: float = 2e-4, timesteps: int = 1000,
                    schedule: str = "cosine", recon_w: float = 0.05, clip_w: float = 0.1, tv_w: float = 1e-4, device: str = "cuda",
                    save_dir=None, base: int = 128, ch_mult=(1, 2, 2), dtype: str = "bf16", num_workers: int = 2, log=print):
    """The reference's ``train_diffusion`` (same arguments, defaults, checkpoint names and log line) on the MI355X kernels.

    Per batch (train/diffusion_train.py:115-140): t ~ U{0..T-1}, noise ~ N, x_t = q_sample, eps_hat = net(x_t, z, t) through the
    library's forward, loss = mse(eps_hat, noise) [+ recon_w * L1(x0_pred, x0) + tv_w * TV(x0_pred): a few elementwise torch ops on
    (B, 3, S, S) whose gradient reaches eps_hat], ``loss.backward()`` runs the library's backward, AdamW step.  The CLIP-alignment
    term (clip_w, :129-136) needs ``open_clip`` with downloaded weights: when that import fails the term is skipped with a note
    (SURVEY.md section 8c).  Additions that default to the reference's behaviour: ``base`` / ``ch_mult`` / ``dtype``.  With
    ``torch.distributed`` initialised the records are sharded over the ranks and the flat gradient buffer is averaged with one
    all-reduce per step.
    """
    from pathlib import Path
    import torch.distributed as dist
    import torch.nn.functional as F
    from ..models.unet import CLIPCondUNet
    from ..diffusion.scheduler import NoiseScheduler
    save_dir = Path(save_dir or store_dir)
    save_dir.mkdir(parents=True, exist_ok=True)
    ds = StoreDataset(store_dir, out_size=out_size)
    ddp = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
    sampler = torch.utils.data.distributed.DistributedSampler(ds, shuffle=True) if ddp else None
    dl = torch.utils.data.DataLoader(ds, batch_size=batch_size, shuffle=sampler is None, sampler=sampler, num_workers=num_workers,
                                     pin_memory=True)
    z_dim = ds[0][1].numel()
    net = CLIPCondUNet(z_dim=z_dim, base=base, ch_mult=tuple(ch_mult), img_ch=3, dtype=dtype).to(device)
    if ddp:                                                    # same initial weights on every rank
        for p in net.parameters():
            dist.broadcast(p.data, src=0)
    sch = NoiseScheduler(timesteps=timesteps, schedule=schedule, device=device)
    net.train()
    state = net.train_state(device)
    # data parallel: finished ranges of the flat gradient buffer are all-reduced (RCCL) while the backward still runs
    state.ddp_bucketed = ddp
    opt = FusedAdamW(net, lr=lr)
    if clip_w > 0:
        try:
            import open_clip  # noqa: F401
            raise ImportError("the CLIP-alignment term is not wired to this build's kernels")
        except ImportError as exc:
            log(f"[train] clip_w={clip_w} ignored: {exc}")
    rank0 = not ddp or dist.get_rank() == 0
    final_path = save_dir / "diffusion_unet_final.pt"
    for ep in range(epochs):
        if sampler is not None:
            sampler.set_epoch(ep)
        running, seen = 0.0, 0
        for x0, z in dl:
            x0 = x0.to(device); z = z.to(device)
            b = x0.size(0)
            t = torch.randint(0, timesteps, (b,), device=device, dtype=torch.long)
            noise = torch.randn_like(x0)
            state.fp.rebind_grads()
            x_t = sch.q_sample(x0, t, noise)
            eps_hat = net(x_t, z, t)
            loss = F.mse_loss(eps_hat, noise)
            if recon_w > 0 or tv_w > 0:
                # predict_x0_from_eps (diffusion/scheduler.py:51-55) written with torch ops: its gradient must reach eps_hat
                sg = sch.sqrt_one_minus_alphas_cumprod[t].view(-1, 1, 1, 1); ac = sch.sqrt_alphas_cumprod[t].view(-1, 1, 1, 1)
                x0_pred = ((x_t - sg * eps_hat) / ac).clamp(-1, 1)
                if recon_w > 0:
                    loss = loss + recon_w * F.l1_loss(x0_pred, x0)
                if tv_w > 0:
                    loss = loss + tv_w * total_variation(x0_pred)
            loss.backward()
            state.wait_grad_sync()
            opt.step()
            opt.zero_grad()
            running += float(loss.detach()) * b
            seen += b
        if rank0:
            torch.save(net.state_dict(), save_dir / f"diffusion_unet_ep{ep + 1}.pt")
            log(f"[train] epoch {ep + 1}/{epochs} loss={running / max(seen, 1):.4f}")
    if rank0:
        torch.save(net.state_dict(), final_path)
    return final_path
