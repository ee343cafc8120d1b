"""DDIMSampler: the reference's ``DDIMSampler(scheduler, eta).sample(...)`` on the MI355X path.

Interface of ``diffusion/ddim.py:14-45`` of the reference.  Two execution routes, same numbers:

* fused  -- ``model`` is this package's ``CLIPCondUNet``: the timestep table and the
  per-step coefficients are computed on the host (no ``.item()`` syncs inside the loop), conditioning
  for all steps is hoisted in front of the loop, and the ``steps`` UNet evaluations + DDIM updates
  replay as ONE captured hipGraph (``ccn_sample``; ``eta > 0``: ``ccn_sample_eta``, the N(0,1) draws of all steps made up
  front, one ``normal_()`` per noisy step in loop order, so the torch generator is consumed exactly as by the stepwise loop).
* stepwise -- any other callable ``model(x, z, t)``: one ``model`` call and one ``ccn_ddim_step`` kernel per step; the
  ``eta > 0`` noise comes from ``torch.randn_like`` on the device.

Reference behaviours kept on purpose: ``ts = linspace(T-1, 0, steps).long()``; ``alpha_bar_prev`` is
``alphas_cumprod_prev[t]`` (not the next sampled step) and 1.0 on the last step; the direction term is
``sqrt(alpha_bar_s - sigma^2) * eps``; ``x0`` is clamped to [-1, 1]; ``cfg_scale`` is accepted and unused;
the result is NOT clamped; the output lives on ``z_clip.device``.
"""
from __future__ import annotations

from typing import Optional

import torch

from .. import _native


class DDIMSampler:
    """Deterministic (eta = 0) / stochastic DDIM sampler."""

    MAX_NOISE_BUFFERS = 4

    def __init__(self, scheduler, eta: float = 0.0) -> None:
        self.sch = scheduler
        self.eta = eta
        self.use_graph = True
        self._noise: dict = {}

    @torch.no_grad()
    def sample(self, model, z_clip: torch.Tensor, shape: tuple, steps: int = 50, cfg_scale: float = 1.0,
               x_T: Optional[torch.Tensor] = None, slot: int = 0) -> torch.Tensor:
        device = z_clip.device
        if device.type != "cuda":
            raise RuntimeError(f"z_clip is on {device}: DDIMSampler (MI355X build) needs a HIP device; no CPU fallback")
        ts = self.sch.ddim_timesteps(steps)
        coef = self.sch.ddim_coefficients(steps, self.eta)
        x = torch.randn(shape, device=device) if x_T is None else x_T.to(device)
        if hasattr(model, "sample_ddim"):
            if self.eta == 0:
                return model.sample_ddim(z_clip, x, ts, coef[:, :4], use_graph=self.use_graph, slot=slot)
            # eta > 0: every noisy step's draw up front, into a buffer kept per (shape, steps, slot) so that the captured graph
            # (keyed by the buffer's address) is replayed by later calls
            # Memory: steps x batch x C x H x W fp32 per key (314 MB at C2: 50 steps, batch 8, 256 px; 1.26 GB at C4) -- at most
            # MAX_NOISE_BUFFERS keys are kept, least recently used first (the library bounds its captured graphs per plan likewise)
            key = (tuple(shape), steps, slot, str(device))
            buf = self._noise.pop(key, None)
            if buf is None:
                while len(self._noise) >= self.MAX_NOISE_BUFFERS:
                    torch.cuda.synchronize(device)                    # a replay on a side stream may still read the evicted draws
                    self._noise.pop(next(iter(self._noise)))
                buf = torch.empty((steps,) + tuple(shape), dtype=torch.float32, device=device)
            self._noise[key] = buf                                    # (re)insert as most recently used
            for i in range(steps):
                if float(coef[i, 4]) > 0:
                    buf[i].normal_()                                  # == torch.randn_like(x): same generator, same order
            return model.sample_ddim(z_clip, x, ts, coef[:, :4], use_graph=self.use_graph, slot=slot, sigma=coef[:, 4], noise=buf)
        x = _native.require_dev(x, "x_T").clone()
        for i in range(steps):
            t_b = torch.full((shape[0],), int(ts[i]), device=device, dtype=torch.long)
            eps = model(x, z_clip, t_b)
            sigma = float(coef[i, 4])
            noise = torch.randn_like(x) if (self.eta > 0 and sigma > 0) else None
            _native.ddim_step(x, eps, coef[i, :4], sigma, noise)
        return x
