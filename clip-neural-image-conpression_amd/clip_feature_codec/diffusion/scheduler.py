"""NoiseScheduler: the reference's public attributes and methods, MI355X build.

Interface of ``diffusion/scheduler.py:18-55`` of the reference: constructor
``NoiseScheduler(timesteps=1000, schedule='cosine'|'linear', device)``, the eight table
attributes, ``q_sample`` and ``predict_x0_from_eps``; ``ValueError`` for an unknown schedule.

The 1000-entry tables are host logic: they are built once on the CPU in fp32 with the same
torch op sequence as the reference (so they are bit-identical to its ``device='cpu'`` tables --
device transcendentals / cumprod may differ in the last bit) and then copied to ``device``.
The per-element work (``q_sample``, ``predict_x0_from_eps``) runs in HIP kernels.
``p_mean_variance`` is not provided: nothing in the reference calls it.
"""
from __future__ import annotations

import math
from typing import Dict

import numpy as np
import torch

from .. import _native

_TABLES = ("betas", "alphas", "alphas_cumprod", "alphas_cumprod_prev", "sqrt_alphas_cumprod",
           "sqrt_one_minus_alphas_cumprod", "sqrt_recip_alphas", "posterior_variance")


def _host_tables(timesteps: int, schedule: str) -> Dict[str, torch.Tensor]:
    cpu = torch.device("cpu")
    if schedule == "cosine":
        offset = 0.008
        u = torch.linspace(0, timesteps, timesteps + 1, device=cpu) / timesteps
        f = torch.cos((u + offset) / (1 + offset) * math.pi / 2) ** 2
        f = f / f[0]
        betas = (1 - (f[1:] / f[:-1])).clamp(0.0001, 0.9999)
    elif schedule == "linear":
        betas = torch.linspace(1e-4, 0.02, timesteps, device=cpu)
    else:
        raise ValueError(f"Unknown schedule {schedule}")
    tab = {"betas": betas, "alphas": 1.0 - betas}
    tab["alphas_cumprod"] = torch.cumprod(tab["alphas"], dim=0)
    tab["alphas_cumprod_prev"] = torch.cat([torch.ones(1), tab["alphas_cumprod"][:-1]], dim=0)
    tab["sqrt_alphas_cumprod"] = torch.sqrt(tab["alphas_cumprod"])
    tab["sqrt_one_minus_alphas_cumprod"] = torch.sqrt(1.0 - tab["alphas_cumprod"])
    tab["sqrt_recip_alphas"] = torch.sqrt(1.0 / tab["alphas"])
    tab["posterior_variance"] = betas * (1.0 - tab["alphas_cumprod_prev"]) / (1.0 - tab["alphas_cumprod"])
    return tab


class NoiseScheduler:
    """DDPM schedule tables (linear or cosine betas) + forward-process helpers."""

    def __init__(self, timesteps: int = 1000, schedule: str = "cosine", device: str = "cuda") -> None:
        self.timesteps = timesteps
        self.schedule = schedule
        self.device = device
        self.host = _host_tables(timesteps, schedule)       # fp32 CPU tables: source of the DDIM coefficients
        for name in _TABLES:
            setattr(self, name, self.host[name].to(device))

    # -- DDIM host-side tables (diffusion/ddim.py:25,34-43) -------------------------------------
    def ddim_timesteps(self, steps: int) -> np.ndarray:
        """``linspace(T-1, 0, steps).long()`` -- fp32 linspace truncated, like the reference."""
        return torch.linspace(self.timesteps - 1, 0, steps).long().numpy().astype(np.int32)

    def ddim_coefficients(self, steps: int, eta: float = 0.0) -> np.ndarray:
        """(steps, 5) fp32: sqrt(1-ab_t), sqrt(ab_t), sqrt(ab_s), sqrt(ab_s - sigma^2), sigma.

        ab_s is ``alphas_cumprod_prev[t]`` (the reference's choice, ddim.py:35) and 1.0 on the
        last step; the direction coefficient has no "1 -" (ddim.py:42).  Both are reproduced.
        """
        ts = torch.linspace(self.timesteps - 1, 0, steps).long()
        acp, prev = self.host["alphas_cumprod"], self.host["alphas_cumprod_prev"]
        rows = np.zeros((steps, 5), np.float32)
        for i in range(steps):
            t = ts[i]
            ab_t = acp[t]
            ab_s = prev[t] if i < steps - 1 else torch.tensor(1.0)
            if float(ab_s) != 0.0:
                sigma = eta * torch.sqrt((1 - ab_s) / (1 - ab_t) * (1 - ab_t / ab_s))
            else:
                sigma = torch.tensor(0.0)
            rows[i] = (float(torch.sqrt(1 - ab_t)), float(torch.sqrt(ab_t)), float(torch.sqrt(ab_s)),
                       float(torch.sqrt(ab_s - sigma ** 2)), float(sigma))
        return rows

    # -- per-element helpers --------------------------------------------------------------------
    def _gather(self, table: torch.Tensor, t: torch.Tensor, like: torch.Tensor) -> torch.Tensor:
        return table.to(like.device)[t.to(like.device)].to(torch.float32).contiguous()

    def q_sample(self, x0: torch.Tensor, t: torch.Tensor, noise: torch.Tensor, out: torch.Tensor = None) -> torch.Tensor:
        """x_t = sqrt(ab_t) x0 + sqrt(1-ab_t) noise, per-sample t (scheduler.py:46-49); ``out``: optional result buffer."""
        return _native.q_sample(x0, noise, self._gather(self.sqrt_alphas_cumprod, t, x0),
                                self._gather(self.sqrt_one_minus_alphas_cumprod, t, x0), out)

    def predict_x0_from_eps(self, x_t: torch.Tensor, t: torch.Tensor, eps_hat: torch.Tensor) -> torch.Tensor:
        """(x_t - sqrt(1-ab_t) eps) / sqrt(ab_t) (scheduler.py:51-55)."""
        return _native.predict_x0(x_t, eps_hat, self._gather(self.sqrt_alphas_cumprod, t, x_t),
                                  self._gather(self.sqrt_one_minus_alphas_cumprod, t, x_t))
