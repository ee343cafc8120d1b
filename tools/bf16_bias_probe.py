#!/usr/bin/env python3
"""Why does bf16 mode score a slightly HIGHER PSNR against unrelated originals than fp32 mode (bench.py parity block)?
Measures (1) gain / noise of one bf16 forward against fp32, (2) contrast of the final images, (3) the PSNR shift of the fp32 mode
when zero-mean noise of the bf16 error's size is added to its eps every step (generic stepwise sampler route)."""
import sys
from pathlib import Path
import numpy as np, torch
REPO = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(REPO), str(REPO / "clip-neural-image-conpression_amd")]
from clip_feature_codec.utils import synth
from clip_feature_codec.models.unet import CLIPCondUNet
from clip_feature_codec.diffusion.scheduler import NoiseScheduler
from clip_feature_codec.diffusion.ddim import DDIMSampler
from clip_feature_codec.eval.metrics import psnr

dev = "cuda:0"; B, S, T = 8, 256, 50
sd = synth.synth_state_dict(synth.unet_param_spec(512, 128, (1, 2, 2)))
def mk(dt):
    n = CLIPCondUNet(512, 128, (1, 2, 2), dtype=dt).to(dev).eval(); n.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}); return n
n16, n32 = mk("bf16"), mk("fp32")
z = torch.from_numpy(synth.synth_z(B)).to(dev); xT = torch.from_numpy(synth.start_noise(range(B), S, 100)).to(dev)
sm = DDIMSampler(NoiseScheduler(1000, "cosine", dev), 0.0)
for tt in (999, 500, 100):
    t = torch.full((B,), tt, device=dev, dtype=torch.long)
    e16, e32 = n16(xT, z, t).double(), n32(xT, z, t).double()
    g = float((e16 * e32).sum() / (e32 * e32).sum())
    print(f"t={tt}: gain {g:.6f}  residual rms / eps rms {float((e16 - g * e32).pow(2).mean().sqrt() / e32.pow(2).mean().sqrt()):.4e}  mean diff {float((e16-e32).mean()):.2e}")
x16 = sm.sample(n16, z, (B, 3, S, S), steps=T, x_T=xT); x32 = sm.sample(n32, z, (B, 3, S, S), steps=T, x_T=xT)
c16, c32 = x16.clamp(-1, 1), x32.clamp(-1, 1)
print(f"std of final images: bf16 {float(c16.std()):.5f} fp32 {float(c32.std()):.5f}; mean |x|: {float(c16.abs().mean()):.5f} vs {float(c32.abs().mean()):.5f}; frac clamped {float((x16.abs()>=1).float().mean()):.4f} vs {float((x32.abs()>=1).float().mean()):.4f}")
orig = [synth.synth_image(i, S).astype(np.float32).transpose(2, 0, 1) / 127.5 - 1.0 for i in range(B)]
def mp(x): return float(np.mean([psnr(orig[k], x[k].clamp(-1, 1).cpu().numpy()) for k in range(B)]))
print(f"PSNR fp32 {mp(x32):.4f} bf16 {mp(x16):.4f}")
t = torch.full((B,), 999, device=dev, dtype=torch.long)
rel = float((n16(xT, z, t) - n32(xT, z, t)).pow(2).mean().sqrt())
for scale in (1.0, 2.0):
    gen = torch.Generator(device=dev).manual_seed(1)
    noisy = lambda x, zz, tt: n32(x, zz, tt) + scale * rel * torch.randn(x.shape, device=dev, generator=gen)
    xn = sm.sample(noisy, z, (B, 3, S, S), steps=T, x_T=xT)
    print(f"fp32 + zero-mean eps noise of {scale} x the bf16 error rms ({rel:.2e}): PSNR {mp(xn):.4f}; mean-abs vs fp32 {float((xn - x32).abs().mean()):.4f}")
# (4) fp32 arithmetic on bf16-ROUNDED weights: the static part of the bf16 error (what any bf16 implementation, the reference's
# autocast included, shares)
def rb(v):
    t = torch.from_numpy(v)
    return t.to(torch.bfloat16).float() if t.dim() == 4 else t
n32r = CLIPCondUNet(512, 128, (1, 2, 2), dtype="fp32").to(dev).eval(); n32r.load_state_dict({k: rb(v) for k, v in sd.items()})
x32r = sm.sample(n32r, z, (B, 3, S, S), steps=T, x_T=xT)
print(f"fp32 arithmetic, conv weights rounded to bf16: PSNR {mp(x32r):.4f}; std {float(x32r.clamp(-1,1).std()):.5f}; mean-abs vs fp32 {float((x32r - x32).abs().mean()):.4f}; bf16 mode vs this {float((x16 - x32r).abs().mean()):.4f}")
