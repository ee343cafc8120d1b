#!/usr/bin/env python3
"""Does quantising the DDIM STATE to bf16 at the stem's input move the PSNR?  fp32 arithmetic, fp32 weights, stepwise sampler with a
wrapper that rounds x_t to bf16 (what the bf16 mode's stem does to its im2col operand) or to bf16 hi + lo (exact to 16 bits)."""
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
n32 = CLIPCondUNet(512, 128, (1, 2, 2), dtype="fp32").to(dev).eval(); n32.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
z = torch.from_numpy(synth.synth_z(B)).to(dev); xT = torch.from_numpy(synth.start_noise(range(B), S, 100)).to(dev)
sm = DDIMSampler(NoiseScheduler(1000, "cosine", dev), 0.0)
orig = [synth.synth_image(i, S).astype(np.float32).transpose(2, 0, 1) / 127.5 - 1.0 for i in range(B)]
def ps(x): return np.array([psnr(orig[k], x[k].clamp(-1, 1).cpu().numpy()) for k in range(B)])
with torch.no_grad():
    x_ref = sm.sample(n32, z, (B, 3, S, S), steps=T, x_T=xT); p0 = ps(x_ref)
    def rep(name, fn):
        x = sm.sample(fn, z, (B, 3, S, S), steps=T, x_T=xT); p = ps(x); rel = (p - p0) / p0
        print(f"{name:50s} PSNR {p.mean():.4f} rel delta mean {rel.mean():+.3e} max|.| {np.abs(rel).max():.3e}  mean-abs vs fused fp32 {float((x - x_ref).abs().mean()):.5f}", flush=True)
    rep("stepwise fp32 (control)", lambda x, zz, t: n32(x, zz, t))
    rep("stepwise fp32, x_t rounded to bf16 at the stem", lambda x, zz, t: n32(x.to(torch.bfloat16).float(), zz, t))
    def hilo(x):
        h = x.to(torch.bfloat16).float(); return h + (x - h).to(torch.bfloat16).float()
    rep("stepwise fp32, x_t as bf16 hi + lo", lambda x, zz, t: n32(hilo(x), zz, t))
