#!/usr/bin/env python3
"""One UNet forward (and a short sample) at the bench shape, saved to .npy, to compare kernel variants across processes
(the variant switch CCN_CONV_DMA is read once per process)."""
import argparse, sys
from pathlib import Path
REPO = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(REPO), str(REPO / "clip-neural-image-conpression_amd")]
import numpy as np, torch
from clip_feature_codec.utils import synth
from clip_feature_codec.models.unet import CLIPCondUNet
from clip_feature_codec.diffusion.scheduler import NoiseScheduler
from clip_feature_codec.diffusion.ddim import DDIMSampler

ap = argparse.ArgumentParser()
ap.add_argument("--dtype", default="bf16"); ap.add_argument("--batch", type=int, default=2)
ap.add_argument("--size", type=int, default=256); ap.add_argument("--steps", type=int, default=4); ap.add_argument("--out", required=True)
a = ap.parse_args()
dev = "cuda:0"
sd = synth.synth_state_dict(synth.unet_param_spec(512, 128, (1, 2, 2)))
net = CLIPCondUNet(512, 128, (1, 2, 2), dtype=a.dtype).to(dev).eval()
net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
z = torch.from_numpy(synth.synth_z(a.batch)).to(dev)
xT = torch.from_numpy(synth.start_noise(range(a.batch), a.size, 100)).to(dev)
t = torch.full((a.batch,), 999, device=dev, dtype=torch.long)
eps = net(xT, z, t)
s = DDIMSampler(NoiseScheduler(1000, "cosine", dev), 0.0)
x = s.sample(net, z, (a.batch, 3, a.size, a.size), steps=a.steps, x_T=xT)
torch.cuda.synchronize()
np.save(a.out + "_eps.npy", eps.float().cpu().numpy()); np.save(a.out + "_x.npy", x.float().cpu().numpy())
print("saved", a.out, float(eps.abs().mean()), float(x.abs().mean()), bool(torch.isfinite(eps).all()))
