#!/usr/bin/env python3
"""A/B of the headline workload: the 50-step loop replayed as one hipGraph against plain stream launches (same library calls).
Measured equal on this stack (113.5-113.9 ms per batch of 8 either way): the loop is GPU-bound and launches run ahead."""
import sys, time, torch
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "clip-neural-image-conpression_amd")]
from clip_feature_codec.utils import synth
from clip_feature_codec.models.unet import CLIPCondUNet
from clip_feature_codec.diffusion.scheduler import NoiseScheduler
from clip_feature_codec.diffusion.ddim import DDIMSampler
dev = "cuda:0"
sd = synth.synth_state_dict(synth.unet_param_spec(512, 128, (1, 2, 2)))
net = CLIPCondUNet(512, 128, (1, 2, 2), dtype="bf16").to(dev).eval()
net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
z = torch.from_numpy(synth.synth_z(8)).to(dev); x_T = torch.from_numpy(synth.start_noise(list(range(8)), 256, seed_base=100)).to(dev)
s = DDIMSampler(NoiseScheduler(1000, "cosine", dev), eta=0.0)
for g in (True, False, True, False):
    s.use_graph = g
    s.sample(net, z, (8, 3, 256, 256), steps=50, x_T=x_T); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): s.sample(net, z, (8, 3, 256, 256), steps=50, x_T=x_T)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    print("graph" if g else "plain", round(dt * 1e3, 2), "ms", round(8 / dt, 2), "img/s")
