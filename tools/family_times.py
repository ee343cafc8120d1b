#!/usr/bin/env python3
"""Per-family device time of one 50-step sample on the bench workload (HIP events around every launch, ccn_profile_*):
    [CCN_HIP_LIB=.../libccn_hip_diag.so CCN_STEM_UPW=8] python tools/family_times.py [--steps 10]"""
import argparse, sys
from pathlib import Path
REPO = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(REPO), str(REPO / "clip-neural-image-conpression_amd")]
import torch
from clip_feature_codec.utils import synth
from clip_feature_codec.models.unet import CLIPCondUNet
from clip_feature_codec.diffusion.scheduler import NoiseScheduler
from clip_feature_codec.diffusion.ddim import DDIMSampler
ap = argparse.ArgumentParser(); ap.add_argument("--steps", type=int, default=10); ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--size", type=int, default=256); ap.add_argument("--base", type=int, default=128); ap.add_argument("--ch-mult", default="1,2,2")
a = ap.parse_args()
dev = "cuda:0"; cm = tuple(int(v) for v in a.ch_mult.split(","))
sd = synth.synth_state_dict(synth.unet_param_spec(512, a.base, cm))
net = CLIPCondUNet(512, a.base, cm, dtype="bf16").to(dev).eval(); net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
z = torch.from_numpy(synth.synth_z(a.batch)).to(dev); xT = torch.from_numpy(synth.start_noise(range(a.batch), a.size, 100)).to(dev)
s = DDIMSampler(NoiseScheduler(1000, "cosine", dev), 0.0)
s.sample(net, z, (a.batch, 3, a.size, a.size), steps=a.steps, x_T=xT)
nat = net.native(); nat.profile(True)
s.sample(net, z, (a.batch, 3, a.size, a.size), steps=a.steps, x_T=xT)
fams = nat.profile_read(); nat.profile(False)
print("  ".join(f"{f['name']} {f['ms'] * 1e3 / f['calls']:.1f}us x{f['calls'] // a.steps}" for f in fams))
