#!/usr/bin/env python3
"""Small fixed workload for rocprofv3 passes: batch-8 256px bf16 sample with a few DDIM steps, launch by launch."""
import argparse, sys
from pathlib import Path
REPO = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(REPO), str(REPO / "clip-neural-image-conpression_amd")]
import torch
from clip_feature_codec.utils import synth
from clip_feature_codec.models.unet import CLIPCondUNet
from clip_feature_codec.diffusion.scheduler import NoiseScheduler
from clip_feature_codec.diffusion.ddim import DDIMSampler

ap = argparse.ArgumentParser()
ap.add_argument("--dtype", default="bf16"); ap.add_argument("--steps", type=int, default=2); ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--size", type=int, default=256); ap.add_argument("--reps", type=int, default=1)
ap.add_argument("--base", type=int, default=128); ap.add_argument("--ch-mult", default="1,2,2")    # C4: --size 512 --base 192 --ch-mult 1,2,2,4 --batch 4
a = ap.parse_args()
dev = "cuda:0"
cm = tuple(int(v) for v in a.ch_mult.split(","))
sd = synth.synth_state_dict(synth.unet_param_spec(512, a.base, cm))
net = CLIPCondUNet(512, a.base, cm, dtype=a.dtype).to(dev).eval()
net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
z = torch.from_numpy(synth.synth_z(a.batch)).to(dev)
xT = torch.from_numpy(synth.start_noise(range(a.batch), a.size, 100)).to(dev)
s = DDIMSampler(NoiseScheduler(1000, "cosine", dev), 0.0); s.use_graph = False
for _ in range(a.reps):
    x = s.sample(net, z, (a.batch, 3, a.size, a.size), steps=a.steps, x_T=xT)
torch.cuda.synchronize()
print("ok", float(x.abs().mean()))
import os, ctypes
if os.environ.get("CCN_STAMPS"):
    from clip_feature_codec import _native
    lib = _native.load_library()
    fn = {"2": lib.ccn_internal_dump_stamps_fr, "3": lib.ccn_internal_dump_stamps_fr,
          "4": lib.ccn_internal_dump_stamps_pr}.get(os.environ.get("CCN_CONV_DMA", "4"), lib.ccn_internal_dump_stamps)
    print("stamps dump rc", fn(b"gpurun_out/stamps.txt"))
