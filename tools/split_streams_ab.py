#!/usr/bin/env python3
"""Experiment: the headline batch of 8 as ONE graph replay vs two concurrent replays of 4 samples each on two streams
(the samples are independent; the question is whether one half's launch gaps / kernel tails hide behind the other half)."""
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
def mk():
    net = CLIPCondUNet(512, 128, (1, 2, 2), dtype="bf16").to(dev).eval()
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    return net
nets = [mk(), mk()]
z = torch.from_numpy(synth.synth_z(8)).to(dev); x_T = torch.from_numpy(synth.start_noise(list(range(8)), 256, seed_base=100)).to(dev)
s = DDIMSampler(NoiseScheduler(1000, "cosine", dev), eta=0.0)
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
def whole():
    return s.sample(nets[0], z, (8, 3, 256, 256), steps=50, x_T=x_T)
def split(nparts):
    outs = []
    n = 8 // nparts
    for i in range(nparts):
        with torch.cuda.stream(streams[i % 2]):
            outs.append(s.sample(nets[i % 2], z[i * n:(i + 1) * n], (n, 3, 256, 256), steps=50, x_T=x_T[i * n:(i + 1) * n]))
    return outs
for name, fn in (("one batch of 8", whole), ("2 x 4 on two streams", lambda: split(2)), ("one batch of 8", whole), ("2 x 4 on two streams", lambda: split(2))):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): fn()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    print(f"{name}: {dt * 1e3:.2f} ms, {8 / dt:.2f} img/s")
# two whole batches of 8 in flight on two streams (consecutive batches of an eval loop are independent)
def two_batches():
    outs = []
    for i in range(2):
        with torch.cuda.stream(streams[i]):
            outs.append(s.sample(nets[i], z, (8, 3, 256, 256), steps=50, x_T=x_T))
    return outs
for _ in range(2):
    two_batches(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3): two_batches()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
    print(f"two batches of 8 in flight: {dt * 1e3:.2f} ms per pair, {16 / dt:.2f} img/s")
a = whole(); b = torch.cat(split(2)); torch.cuda.synchronize()
print("max abs difference between the two ways:", float((a - b).abs().max()))
