#!/usr/bin/env python3
"""Time the training step (BASELINE.json configs[4] per-GPU shape: 256 px, base 128, (1,2,2), batch 4) on one GPU.

    python tools/train_bench.py [--dtype bf16|fp32] [--batch 4] [--size 256] [--steps 5] [--warmup 2]
"""
import argparse
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "clip-neural-image-conpression_amd")]

from clip_feature_codec.models.unet import CLIPCondUNet  # noqa: E402
from clip_feature_codec.diffusion.scheduler import NoiseScheduler  # noqa: E402
from clip_feature_codec.train.diffusion_train import FusedAdamW, train_step  # noqa: E402
from clip_feature_codec.utils import synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="bf16"); ap.add_argument("--batch", type=int, default=4); ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--base", type=int, default=128); ap.add_argument("--ch-mult", default="1,2,2")
    ap.add_argument("--steps", type=int, default=5); ap.add_argument("--warmup", type=int, default=2)
    a = ap.parse_args()
    dev = "cuda:0"
    ch_mult = tuple(int(v) for v in a.ch_mult.split(","))
    sd = synth.synth_state_dict(synth.unet_param_spec(512, a.base, ch_mult))
    net = CLIPCondUNet(512, a.base, ch_mult, dtype=a.dtype).to(dev)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    net.train()
    sch = NoiseScheduler(1000, "cosine", device=dev)
    opt = FusedAdamW(net, lr=2e-4)
    g = torch.Generator("cpu").manual_seed(0)
    x0 = (torch.rand((a.batch, 3, a.size, a.size), generator=g) * 2 - 1).to(dev)
    z = torch.from_numpy(synth.synth_z(a.batch)).to(dev)
    losses = []
    for i in range(a.warmup + a.steps):
        if i == a.warmup:
            torch.cuda.synchronize(); t0 = time.perf_counter()
        losses.append(train_step(net, sch, opt, x0, z))
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    print(f"train step {a.dtype} B={a.batch} {a.size}px base={a.base} {ch_mult}: {dt * 1e3:.2f} ms/step, {a.batch / dt:.2f} img/s, "
          f"loss {float(losses[0]):.4f} -> {float(losses[-1]):.4f}, workspace {net.train_state().trainer.workspace(a.batch, a.size, a.size).nbytes / 2**30:.2f} GiB")


if __name__ == "__main__":
    main()
