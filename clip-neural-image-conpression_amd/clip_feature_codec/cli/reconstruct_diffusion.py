"""``python -m clip_feature_codec.cli.reconstruct_diffusion`` -- .clp -> DDIM -> PNG on an MI355X.

Same flags, defaults and output as the reference CLI (cli/reconstruct_diffusion.py:26-58):
``--store_dir --bitstream --weights --out --steps --eta --size --device``; prints ``Saved to <out>``;
the PNG is written from ``((clamp(x,-1,1)+1)*127.5).astype(uint8)`` (truncation).  Additions that
default to the reference behaviour: ``--seed`` (reproducible CPU-generated start noise; the reference
never seeds), ``--dtype {fp32,bf16}`` and the architecture is read from the checkpoint's shapes.
"""
from __future__ import annotations

import argparse
from pathlib import Path

import numpy as np
import torch
from PIL import Image

from ._common import pick_device, load_codec_meta, load_embedding, build_model, build_sampler, start_noise


def main(argv=None) -> None:
    ap = argparse.ArgumentParser(description="Reconstruct an image from a .clp bitstream via DDIM sampling (MI355X build).")
    ap.add_argument("--store_dir", type=str, required=True)
    ap.add_argument("--bitstream", type=str, required=True)
    ap.add_argument("--weights", type=str, required=True)
    ap.add_argument("--out", type=str, default="recon.png")
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--eta", type=float, default=0.0)
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--device", type=str, default=None)
    ap.add_argument("--seed", type=int, default=None, help="seed of the CPU-generated start noise (default: unseeded, like the reference)")
    ap.add_argument("--dtype", choices=["fp32", "bf16"], default="fp32")
    args = ap.parse_args(argv)

    device = pick_device(args.device)
    scale, zero = load_codec_meta(Path(args.store_dir))
    z = torch.from_numpy(load_embedding(Path(args.bitstream), scale, zero)).to(device)
    net = build_model(args.weights, device, z.shape[1], args.dtype)
    sampler = build_sampler(args.eta, device)
    x_T = start_noise([0], args.size, args.seed)
    with torch.no_grad():
        x = sampler.sample(net, z, shape=(1, 3, args.size, args.size), steps=args.steps,
                           x_T=None if x_T is None else x_T.to(device))
    img = x[0].clamp(-1, 1).cpu().numpy().transpose(1, 2, 0)
    Image.fromarray(((img + 1.0) * 127.5).astype(np.uint8)).save(args.out)
    print(f"Saved to {args.out}")


if __name__ == "__main__":
    main()
