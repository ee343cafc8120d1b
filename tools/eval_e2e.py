#!/usr/bin/env python3
"""End-to-end `cli.eval` on a synthetic store (PNG + .clp per record, key-seeded checkpoint): wall time of the whole command body --
decode, reconstruction (two batches in flight), PSNR / SSIM on the host -- against the GPU-only rate of bench.py.

    python tools/eval_e2e.py [--n 64] [--size 256] [--dtype bf16]
"""
import argparse, io, sys, tempfile, time
from contextlib import redirect_stdout
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "clip-neural-image-conpression_amd")]
from clip_feature_codec.cli import eval as cli_eval  # noqa: E402
from clip_feature_codec.io import bitstream  # noqa: E402
from clip_feature_codec.utils import synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=256); ap.add_argument("--size", type=int, default=256); ap.add_argument("--dtype", default="bf16")
    a = ap.parse_args()
    with tempfile.TemporaryDirectory() as d:
        d = Path(d)
        synth.write_synth_store(d, a.n, a.size, write_clp=bitstream.write_bitstream)
        sd = synth.synth_state_dict(synth.unet_param_spec(512, 128, (1, 2, 2)))
        torch.save({k: torch.from_numpy(v) for k, v in sd.items()}, d / "ckpt.pt")
        argv = ["--store_dir", str(d), "--weights", str(d / "ckpt.pt"), "--size", str(a.size), "--steps", "50", "--batch", "8", "--seed", "1",
                "--device", "cuda", "--dtype", a.dtype, "--out_json", str(d / "m.json"), "--timing"]
        for it in range(2):                                   # first pass pays checkpoint repack + graph capture
            buf = io.StringIO()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            with redirect_stdout(buf):
                cli_eval.main(argv)
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
            print(f"pass {it}: {a.n} records in {dt:.2f} s = {a.n / dt:.1f} images/s end to end; " + buf.getvalue().strip().splitlines()[0])


if __name__ == "__main__":
    main()
