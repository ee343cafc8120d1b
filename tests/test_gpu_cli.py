"""The two hot-path CLIs end to end on the GPU: synthetic store + key-seeded checkpoint -> images / metrics,
compared with the oracle reconstructing the same records (same seed, same .clp)."""
import json
from pathlib import Path

import numpy as np
import pytest
import torch
from PIL import Image

from clip_feature_codec.cli import eval as cli_eval, reconstruct_diffusion as cli_recon
from clip_feature_codec.cli._common import start_noise, load_codec_meta, load_embedding
from clip_feature_codec.io import bitstream
from clip_feature_codec.utils import synth
from oracle import ref_unet, ref_diffusion, ref_codec

pytestmark = pytest.mark.gpu
SIZE, N, STEPS, SEED = 32, 5, 6, 11


@pytest.fixture(scope="module")
def store(tmp_path_factory):
    d = tmp_path_factory.mktemp("store")
    synth.write_synth_store(d, N, SIZE, write_clp=bitstream.write_bitstream)
    sd = synth.synth_state_dict(synth.unet_param_spec(512, 32, (1, 2)))
    torch.save({k: torch.from_numpy(v) for k, v in sd.items()}, d / "ckpt.pt")
    return d, sd


def oracle_recon(store_dir, sd, indices):
    manifest = json.loads((store_dir / "manifest.json").read_text())
    scale, zero = load_codec_meta(store_dir)
    z = np.concatenate([ref_codec.decode_z(ref_codec.read_bitstream(manifest[i]["bitstream"]), scale, zero) for i in indices], 0)
    x = ref_diffusion.ddim_sample(ref_unet.make_model(ref_unet.as_torch_sd(sd)), torch.from_numpy(z),
                                  start_noise(indices, SIZE, SEED), steps=STEPS)
    return x.clamp(-1, 1).numpy(), manifest


def test_eval_cli_matches_oracle(store, capsys):
    d, sd = store
    out_json = d / "metrics.json"
    cli_eval.main(["--store_dir", str(d), "--weights", str(d / "ckpt.pt"), "--size", str(SIZE), "--steps", str(STEPS),
                   "--batch", "2", "--seed", str(SEED), "--device", "cuda", "--out_json", str(out_json)])
    printed = capsys.readouterr().out
    assert "Average PSNR:" in printed and "Average SSIM:" in printed and "Average LPIPS:" in printed and "Average CLIP similarity:" in printed
    recs = json.loads(out_json.read_text())
    assert [set(r) for r in recs] == [{"image", "psnr", "ssim", "lpips", "clip_sim"}] * N
    recon, manifest = oracle_recon(d, sd, list(range(N)))
    for i, r in enumerate(recs):
        orig = cli_eval.load_original(manifest[i]["image"], SIZE)
        want = ref_codec.psnr(orig, recon[i])
        assert r["image"] == manifest[i]["image"]
        assert abs(r["psnr"] - want) <= 1e-3 * abs(want), (i, r["psnr"], want)     # PSNR within 0.1 %


def test_reconstruct_cli_png(store, capsys):
    d, sd = store
    manifest = json.loads((d / "manifest.json").read_text())
    out = d / "recon.png"
    cli_recon.main(["--store_dir", str(d), "--bitstream", manifest[3]["bitstream"], "--weights", str(d / "ckpt.pt"),
                    "--out", str(out), "--steps", str(STEPS), "--size", str(SIZE), "--seed", str(SEED), "--device", "cuda"])
    assert f"Saved to {out}" in capsys.readouterr().out
    got = np.array(Image.open(out))
    scale, zero = load_codec_meta(d)
    z = load_embedding(Path(manifest[3]["bitstream"]), scale, zero)
    x = ref_diffusion.ddim_sample(ref_unet.make_model(ref_unet.as_torch_sd(sd)), torch.from_numpy(z),
                                  start_noise([0], SIZE, SEED), steps=STEPS)
    want = ((x[0].clamp(-1, 1).numpy().transpose(1, 2, 0) + 1.0) * 127.5).astype(np.uint8)
    assert got.shape == (SIZE, SIZE, 3)
    assert np.abs(got.astype(int) - want.astype(int)).max() <= 1          # uint8 truncation of values within 1e-3


def test_cli_refuses_cpu(store):
    d, _ = store
    with pytest.raises(SystemExit):
        cli_eval.main(["--store_dir", str(d), "--weights", str(d / "ckpt.pt"), "--device", "cpu"])
