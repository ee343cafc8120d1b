"""Host-side logic of the product package (no GPU, no HIP compute calls)."""
import json
import struct

import numpy as np
from conftest import bits_or_close
import pytest
import torch

from clip_feature_codec.diffusion.scheduler import NoiseScheduler
from clip_feature_codec.io import bitstream
from clip_feature_codec.eval import metrics
from clip_feature_codec.models.unet import CLIPCondUNet, infer_arch
from clip_feature_codec.cli import eval as cli_eval
from oracle import ref_codec, ref_diffusion


def test_scheduler_tables_match_reference_bits(golden):
    g = golden("scheduler.npz")
    for sched in ("cosine", "linear"):
        s = NoiseScheduler(1000, sched, device="cpu")
        for name in ref_diffusion.TABLE_NAMES:
            bits_or_close(getattr(s, name).numpy(), g[f"{sched}.{name}"], what=f"{sched}.{name}")
    with pytest.raises(ValueError, match="Unknown schedule"):
        NoiseScheduler(1000, "quadratic", device="cpu")


def test_ddim_tables_match_oracle(golden):
    g = golden("scheduler.npz")
    s = NoiseScheduler(1000, "cosine", device="cpu")
    for steps in (10, 50, 100):
        assert np.array_equal(s.ddim_timesteps(steps), g[f"ts.{steps}"])
        for eta in (0.0, 0.5):
            mine = s.ddim_coefficients(steps, eta)
            assert np.array_equal(mine, ref_diffusion.ddim_coefficients(ref_diffusion.scheduler_tables(), steps, eta),
                                  equal_nan=True)
        # reference quirk Q2 (ddim.py:42): with eta > 0 the direction coefficient sqrt(ab_s - sigma^2) is NaN on the
        # early cosine-schedule steps (ab_s ~ 2e-6 < sigma^2); reproduced, not fixed
        assert np.isnan(mine[0, 3]) and not np.isnan(s.ddim_coefficients(steps, 0.0)).any()


def test_param_spec_counts(synth):
    def count(spec):
        return sum(int(np.prod(s)) for _, s in spec)
    assert len(synth.unet_param_spec(512, 128, (1, 2, 2))) == 192
    assert count(synth.unet_param_spec(512, 128, (1, 2, 2))) == 32_530_435
    assert count(synth.unet_param_spec(512, 32, (1, 2))) == 1_374_147
    assert count(synth.unet_param_spec(512, 192, (1, 2, 2, 4))) == 815_721_475


def test_module_tree_has_reference_keys_and_loads_strict(synth, tiny_sd):
    net = CLIPCondUNet(z_dim=512, base=32, ch_mult=(1, 2))
    spec = synth.unet_param_spec(512, 32, (1, 2))
    assert [(k, tuple(v.shape)) for k, v in net.state_dict().items()] == [(k, tuple(s)) for k, s in spec]
    sd = {k: torch.from_numpy(v) for k, v in tiny_sd.items()}
    net.load_state_dict(sd, strict=True)
    assert infer_arch(sd) == dict(z_dim=512, base=32, ch_mult=(1, 2), time_dim=256, img_ch=3)
    bad = dict(sd); bad.pop("out.bias")
    with pytest.raises(RuntimeError):
        net.load_state_dict(bad, strict=True)


def test_default_init_consumes_rng_like_torch_modules():
    """Same module kinds in the same order => same default init as the reference for a given seed."""
    torch.manual_seed(7); a = CLIPCondUNet(base=32, ch_mult=(1, 2)).state_dict()
    torch.manual_seed(7); b = CLIPCondUNet(base=32, ch_mult=(1, 2)).state_dict()
    assert all(torch.equal(a[k], b[k]) for k in a)
    assert float(a["out_norm.weight"].min()) == 1.0 and float(a["out_norm.bias"].abs().max()) == 0.0


def test_product_refuses_cpu_tensors(tiny_sd):
    net = CLIPCondUNet(base=32, ch_mult=(1, 2)).eval()
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        net(torch.zeros(1, 3, 16, 16), torch.zeros(1, 512), torch.zeros(1, dtype=torch.long))
    s = NoiseScheduler(device="cpu")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        s.q_sample(torch.zeros(1, 3, 8, 8), torch.zeros(1, dtype=torch.long), torch.zeros(1, 3, 8, 8))


def test_bitstream_roundtrip_and_layout(tmp_path):
    rng = np.random.default_rng(0)
    q = rng.integers(0, 256, 512, dtype=np.uint8)
    p = tmp_path / "a.clp"
    bitstream.write_bitstream(q.tobytes(), 512, p)
    blob = p.read_bytes()
    assert blob[:4] == b"CLPF" and struct.unpack("<I", blob[4:8])[0] == len(blob) - 8
    assert blob[8:12] == b"\x28\xb5\x2f\xfd"                   # zstd frame magic
    assert np.array_equal(bitstream.read_bitstream(p), q)
    assert np.array_equal(ref_codec.read_bitstream(p), q)       # oracle reads the product's file
    ref_codec.write_bitstream(q.tobytes(), 512, tmp_path / "b.clp")
    assert np.array_equal(bitstream.read_bitstream(tmp_path / "b.clp"), q)
    (tmp_path / "bad.clp").write_bytes(b"XXXX" + blob[4:])
    with pytest.raises(AssertionError, match="Bad magic"):
        bitstream.read_bitstream(tmp_path / "bad.clp")
    bitstream.write_bitstream(b"", 0, tmp_path / "e.clp")       # empty payload
    assert bitstream.read_bitstream(tmp_path / "e.clp").size == 0


def test_decode_embedding_matches_oracle():
    rng = np.random.default_rng(1)
    q = rng.integers(0, 256, 512, dtype=np.uint8)
    scale = rng.uniform(1e-4, 1e-3, 512).astype(np.float32); zero = rng.uniform(-0.2, 0, 512).astype(np.float32)
    z = bitstream.decode_embedding(q, scale, zero)
    assert z.shape == (1, 512) and z.dtype == np.float32
    assert np.array_equal(z, ref_codec.decode_z(q, scale, zero))
    assert abs(float(np.linalg.norm(z)) - 1.0) < 1e-6


def test_psnr_known_answers(golden):
    g = golden("psnr_kat.npz")
    assert np.array_equal(metrics._to_uint8(g["a"]), g["u8_a"])
    assert float(metrics.psnr(g["a"], g["b"])) == float(g["psnr_ab"])
    assert metrics.psnr(g["a"], g["a"]) == float("inf")
    s = metrics.ssim(g["a"], g["b"])
    assert 0.0 < s < 1.0 and abs(metrics.ssim(g["a"], g["a"]) - 1.0) < 1e-12
    assert np.isnan(metrics.lpips_distance(g["a"], g["b"])) and np.isnan(metrics.clip_similarity(g["a"], g["b"]))


def test_synth_store_layout(tmp_path, synth):
    man = synth.write_synth_store(tmp_path / "store", 5, 32, write_clp=bitstream.write_bitstream)
    meta = np.load(tmp_path / "store" / "codec_meta.npz")
    assert set(meta.files) == {"scale", "zero", "dim"} and meta["scale"].shape == (512,)
    assert json.loads((tmp_path / "store" / "manifest.json").read_text()) == man and len(man) == 5
    z = bitstream.decode_embedding(bitstream.read_bitstream(man[2]["bitstream"]), meta["scale"], meta["zero"])
    assert np.abs(z[0] - synth.synth_z(5)[2]).max() < 2e-3      # 8-bit quantisation error
    assert cli_eval.load_original(man[0]["image"], 32).shape == (3, 32, 32)


def test_sharding_and_aggregate():
    assert cli_eval.shard_indices(64, 3, 8) == list(range(3, 64, 8))
    allidx = sorted(i for r in range(8) for i in cli_eval.shard_indices(61, r, 8))
    assert allidx == list(range(61))                              # ragged tail covered exactly once
    assert cli_eval.shard_indices(3, 5, 8) == []                  # more ranks than records
    rows = np.array([[30.0, 0.9, np.nan, np.nan], [20.0, np.nan, np.nan, np.nan]])
    agg = cli_eval.aggregate(rows)
    assert agg["psnr"] == 25.0 and agg["ssim"] == 0.9 and np.isnan(agg["lpips"])


def test_store_dataset_items_follow_the_reference_recipe(tmp_path, synth):
    """StoreDataset (train/diffusion_train.py:36-60 of the reference): image -> RGB, bicubic resize, /127.5 - 1, CHW fp32;
    embedding -> dequantised with codec_meta and L2-normalised.  Host logic only (PIL + numpy + libzstd)."""
    from PIL import Image
    from clip_feature_codec.train.diffusion_train import StoreDataset, total_variation
    store = tmp_path / "store"
    manifest = synth.write_synth_store(store, 3, 24, write_clp=bitstream.write_bitstream)
    ds = StoreDataset(store, out_size=16)
    assert len(ds) == 3
    img, z = ds[1]
    assert img.shape == (3, 16, 16) and img.dtype == torch.float32 and z.shape == (512,) and z.dtype == torch.float32
    ref_img = np.array(Image.open(manifest[1]["image"]).convert("RGB").resize((16, 16), Image.BICUBIC)).astype(np.float32) / 127.5 - 1.0
    assert np.array_equal(img.numpy(), ref_img.transpose(2, 0, 1))
    meta = np.load(store / "codec_meta.npz")
    q = bitstream.read_bitstream(manifest[1]["bitstream"])
    zz = q.astype(np.float32) * meta["scale"].astype("float32") + meta["zero"].astype("float32")
    zz = zz / max(np.linalg.norm(zz), 1e-9)
    assert np.allclose(z.numpy(), zz, atol=1e-7) and abs(float(z.norm()) - 1.0) < 1e-5
    x = torch.arange(2 * 3 * 4 * 5, dtype=torch.float32).reshape(2, 3, 4, 5)
    assert float(total_variation(x)) == pytest.approx(5.0 + 1.0)       # |d/dH| = 5, |d/dW| = 1 for a ramp


def test_training_forward_refuses_cpu_tensors(tiny_sd):
    """In .train() mode with gradients enabled the module routes to the library's training path, which has no CPU fallback."""
    net = CLIPCondUNet(base=32, ch_mult=(1, 2)).train()
    with pytest.raises(RuntimeError, match="HIP device"):
        net(torch.zeros(1, 3, 32, 32), torch.zeros(1, 512), torch.zeros(1, dtype=torch.long))


def test_uint8_metric_path_equals_the_float_path(tmp_path):
    """cli.eval moves uint8 off the GPU and scores uint8 images: the conversion is the reference's ``_to_uint8`` (two separately
    rounded fp32 ops, clip, truncation; eval/metrics.py:16-19) done with torch ops, and the originals go through the same float
    round trip the reference applies to them -- PSNR and SSIM must equal the float-array route bit for bit."""
    import torch
    from PIL import Image
    from clip_feature_codec.cli import eval as cli_eval
    from clip_feature_codec.eval import metrics
    rng = np.random.default_rng(5)
    img = rng.integers(0, 256, (48, 40, 3), dtype=np.uint8)
    path = tmp_path / "o.png"
    Image.fromarray(img).save(path)
    recon = torch.from_numpy(rng.uniform(-1.3, 1.3, (3, 48, 48)).astype(np.float32)).clamp(-1, 1)
    orig_f = cli_eval.load_original(str(path), 48)
    want = [metrics.psnr(orig_f, recon.numpy()), metrics.ssim(orig_f, recon.numpy())]
    recon_u8 = ((recon + 1.0) * 127.5).clamp(0, 255).to(torch.uint8).numpy()
    assert np.array_equal(recon_u8, metrics._to_uint8(recon.numpy()))
    got = cli_eval.metric_row_u8(cli_eval.original_u8(str(path), 48), recon_u8)
    assert got[0] == want[0] and got[1] == want[1] and np.isnan(got[2]) and np.isnan(got[3])
    assert 2 <= cli_eval.default_workers(1) <= 16 and cli_eval.default_workers(64) == 2


def test_bf16_weight_rounding_arithmetic_on_the_host():
    """ccn_set_weight_rounding's arithmetic (ccn_api.hip: diffuse_round_conv / _convT / _phases), run on the HOST through an internal
    export of the library (no GPU call).  Properties the PSNR gate relies on:
      * every rounded weight is bf16-representable and within one bf16 ulp (at the tensor's largest magnitude: a small weight
        absorbs the carried error of a large neighbour) of the fp32 weight;
      * within every output channel, EVERY prefix sum of the rounded weights along (cin, ky, kx) is within half an ulp of the fp32
        prefix sum (independent rounding drifts like a random walk: ~sqrt(K) half-ulps);
      * along the phases, the running sum W_0 + ... + W_k has the same property against (k + 1) W, so the MEAN weight over a period of
        four steps is four times closer to the fp32 weight than one rounding;
      * ConvTranspose2d: the diffusion runs per (output channel, output parity) over the four taps that parity sees."""
    import ctypes
    from clip_feature_codec import _native
    lib = ctypes.CDLL(str(_native.LIB_PATH))
    fn = lib.ccn_internal_round_weights
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    rng = np.random.default_rng(5)

    def bf16_ok(a):
        return np.all((a.view(np.uint32) & 0xFFFF) == 0)

    def ulp(a):                                    # bf16 ulp at the magnitude of a (8 significant bits)
        return 2.0 ** (np.floor(np.log2(np.maximum(np.abs(a), 1e-30))) - 7)

    O, I, taps, P = 24, 40, 9, 4
    w = (rng.standard_normal((O, I, taps)) / np.sqrt(3.0 * I * taps)).astype(np.float32)
    out = np.empty((P, O, I, taps), np.float32)
    assert fn(w.ctypes.data, O, I, taps, 0, P, out.ctypes.data) == 0
    assert bf16_ok(out)
    assert np.all(np.abs(out[0] - w) <= ulp(np.abs(w).max()) * 1.001)
    bound = 0.5 * ulp(np.abs(w).max()) * 1.001
    run = np.zeros_like(w, dtype=np.float64)
    for k in range(P):
        run += out[k]
        drift = np.cumsum((run - (k + 1) * w.astype(np.float64)).reshape(O, -1), axis=1)
        assert np.abs(drift).max() <= bound, (k, np.abs(drift).max(), bound)
    # independent rounding for comparison: its prefix sums wander several half-ulps away
    import torch
    rne = torch.from_numpy(w).to(torch.bfloat16).float().numpy().astype(np.float64)
    assert np.abs(np.cumsum((rne - w).reshape(O, -1), axis=1)).max() > 3 * bound
    # the period mean is closer to the fp32 weight than a single rounding (rms over all weights)
    e1 = np.sqrt(np.mean((out[0] - w) ** 2)); e4 = np.sqrt(np.mean((out.mean(0) - w) ** 2))
    assert e4 < 0.45 * e1, (e1, e4)
    # phases == 1 is the plain diffused rounding = version 0 of the phased one
    one = np.empty((1, O, I, taps), np.float32)
    assert fn(w.ctypes.data, O, I, taps, 0, 1, one.ctypes.data) == 0 and np.array_equal(one[0], out[0])

    # ConvTranspose2d (I, O, 4, 4): per (output channel, parity) over (cin, the parity's 2x2 taps)
    Ci, Co = 20, 12
    wt = (rng.standard_normal((Ci, Co, 4, 4)) / np.sqrt(3.0 * Ci * 4)).astype(np.float32)
    ot = np.empty((1, Ci, Co, 4, 4), np.float32)
    assert fn(wt.ctypes.data, Co, Ci, 16, 1, 1, ot.ctypes.data) == 0 and bf16_ok(ot)
    kk2 = [[1, 3], [0, 2]]                          # kernel indices feeding even / odd outputs (stride 2, padding 1)
    bt = 0.5 * ulp(np.abs(wt).max()) * 1.001
    for par in range(4):
        ky, kx = kk2[par >> 1], kk2[par & 1]
        sel = np.stack([(ot[0] - wt.astype(np.float64))[:, :, ky[t >> 1], kx[t & 1]] for t in range(4)], axis=-1)   # (Ci, Co, 4)
        drift = np.cumsum(np.transpose(sel, (1, 0, 2)).reshape(Co, -1), axis=1)
        assert np.abs(drift).max() <= bt, (par, np.abs(drift).max(), bt)
