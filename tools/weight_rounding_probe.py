#!/usr/bin/env python3
"""Which conv weights carry the bf16 mode's PSNR shift, and which zero-cost rounding removes it?

The bf16 mode's +0.1 % PSNR shift against the fp32 mode is the static perturbation of the model by storing its conv weights in
bf16 (tools/bf16_bias_probe.py).  This probe runs the fp32 MODE (fp32 storage, fp32 MFMA) -- i.e. exact arithmetic -- on weights that
were pre-rounded on the host in different ways, on the exact bench workload (batch 8, 256 px, 50 DDIM steps):

  * leave-one-IN table: only ONE level's conv weights rounded to bf16 (which layers carry the shift);
  * bf16 RNE everywhere (the bf16 mode's storage), bf16 with sum-preserving error diffusion per output channel
    along (cin, ky, kx) (ccn_commit_params could do this at zero run-time cost), fp16 RNE (what f16 MFMA operands would store),
    bf16 hi + lo (two MFMAs per product).

Per variant: PSNR against the synthetic originals per record (eval/metrics.py:22-29 of the reference), relative delta against the
fp32 weights (north_star gate: 0.1 %), and mean-abs of the final tensor against the fp32 run.
Usage (GPU box):  python tools/weight_rounding_probe.py [--quick]
"""
import sys
from pathlib import Path
import numpy as np, torch
REPO = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(REPO), str(REPO / "clip-neural-image-conpression_amd")]
from clip_feature_codec.utils import synth
from clip_feature_codec.models.unet import CLIPCondUNet
from clip_feature_codec.diffusion.scheduler import NoiseScheduler
from clip_feature_codec.diffusion.ddim import DDIMSampler
from clip_feature_codec.eval.metrics import psnr

dev = "cuda:0"; B, S, T = 8, 256, 50
quick = "--quick" in sys.argv
sd = synth.synth_state_dict(synth.unet_param_spec(512, 128, (1, 2, 2)))
conv_keys = [k for k, v in sd.items() if v.ndim == 4]


def bf16_rne(w):
    return torch.from_numpy(w).to(torch.bfloat16).float().numpy()


def fp16_rne(w):
    return w.astype(np.float16).astype(np.float32)


def bf16_hilo(w):
    hi = bf16_rne(w)
    return hi + bf16_rne(w - hi)


def bf16_diffused(w, transposed):
    """Sequential error diffusion per output channel along (cin, ky, kx), kx fastest: every rounding error is carried into the next
    element, so partial sums over taps / input channels track the fp32 weights (|sum error| <= half an ulp of the last element)."""
    a = np.moveaxis(w, 1, 0) if transposed else w             # ConvTranspose2d stores (Cin, Cout, 4, 4)
    co = a.shape[0]
    flat = np.ascontiguousarray(a.reshape(co, -1)).astype(np.float64)
    out = np.empty_like(flat)
    err = np.zeros(co)
    for k in range(flat.shape[1]):
        t = flat[:, k] + err
        r = torch.from_numpy(t.astype(np.float32)).to(torch.bfloat16).float().numpy().astype(np.float64)
        out[:, k] = r
        err = t - r
    o = out.astype(np.float32).reshape(a.shape)
    return np.ascontiguousarray(np.moveaxis(o, 0, 1)) if transposed else o


def bf16_diffused_tapmajor(w, transposed):
    """The same diffusion walking (ky, kx, cin) with cin fastest: channel sums of one tap first, taps last."""
    a = np.moveaxis(w, 1, 0) if transposed else w
    a2 = np.ascontiguousarray(np.transpose(a, (0, 2, 3, 1)))          # (co, ky, kx, ci)
    o = bf16_diffused(a2.reshape(a2.shape[0], 1, 1, -1), False).reshape(a2.shape)
    o = np.ascontiguousarray(np.transpose(o, (0, 3, 1, 2)))
    return np.ascontiguousarray(np.moveaxis(o, 0, 1)) if transposed else o


def level_of(k):
    if k.startswith("in_conv"): return "stem"
    if k.startswith("out."): return "head"
    if k.startswith("mid"): return "mid (512ch 32^2)"
    blk, idx = k.split(".")[0], int(k.split(".")[1])
    lv = idx // 3
    kind = "s2/convT" if idx % 3 == 2 else "res"
    names = {"down": ["down 128ch 256^2", "down 128ch 128^2", "down 256ch 64^2"], "up": ["up 512ch 32^2", "up 256ch 64^2", "up 128ch 128^2"]}
    return f"{names[blk][lv]} {kind}"


z = torch.from_numpy(synth.synth_z(B)).to(dev); xT = torch.from_numpy(synth.start_noise(range(B), S, 100)).to(dev)
sm = DDIMSampler(NoiseScheduler(1000, "cosine", dev), 0.0)
orig = [synth.synth_image(i, S).astype(np.float32).transpose(2, 0, 1) / 127.5 - 1.0 for i in range(B)]


def run(weights, dtype="fp32", rounding="phases"):
    n = CLIPCondUNet(512, 128, (1, 2, 2), dtype=dtype, weight_rounding=rounding).to(dev).eval()
    n.load_state_dict({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in weights.items()})
    x = sm.sample(n, z, (B, 3, S, S), steps=T, x_T=xT)
    torch.cuda.synchronize()
    ps = np.array([psnr(orig[k], x[k].clamp(-1, 1).cpu().numpy()) for k in range(B)])
    return x, ps


x32, p32 = run(sd)
print(f"fp32 weights, fp32 arithmetic: PSNR mean {p32.mean():.4f}")


def report(name, weights, dtype="fp32", rounding="phases"):
    x, ps = run(weights, dtype, rounding)
    rel = (ps - p32) / p32
    print(f"{name:58s} PSNR {ps.mean():.4f}  rel delta mean {rel.mean():+.3e} max|.| {np.abs(rel).max():.3e}  {'PASS' if np.abs(rel).max() <= 1e-3 else 'fail'}"
          f"  mean-abs vs fp32 {float((x - x32).abs().mean()):.4f}  std {float(x.clamp(-1, 1).std()):.5f}", flush=True)
    return ps


def with_rounding(fn, keys):
    out = dict(sd)
    for k in keys:
        out[k] = fn(sd[k])
    return out


print("--- whole model, fp32 arithmetic on pre-rounded conv weights")
report("bf16 RNE (= the bf16 mode's weight storage)", with_rounding(bf16_rne, conv_keys))
report("fp16 RNE", with_rounding(fp16_rne, conv_keys))
report("bf16 hi + lo", with_rounding(bf16_hilo, conv_keys))
dif = {k: bf16_diffused(sd[k], k.startswith("up.") and sd[k].shape[2] == 4) for k in conv_keys}
report("bf16, error diffused per output channel (cin, ky, kx)", {**sd, **dif})
print("--- bf16 MODE (bf16 storage + bf16 MFMA) on the same weights")
report("bf16 mode, weight_rounding='nearest' (RNE in ccn_commit_params)", sd, "bf16", "nearest")
report("bf16 mode, weight_rounding='diffused' (within the output channel)", sd, "bf16", "diffused")
report("bf16 mode, weight_rounding='phases' (+ along the DDIM steps)", sd, "bf16", "phases")
if "--bf16-only" in sys.argv:
    sys.exit(0)
print("--- under error diffusion: which levels still carry the shift (fp32 arithmetic; this level kept fp32 / stored as fp16)")
levels = []
for k in conv_keys:
    if level_of(k) not in levels: levels.append(level_of(k))
for lv in levels:
    ks = [k for k in conv_keys if level_of(k) == lv]
    report(f"diffused bf16, but {lv} kept fp32", {**sd, **dif, **{k: sd[k] for k in ks}})
hot = [k for k in conv_keys if level_of(k) in ("stem", "down 128ch 256^2 res")]
report("diffused bf16, stem + 256^2 res convs as fp16", {**sd, **dif, **{k: fp16_rne(sd[k]) for k in hot}})
hot2 = hot + [k for k in conv_keys if level_of(k) in ("down 128ch 128^2 res", "head", "up 128ch 128^2 res")]
report("diffused bf16, stem + 256^2 + 128^2 res + head as fp16", {**sd, **dif, **{k: fp16_rne(sd[k]) for k in hot2}})
dif_t = {k: bf16_diffused_tapmajor(sd[k], k.startswith("up.") and sd[k].shape[2] == 4) for k in conv_keys}
report("bf16, error diffused tap-major (cin fastest)", {**sd, **dif_t})
if not quick:
    print("--- leave-one-IN: only this level's conv weights rounded to bf16 (RNE), everything else fp32")
    levels = []
    for k in conv_keys:
        if level_of(k) not in levels: levels.append(level_of(k))
    for lv in levels:
        ks = [k for k in conv_keys if level_of(k) == lv]
        report(f"only {lv} ({len(ks)} convs)", with_rounding(bf16_rne, ks))
    print("--- leave-one-OUT: everything bf16 RNE except this level (kept fp32)")
    for lv in levels:
        ks = [k for k in conv_keys if level_of(k) != lv]
        report(f"all but {lv}", with_rounding(bf16_rne, ks))
