#!/usr/bin/env python3
"""Generate the golden fixtures in this directory by RUNNING THE REFERENCE on the CPU.

Run in the build container only (the reference does not exist on the GPU box):

    cd /root/repo && PYTHONDONTWRITEBYTECODE=1 PYTHONPATH=/root/reference/src \
        python tests/golden/make_golden.py [--skip-c2]

What is imported from the reference (unmodified, read-only mount):
    clip_feature_codec.models.unet      (CLIPCondUNet, timestep_embedding)
    clip_feature_codec.models.blocks    (ResBlock, FiLM)
    clip_feature_codec.diffusion.scheduler / .ddim
    clip_feature_codec.eval.metrics     (psnr, _to_uint8)
io.bitstream cannot be imported (``zstandard`` is absent), so the .clp
fixture is pinned by its byte layout only.

Weights are NOT stored: they are regenerated from the key-seeded generator
``clip_feature_codec/utils/synth.py`` of this repo (loaded by file path, because both
packages are named ``clip_feature_codec``).  Every fixture is data: inputs and the
reference's outputs for them.
"""
from __future__ import annotations

import argparse
import importlib.util
import sys
import time
from pathlib import Path

import numpy as np
import torch

HERE = Path(__file__).resolve().parent
REPO = HERE.parent.parent
sys.path.insert(0, str(REPO))            # for `oracle` (cross-check only)


def _load_synth():
    p = REPO / "clip-neural-image-conpression_amd" / "clip_feature_codec" / "utils" / "synth.py"
    spec = importlib.util.spec_from_file_location("ccn_synth", p)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--skip-c2", action="store_true", help="skip the 256px / 50-step fixture (~1 min)")
    args = ap.parse_args()
    torch.manual_seed(0)
    torch.set_num_threads(8)
    synth = _load_synth()

    from clip_feature_codec.models.unet import CLIPCondUNet, timestep_embedding
    from clip_feature_codec.models.blocks import ResBlock, FiLM
    from clip_feature_codec.diffusion.scheduler import NoiseScheduler
    from clip_feature_codec.diffusion.ddim import DDIMSampler
    from clip_feature_codec.eval.metrics import psnr, _to_uint8
    import clip_feature_codec
    assert "/root/reference" in clip_feature_codec.__file__, clip_feature_codec.__file__

    from oracle import ref_unet, ref_diffusion, ref_codec

    def ref_net(base, ch_mult, z_dim=512, seed=0):
        net = CLIPCondUNet(z_dim=z_dim, base=base, ch_mult=ch_mult, img_ch=3).eval()
        spec = synth.unet_param_spec(z_dim, base, ch_mult)
        ref_keys = [(k, tuple(v.shape)) for k, v in net.state_dict().items()]
        assert ref_keys == [(k, tuple(s)) for k, s in spec], "param spec differs from the reference state dict"
        sd = synth.synth_state_dict(spec, seed=seed)
        net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
        return net, sd

    # ------------------------------------------------------------------ 1. scheduler tables, ts, temb
    out = {}
    for sched in ("cosine", "linear"):
        s = NoiseScheduler(1000, sched, device="cpu")
        mine = ref_diffusion.scheduler_tables(1000, sched)
        for name in ref_diffusion.TABLE_NAMES:
            ref = getattr(s, name)
            assert torch.equal(ref, mine[name]), (sched, name)
            out[f"{sched}.{name}"] = ref.numpy()
    for steps in (10, 50, 100):
        ts = torch.linspace(999, 0, steps).long()
        out[f"ts.{steps}"] = ts.numpy()
        out[f"temb.{steps}"] = timestep_embedding(ts, 256).numpy()
        # per-step DDIM coefficients as THIS host's torch computes them from the reference tables (0-d tensor sqrt
        # results were seen to differ by 1 ulp on another CPU model, so the bit-exact update tests read them from here)
        out[f"coef.{steps}"] = ref_diffusion.ddim_coefficients({k: getattr(NoiseScheduler(1000, "cosine", "cpu"), k)
                                                                for k in ref_diffusion.TABLE_NAMES}, steps)
        assert torch.equal(ref_unet.timestep_embedding(ts, 256), timestep_embedding(ts, 256))
    np.savez_compressed(HERE / "scheduler.npz", **out)
    print("scheduler.npz", sum(v.nbytes for v in out.values()))

    # ------------------------------------------------------------------ 2. tiny UNet with taps (per-op goldens)
    net, sd = ref_net(32, (1, 2))
    g = torch.Generator("cpu").manual_seed(11)
    x = torch.randn((2, 3, 16, 16), generator=g)
    z = torch.from_numpy(synth.synth_z(2, 512, seed=77))
    t = torch.tensor([999, 417])
    taps = {}
    hooks = []

    def hook(name):
        def f(mod, inp, outp):
            taps[name] = outp.detach().clone()
        return f
    for name in ["in_conv", "down.0", "down.1", "down.2", "down.3", "down.5", "mid1", "mid2",
                 "up.0", "up.1", "up.2", "up.5", "out_norm"]:
        mod = net
        for part in name.split("."):
            mod = getattr(mod, part) if not part.isdigit() else mod[int(part)]
        hooks.append(mod.register_forward_hook(hook(name)))
    with torch.no_grad():
        eps = net(x, z, t)
    for h in hooks:
        h.remove()
    mine_taps = {}
    with torch.no_grad():
        eps_mine = ref_unet.unet_forward(ref_unet.as_torch_sd(sd), x, z, t, tap=lambda n, v: mine_taps.__setitem__(n, v))
    assert torch.equal(eps, eps_mine), float((eps - eps_mine).abs().max())
    assert torch.equal(taps["down.0"], mine_taps["down.0.out"])
    assert torch.equal(taps["down.2"], mine_taps["down.2"])
    fx = {"x": x.numpy(), "z": z.numpy(), "t": t.numpy(), "eps": eps.numpy()}
    fx.update({f"tap.{k}": v.numpy() for k, v in taps.items()})
    np.savez_compressed(HERE / "unet_tiny_taps.npz", **fx)
    print("unet_tiny_taps.npz", sum(v.nbytes for v in fx.values()))

    # standalone FiLM / ResBlock (reference classes, synthetic params via the same generator)
    rb = ResBlock(32, 256).eval()
    spec_rb = [(k, tuple(v.shape)) for k, v in rb.state_dict().items()]
    sd_rb = synth.synth_state_dict([(f"down.0.{k}", s) for k, s in spec_rb], seed=3)
    rb.load_state_dict({k[len("down.0."):]: torch.from_numpy(v) for k, v in sd_rb.items()})
    xr = torch.randn((2, 32, 16, 16), generator=g)
    hr = torch.randn((2, 256), generator=g)
    with torch.no_grad():
        yr = rb(xr, hr)
        yf = rb.film(xr, hr)
        ym = ref_unet.resblock(ref_unet.as_torch_sd(sd_rb), "down.0", xr, hr)
    assert torch.equal(yr, ym)
    np.savez_compressed(HERE / "resblock.npz", x=xr.numpy(), h=hr.numpy(), y=yr.numpy(), film=yf.numpy())

    # ------------------------------------------------------------------ 3. C1: 64px, base 32, (1,2), 10 steps, B=1
    xT = torch.from_numpy(synth.start_noise([0], 64, seed_base=100))
    z1 = torch.from_numpy(synth.synth_z(1, 512, seed=1234))
    sch = NoiseScheduler(1000, "cosine", device="cpu")
    rec = {"eps": [], "x_in": []}

    class Recorder(torch.nn.Module):
        def __init__(self, inner):
            super().__init__()
            self.inner = inner

        def forward(self, x_, z_, t_):
            e = self.inner(x_, z_, t_)
            rec["x_in"].append(x_.detach().clone())
            rec["eps"].append(e.detach().clone())
            return e
    t0 = time.time()
    x_fin = DDIMSampler(sch, eta=0.0).sample(Recorder(net), z1, (1, 3, 64, 64), steps=10, x_T=xT)
    print(f"C1 reference sample: {time.time() - t0:.2f}s")
    xs = torch.stack(rec["x_in"][1:] + [x_fin], 0)       # x after step 1..10
    steps_rec = {}
    x_or = ref_diffusion.ddim_sample(ref_unet.make_model(ref_unet.as_torch_sd(sd)), z1, xT, steps=10,
                                     record=lambda i, tt, e, xn: steps_rec.__setitem__(i, (e, xn)))
    assert torch.equal(x_or, x_fin), float((x_or - x_fin).abs().max())
    for i in range(10):
        assert torch.equal(steps_rec[i][1], xs[i])
    np.savez_compressed(HERE / "c1_sample.npz", x_T=xT.numpy(), z=z1.numpy(), eps=torch.stack(rec["eps"], 0).numpy(),
                        x_steps=xs.numpy(), x_final=x_fin.numpy())
    # conditioning: the same net in fp64 -> how far does fp32 drift over the loop?
    net64 = CLIPCondUNet(512, 32, (1, 2)).double().eval()
    net64.load_state_dict({k: torch.from_numpy(v).double() for k, v in sd.items()})

    class As64(torch.nn.Module):
        def forward(self, x_, z_, t_):
            return net64(x_.double(), z_.double(), t_).float()
    x64 = DDIMSampler(sch, 0.0).sample(As64(), z1, (1, 3, 64, 64), steps=10, x_T=xT)
    print("C1 fp64-net vs fp32 after 10 steps: max-abs", float((x64 - x_fin).abs().max()))

    # ------------------------------------------------------------------ 4. PSNR / uint8 known answers
    rng = np.random.default_rng(5)
    a = rng.uniform(-1.2, 1.2, (3, 24, 24)).astype(np.float32)
    b = (a + rng.normal(0, 0.05, a.shape)).astype(np.float32)
    c = rng.uniform(-1, 1, (3, 64, 64)).astype(np.float32)
    kat = {"a": a, "b": b, "c": c, "u8_a": _to_uint8(a), "psnr_ab": np.float64(psnr(a, b)),
           "psnr_aa": np.float64(psnr(a, a)),
           "psnr_x10_c": np.float64(psnr(x_fin[0].clamp(-1, 1).numpy(), c))}
    assert ref_codec.psnr(a, b) == float(kat["psnr_ab"])
    assert np.array_equal(ref_codec.to_uint8(a), kat["u8_a"])
    np.savez_compressed(HERE / "psnr_kat.npz", **kat)

    # ------------------------------------------------------------------ 5. C2: 256px, base 128, (1,2,2), B=1
    if not args.skip_c2:
        net2, sd2 = ref_net(128, (1, 2, 2))
        xT2 = torch.from_numpy(synth.start_noise([0], 256, seed_base=100))
        z2 = torch.from_numpy(synth.synth_z(1, 512, seed=1234))
        with torch.no_grad():
            e0 = net2(xT2, z2, torch.tensor([999]))
            e1 = net2(xT2 * 0.5, z2, torch.tensor([500]))
        t0 = time.time()
        xf2 = DDIMSampler(NoiseScheduler(1000, "cosine", "cpu"), 0.0).sample(net2, z2, (1, 3, 256, 256), steps=50, x_T=xT2)
        print(f"C2 reference 50-step sample: {time.time() - t0:.1f}s")

        def pack(name, v):
            v = v[0]
            return {f"{name}.sub": v[:, ::4, ::4].numpy(),
                    f"{name}.mean": v.double().mean((1, 2)).numpy(), f"{name}.var": v.double().var((1, 2)).numpy(),
                    f"{name}.absmax": v.abs().amax((1, 2)).numpy(), f"{name}.abssum": np.float64(v.double().abs().sum())}
        c2 = {}
        c2.update(pack("eps_t999", e0)); c2.update(pack("eps_t500_halfx", e1)); c2.update(pack("x_final", xf2))
        np.savez_compressed(HERE / "c2_sample.npz", **c2)
        print("c2_sample.npz", sum(v.nbytes for v in c2.values()))
    print("done")


if __name__ == "__main__":
    main()
