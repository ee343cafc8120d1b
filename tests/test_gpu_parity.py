"""GPU parity: the HIP path, called through the C ABI (ctypes), against the CPU oracle and the reference's
golden vectors.  fp32 mode is the parity mode (tolerances written at each assert); bf16 mode is the throughput
mode and is checked against looser, stated bounds (a bf16 forward of the REFERENCE itself differs from fp32
by 2.7e-3 max-abs on these weights)."""
import ctypes

import numpy as np
import pytest
import torch

from clip_feature_codec import _native
from clip_feature_codec.models.unet import CLIPCondUNet, timestep_embedding
from clip_feature_codec.models.blocks import ResBlock, FiLM
from clip_feature_codec.diffusion.scheduler import NoiseScheduler
from clip_feature_codec.diffusion.ddim import DDIMSampler
from oracle import ref_unet, ref_diffusion

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

# fp32-mode tolerances.  eps is O(0.1-0.4); activations are O(1-5).
TOL_EPS_FP32 = 2e-5
TOL_ACT_FP32 = 1e-4
TOL_E2E_FP32 = 1e-3       # BASELINE.json north_star: max-abs on the reconstructed tensor


def to_dev(a):
    return (a if isinstance(a, torch.Tensor) else torch.from_numpy(np.asarray(a))).to(DEV)


def make_net(sd, base, ch_mult, dtype="fp32", z_dim=512):
    net = CLIPCondUNet(z_dim=z_dim, base=base, ch_mult=ch_mult, dtype=dtype).to(DEV).eval()
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    return net


def maxerr(a, b):
    return float((a.detach().cpu().double() - b.detach().cpu().double()).abs().max())


@pytest.fixture(scope="module")
def tiny_net(tiny_sd):
    return make_net(tiny_sd, 32, (1, 2))


def test_native_library_is_loaded():
    lib = _native.load_library()
    assert b"gfx950" in lib.ccn_version()
    with open("/proc/self/maps") as f:
        assert "libccn_hip.so" in f.read()


def test_timestep_embedding(golden):
    g = golden("scheduler.npz")
    for steps in (10, 50):
        emb = timestep_embedding(to_dev(g[f"ts.{steps}"].astype(np.int64)), 256).cpu().numpy()
        # device expf/cosf/sinf vs the CPU's: a few ulp of the argument at t*f up to 999 rad
        assert np.abs(emb - g[f"temb.{steps}"]).max() < 2e-4
        assert np.abs(emb[:, :8] - g[f"temb.{steps}"][:, :8]).max() < 1.5e-4
    assert timestep_embedding(to_dev(np.array([5], np.int64)), 7).shape == (1, 7)     # odd dim: zero pad
    assert float(timestep_embedding(to_dev(np.array([5], np.int64)), 7)[0, 6]) == 0.0


def test_film_operator(golden, synth):
    g = golden("resblock.npz")
    film = FiLM(32, 256).to(DEV)
    spec = [(f"down.0.film.{k}", tuple(v.shape)) for k, v in film.state_dict().items()]
    sd = synth.synth_state_dict(spec, seed=3)
    film.load_state_dict({k[len("down.0.film."):]: torch.from_numpy(v) for k, v in sd.items()})
    y = film(to_dev(g["x"]), to_dev(g["h"]))
    assert y.shape == g["x"].shape                         # the reference's own test (tests/test_blocks.py:10)
    assert maxerr(y, torch.from_numpy(g["film"])) < 5e-6


@pytest.mark.parametrize("dtype,tol", [("fp32", 3e-5), ("bf16", 8e-2)])
def test_resblock_operator(golden, synth, dtype, tol):
    g = golden("resblock.npz")
    rb = ResBlock(32, 256).to(DEV)
    rb.compute_dtype = dtype
    spec = [(f"down.0.{k}", tuple(v.shape)) for k, v in rb.state_dict().items()]
    sd = synth.synth_state_dict(spec, seed=3)
    rb.load_state_dict({k[len("down.0."):]: torch.from_numpy(v) for k, v in sd.items()})
    y = rb(to_dev(g["x"]), to_dev(g["h"]))
    assert maxerr(y, torch.from_numpy(g["y"])) < tol, maxerr(y, torch.from_numpy(g["y"]))


def test_unet_taps_fp32(golden, tiny_net):
    """Every intermediate of the tiny UNet vs the reference's forward-hook captures, in forward order."""
    g = golden("unet_tiny_taps.npz")
    eps = tiny_net(to_dev(g["x"]), to_dev(g["z"]), to_dev(g["t"]))
    report = []
    for name in ["in_conv", "down.0", "down.1", "down.2", "down.3", "down.5", "mid1", "mid2", "up.0", "up.1"]:
        ref = g[f"tap.{name}"]
        got = tiny_net.read_activation(name, ref.shape)
        report.append((name, maxerr(got, torch.from_numpy(ref)), float(np.abs(ref).max())))
    # the reference's hooks see the ConvTranspose output BEFORE the skip add; ours is after: compare via the next tap
    err_eps = maxerr(eps, torch.from_numpy(g["eps"]))
    msg = "; ".join(f"{n}: err {e:.2e} (|ref| {m:.1f})" for n, e, m in report) + f"; eps err {err_eps:.2e}"
    for n, e, m in report:
        assert e < TOL_ACT_FP32, msg
    assert eps.shape == g["x"].shape and err_eps < TOL_EPS_FP32, msg


@pytest.mark.parametrize("base,ch_mult,B,H,W", [
    (32, (1, 2), 3, 24, 40),        # W not a multiple of the 32-pixel tile, H not of 4 at the deeper levels
    (48, (2, 1), 2, 16, 16),        # 6 and 12 channels per group, Cout = 96 (N-tile masking)
    (64, (1, 2, 2), 1, 32, 64),     # three levels, 256 channels at the bottleneck (two N tiles, group spans)
    (16, (1,), 2, 8, 8),            # 2 channels per group, single level
])
def test_forward_shapes_vs_oracle_fp32(synth, base, ch_mult, B, H, W):
    sd = synth.synth_state_dict(synth.unet_param_spec(512, base, ch_mult), seed=1)
    net = make_net(sd, base, ch_mult)
    g = torch.Generator().manual_seed(base + H)
    x = torch.randn((B, 3, H, W), generator=g); z = torch.from_numpy(synth.synth_z(B, seed=9))
    t = torch.randint(0, 1000, (B,), generator=g)
    with torch.no_grad():
        ref = ref_unet.unet_forward(ref_unet.as_torch_sd(sd), x, z, t)
    eps = net(x.to(DEV), z.to(DEV), t.to(DEV))
    assert eps.shape == x.shape and eps.dtype == torch.float32 and eps.device == torch.device(DEV)
    assert maxerr(eps, ref) < TOL_EPS_FP32, maxerr(eps, ref)


def test_forward_bf16_vs_oracle(golden, tiny_sd):
    g = golden("unet_tiny_taps.npz")
    net = make_net(tiny_sd, 32, (1, 2), dtype="bf16")
    eps = net(to_dev(g["x"]), to_dev(g["z"]), to_dev(g["t"]))
    err = maxerr(eps, torch.from_numpy(g["eps"]))
    assert err < 1e-2, err          # bf16 storage + bf16 MFMA inputs; |eps| ~ 0.2


def test_ddim_step_bit_exact(golden):
    """The update kernel reproduces the reference's fp32 op sequence bit for bit.  The coefficient table is
    read from the golden fixture: torch-CPU results (cos/cumprod, even a 0-d sqrt) were seen to differ by 1 ulp
    between the build container's CPU and the GPU box's, so "the reference's bits" are those of the host that ran it."""
    g = golden("c1_sample.npz")
    coefs = golden("scheduler.npz")["coef.10"]
    x = torch.from_numpy(g["x_T"]).clone()
    for i in range(10):
        xd = to_dev(x).clone()
        _native.ddim_step(xd, to_dev(g["eps"][i]), coefs[i, :4])
        assert np.array_equal(xd.cpu().numpy(), g["x_steps"][i]), f"step {i}"      # byte-identical to the reference
        x = torch.from_numpy(g["x_steps"][i])


def test_q_sample_and_predict_x0_bit_exact():
    sch = NoiseScheduler(1000, "cosine", DEV)
    tab = ref_diffusion.scheduler_tables()
    g = torch.Generator().manual_seed(3)
    x0 = torch.randn((3, 3, 16, 16), generator=g); n = torch.randn((3, 3, 16, 16), generator=g); t = torch.tensor([0, 500, 999])
    xt = sch.q_sample(x0.to(DEV), t.to(DEV), n.to(DEV))
    assert torch.equal(xt.cpu(), ref_diffusion.q_sample(tab, x0, t, n))
    back = sch.predict_x0_from_eps(xt, t.to(DEV), n.to(DEV))
    assert torch.equal(back.cpu(), ref_diffusion.predict_x0_from_eps(tab, xt.cpu(), t, n))


def test_c1_teacher_forced_and_end_to_end_fp32(golden, tiny_net):
    """BASELINE config 1 shapes: 64 px, base 32, (1,2), 10 steps.  Per-step (teacher-forced) and end-to-end."""
    g = golden("c1_sample.npz")
    z = to_dev(g["z"])
    ts = ref_diffusion.ddim_timesteps(1000, 10)
    x_in = [g["x_T"]] + [g["x_steps"][i] for i in range(9)]
    worst = 0.0
    for i in range(10):
        eps = tiny_net(to_dev(x_in[i]), z, to_dev(np.array([ts[i]], np.int64)))
        worst = max(worst, maxerr(eps, torch.from_numpy(g["eps"][i])))
    assert worst < TOL_EPS_FP32, worst
    sampler = DDIMSampler(NoiseScheduler(1000, "cosine", DEV), eta=0.0)
    x = sampler.sample(tiny_net, z, (1, 3, 64, 64), steps=10, x_T=to_dev(g["x_T"]))
    assert maxerr(x, torch.from_numpy(g["x_final"])) < TOL_E2E_FP32, maxerr(x, torch.from_numpy(g["x_final"]))


def test_graph_equals_stepwise_and_is_deterministic(golden, tiny_net):
    g = golden("c1_sample.npz")
    z, xT = to_dev(g["z"]), to_dev(g["x_T"])
    sampler = DDIMSampler(NoiseScheduler(1000, "cosine", DEV), eta=0.0)
    a = sampler.sample(tiny_net, z, (1, 3, 64, 64), steps=10, x_T=xT)
    b = sampler.sample(tiny_net, z, (1, 3, 64, 64), steps=10, x_T=xT)          # graph replay
    sampler.use_graph = False
    c = sampler.sample(tiny_net, z, (1, 3, 64, 64), steps=10, x_T=xT)          # launch by launch
    d = sampler.sample(lambda x, zz, t: tiny_net(x, zz, t), z, (1, 3, 64, 64), steps=10, x_T=xT)   # generic callable route
    assert torch.equal(a, b) and torch.equal(a, c)
    assert maxerr(a, d) < 1e-5      # conditioning via the int64 forward path instead of the hoisted table


def test_batch_rows_are_independent(synth, tiny_net):
    """Independent units: a batch of 3 equals three batches of 1, bit for bit (what makes rank sharding exact)."""
    z = to_dev(synth.synth_z(3, seed=50)); xT = to_dev(synth.start_noise([7, 8, 9], 32, seed_base=1))
    sampler = DDIMSampler(NoiseScheduler(1000, "cosine", DEV), eta=0.0)
    full = sampler.sample(tiny_net, z, (3, 3, 32, 32), steps=4, x_T=xT)
    for i in range(3):
        one = sampler.sample(tiny_net, z[i:i + 1], (1, 3, 32, 32), steps=4, x_T=xT[i:i + 1])
        assert torch.equal(one[0], full[i]), i


def test_eta_positive_fused_graph_matches_oracle_and_stepwise(tiny_net, tiny_sd, synth):
    """eta > 0 (diffusion/ddim.py:41-45) through the fused graph (ccn_sample_eta): the per-step N(0,1) draws are made up front, one
    ``normal_()`` per noisy step in loop order.  (a) Against the CPU oracle fed the SAME draws: the C1 gate of the eta = 0 path;
    (b) against the step-by-step route (generic callable + ccn_ddim_step, ``torch.randn_like`` per step) after re-seeding the device
    generator: equal to the 1e-5 of the two conditioning paths, which shows the generator is consumed identically; (c) the second
    call replays the captured graph (same noise buffer address) and, re-seeded, reproduces the first bit for bit."""
    # (the reference's direction term sqrt(ab_s - sigma^2) with ab_s = alphas_cumprod_prev[t] goes NaN at the noisy end of the
    # schedule unless eta^2 beta_t < ab_s: linear schedule, eta = 0.03 keeps every step real)
    B, S, T, eta = 2, 32, 6, 0.03
    z = to_dev(synth.synth_z(B)); xT = to_dev(synth.start_noise(range(B), S))
    sch = NoiseScheduler(1000, "linear", DEV)
    assert np.isfinite(sch.ddim_coefficients(T, eta)).all()
    assert (sch.ddim_coefficients(T, eta)[:-1, 4] > 0).all() and sch.ddim_coefficients(T, eta)[-1, 4] == 0    # last step: no noise
    sampler = DDIMSampler(sch, eta=eta)
    torch.manual_seed(1234)
    a = sampler.sample(tiny_net, z, (B, 3, S, S), steps=T, x_T=xT)
    draws = next(iter(sampler._noise.values())).cpu().clone()               # what the fused loop consumed
    it = iter(range(T))
    ref = ref_diffusion.ddim_sample(ref_unet.make_model(ref_unet.as_torch_sd(tiny_sd)), z.cpu(), xT.cpu(), steps=T, schedule="linear", eta=eta,
                                    noise_fn=lambda x: draws[next(it)])
    assert maxerr(a, ref) < TOL_E2E_FP32, maxerr(a, ref)
    torch.manual_seed(1234)
    b = sampler.sample(lambda x, zz, t: tiny_net(x, zz, t), z, (B, 3, S, S), steps=T, x_T=xT)      # stepwise route
    assert maxerr(a, b) < 1e-5, maxerr(a, b)
    torch.manual_seed(1234)
    c = sampler.sample(tiny_net, z, (B, 3, S, S), steps=T, x_T=xT)
    assert torch.equal(a, c)
    torch.manual_seed(99)
    d = sampler.sample(tiny_net, z, (B, 3, S, S), steps=T, x_T=xT)
    assert maxerr(a, d) > 1e-3                                              # other draws, other sample


def test_error_paths(tiny_net):
    with pytest.raises(ValueError):
        tiny_net(torch.zeros(1, 3, 30, 30, device=DEV), torch.zeros(1, 512, device=DEV), torch.zeros(1, dtype=torch.long, device=DEV))
    with pytest.raises(ValueError):
        tiny_net(torch.zeros(2, 3, 32, 32, device=DEV), torch.zeros(1, 512, device=DEV), torch.zeros(2, dtype=torch.long, device=DEV))
    nat = _native.NativeUNet(512, 32, (1, 2), 256, 3, device=DEV)
    with pytest.raises(RuntimeError, match="Missing key"):
        nat.load_state_dict({"out.bias": torch.zeros(3)})
    with pytest.raises(RuntimeError, match="Unexpected key"):
        nat.load_state_dict({"nope": torch.zeros(3)})
    with pytest.raises(RuntimeError, match="size mismatch"):
        nat.load_state_dict({"out.bias": torch.zeros(4)})


# ---------------------------------------------------------------- BASELINE config 2 sizes (256 px, base 128, (1,2,2))
@pytest.fixture(scope="module")
def c2_sd(synth):
    return synth.synth_state_dict(synth.unet_param_spec(512, 128, (1, 2, 2)))


def _check_packed(name, got, g, tol):
    got = got[0].detach().cpu()
    sub = got[:, ::4, ::4].numpy()
    err = float(np.abs(sub - g[f"{name}.sub"]).max())
    mean_err = float(np.abs(got.double().mean((1, 2)).numpy() - g[f"{name}.mean"]).max())
    abssum = float(got.double().abs().sum())
    assert err < tol, (name, err)
    assert mean_err < tol and abs(abssum / float(g[f"{name}.abssum"]) - 1) < 1e-3, (name, mean_err, abssum)
    return err


def test_c2_forward_fp32_vs_reference(golden, synth, c2_sd):
    g = golden("c2_sample.npz")
    net = make_net(c2_sd, 128, (1, 2, 2))
    xT = to_dev(synth.start_noise([0], 256, seed_base=100)); z = to_dev(synth.synth_z(1))
    e0 = net(xT, z, to_dev(np.array([999], np.int64)))
    e1 = net(xT * 0.5, z, to_dev(np.array([500], np.int64)))
    _check_packed("eps_t999", e0, g, TOL_EPS_FP32)
    _check_packed("eps_t500_halfx", e1, g, TOL_EPS_FP32)


def test_c2_50_step_fp32_vs_reference(golden, synth, c2_sd):
    """The headline parity gate: 256 px, 50 DDIM steps, same seed / z as the reference's device='cpu' run."""
    g = golden("c2_sample.npz")
    net = make_net(c2_sd, 128, (1, 2, 2))
    xT = to_dev(synth.start_noise([0], 256, seed_base=100)); z = to_dev(synth.synth_z(1))
    x = DDIMSampler(NoiseScheduler(1000, "cosine", DEV), 0.0).sample(net, z, (1, 3, 256, 256), steps=50, x_T=xT)
    err = _check_packed("x_final", x, g, TOL_E2E_FP32)
    print(f"C2 50-step fp32 max-abs vs reference: {err:.3e}")


def test_c2_bf16_reported_deviation(golden, synth, c2_sd):
    """bf16 throughput mode at full size: bounded, and reported.  The reference under bf16 autocast deviates from
    its own fp32 run by 2.7e-3 per forward and 0.30 max-abs / 0.037 mean-abs over 50 steps on these weights."""
    g = golden("c2_sample.npz")
    net = make_net(c2_sd, 128, (1, 2, 2), dtype="bf16")
    xT = to_dev(synth.start_noise([0], 256, seed_base=100)); z = to_dev(synth.synth_z(1))
    e0 = net(xT, z, to_dev(np.array([999], np.int64)))
    err = float(np.abs(e0[0, :, ::4, ::4].cpu().numpy() - g["eps_t999.sub"]).max())
    assert err < 2e-2, err
    x = DDIMSampler(NoiseScheduler(1000, "cosine", DEV), 0.0).sample(net, z, (1, 3, 256, 256), steps=50, x_T=xT)
    d = np.abs(x[0, :, ::4, ::4].cpu().numpy() - g["x_final.sub"])
    print(f"C2 bf16: forward max-abs {err:.3e}; 50-step max-abs {d.max():.3e} mean-abs {d.mean():.3e}")
    # measured on MI355X (round 2): max-abs 0.21, mean-abs 0.032; the bound is 2x the measured mean
    assert d.mean() < 0.065 and d.max() < 0.6 and np.isfinite(d).all()


def test_c2_batch8_fp32_large_tiles_vs_reference(golden, synth, c2_sd):
    """Batch 8 at 256 px is the bench workload and the only shape that selects the 8-row (256-pixel) tiles.
    Record 0 of the batch has the golden inputs, so its eps must match the reference; the other rows must match
    their own batch-1 evaluation (different tile partition => different fp32 summation order of the GroupNorm
    partial sums, hence a tolerance instead of bit equality)."""
    g = golden("c2_sample.npz")
    net = make_net(c2_sd, 128, (1, 2, 2))
    z = to_dev(synth.synth_z(8)); xT = to_dev(synth.start_noise(range(8), 256, seed_base=100))
    t = to_dev(np.full((8,), 999, np.int64))
    eps = net(xT, z, t)
    _check_packed("eps_t999", eps[0:1], g, TOL_EPS_FP32)
    one = net(xT[5:6], z[5:6], t[5:6])
    assert maxerr(one[0], eps[5]) < 1e-5, maxerr(one[0], eps[5])


def test_c2_batch8_bf16_is_finite_and_close_to_batch1(synth, c2_sd):
    net = make_net(c2_sd, 128, (1, 2, 2), dtype="bf16")
    z = to_dev(synth.synth_z(8)); xT = to_dev(synth.start_noise(range(8), 256, seed_base=100))
    sampler = DDIMSampler(NoiseScheduler(1000, "cosine", DEV), 0.0)
    full = sampler.sample(net, z, (8, 3, 256, 256), steps=3, x_T=xT)
    one = sampler.sample(net, z[5:6], (1, 3, 256, 256), steps=3, x_T=xT[5:6])
    assert torch.isfinite(full).all() and maxerr(one[0], full[5]) < 5e-2, maxerr(one[0], full[5])


def test_persistent_kernel_on_ragged_tiles_matches_free_running_and_fp32(synth, c2_sd):
    """The persistent register-weight kernel (bf16, 8-row tiles of 32 pixels, csrc/ccn_conv_pr.hip) on a batch whose
    feature maps end in PARTIAL column tiles (200 x 168 -> 25 x 5.25 tiles) and partial row tiles (100 x 84 at the next level
    -> 12.5 x 2.6 tiles; sizes must be multiples of 8 for the three stride-2 levels), against
    (a) the free-running kernel on the same bf16 data path and (b) the fp32 parity mode.  Bounds: the two bf16 kernels
    differ only by one extra bf16 rounding of the conv accumulator and the summation order of the GroupNorm partials
    (measured 2.1e-3 max-abs at 256 px); bf16 vs fp32 is the documented 2e-2 forward bound."""
    lib = _native.load_library()
    lib.ccn_internal_set_conv_variant.restype = ctypes.c_int
    B, H, W = 8, 200, 168
    g = torch.Generator("cpu").manual_seed(3)
    x = to_dev(torch.randn((B, 3, H, W), generator=g)); z = to_dev(synth.synth_z(B))
    t = to_dev(np.array([999, 800, 650, 500, 350, 200, 50, 0], np.int64))
    old = lib.ccn_internal_set_conv_variant(3)
    try:
        e_fr = make_net(c2_sd, 128, (1, 2, 2), dtype="bf16")(x, z, t)
        lib.ccn_internal_set_conv_variant(4)
        e_pr = make_net(c2_sd, 128, (1, 2, 2), dtype="bf16")(x, z, t)
    finally:
        lib.ccn_internal_set_conv_variant(old)
    e32 = make_net(c2_sd, 128, (1, 2, 2))(x, z, t)
    assert torch.isfinite(e_pr).all()
    d_kernels, d_fp32, d_fr32 = maxerr(e_pr, e_fr), maxerr(e_pr, e32), maxerr(e_fr, e32)
    print(f"ragged 200x168 bf16: persistent vs free-running {d_kernels:.3e}; vs fp32 {d_fp32:.3e} (free-running vs fp32 {d_fr32:.3e})")
    assert d_kernels < 8e-3 and d_fp32 < 2e-2, (d_kernels, d_fp32)
    # image borders: the rows / columns covered by partial tiles must be as accurate as the interior
    edge = max(maxerr(e_pr[:, :, -4:, :], e32[:, :, -4:, :]), maxerr(e_pr[:, :, :, -8:], e32[:, :, :, -8:]))
    assert edge < 2e-2, edge


def test_other_width_base192_two_levels(synth):
    """A width the kernels were not tuned for (base 192, ch_mult (1,2): 192 and 384 channels -> a half-empty second N tile
    in the persistent kernel, 24/48 channels per GroupNorm group, generic stem/head because 192 is not 32/64/128):
    fp32 mode against the CPU oracle on one sample, bf16 mode against fp32 on the batch."""
    sd = synth.synth_state_dict(synth.unet_param_spec(512, 192, (1, 2)))
    B, S = 4, 128
    g = torch.Generator("cpu").manual_seed(7)
    x = torch.randn((B, 3, S, S), generator=g); z = torch.from_numpy(synth.synth_z(B)); t = torch.tensor([999, 600, 300, 10])
    e32 = make_net(sd, 192, (1, 2))(to_dev(x), to_dev(z), to_dev(t))
    with torch.no_grad():
        ref = ref_unet.unet_forward(ref_unet.as_torch_sd(sd), x[1:2], z[1:2], t[1:2])
    assert maxerr(e32[1:2], ref) < TOL_EPS_FP32, maxerr(e32[1:2], ref)
    e16 = make_net(sd, 192, (1, 2), dtype="bf16")(to_dev(x), to_dev(z), to_dev(t))
    d = maxerr(e16, e32)
    print(f"base 192 @128px: fp32 vs oracle {maxerr(e32[1:2], ref):.2e}; bf16 vs fp32 {d:.3e}")
    assert torch.isfinite(e16).all() and d < 2e-2, d


def test_c4_architecture_four_levels_base192(synth):
    """BASELINE.json configs[3]'s architecture (base 192, ch_mult (1,2,2,4): widths 192/192/384/768/3072, 815.7 M
    parameters, four resolution levels, 384 channels per GroupNorm group at the bottom) at a 64 px input so that the
    oracle finishes in seconds: fp32 mode against the CPU oracle on one sample, bf16 mode against fp32 on the batch."""
    spec = synth.unet_param_spec(512, 192, (1, 2, 2, 4))
    assert sum(int(np.prod(s)) for _, s in spec) == 815_721_475
    sd = synth.synth_state_dict(spec)
    B, S = 2, 64
    g = torch.Generator("cpu").manual_seed(5)
    x = torch.randn((B, 3, S, S), generator=g); z = torch.from_numpy(synth.synth_z(B)); t = torch.tensor([999, 250])
    e32 = make_net(sd, 192, (1, 2, 2, 4))(to_dev(x), to_dev(z), to_dev(t))
    with torch.no_grad():
        ref = ref_unet.unet_forward(ref_unet.as_torch_sd(sd), x[1:2], z[1:2], t[1:2])
    assert maxerr(e32[1:2], ref) < TOL_EPS_FP32, maxerr(e32[1:2], ref)
    e16 = make_net(sd, 192, (1, 2, 2, 4), dtype="bf16")(to_dev(x), to_dev(z), to_dev(t))
    d = maxerr(e16, e32)
    print(f"C4 architecture @64px: fp32 vs oracle {maxerr(e32[1:2], ref):.2e}; bf16 vs fp32 {d:.3e}")
    assert torch.isfinite(e16).all() and d < 2e-2, d


@pytest.mark.parametrize("B,H,W", [(2, 256, 256), (3, 200, 168)])
def test_half_padded_n_tile_192_channels(synth, B, H, W):
    """192-wide layers (C4's first level: N tiles of 128 + 64 channels) with more tiles than CUs: the consumer waves of the second N
    tile whose 32 channels are all padding only keep the barrier protocol (ccn_conv_pr.hip, idle_w).  bf16 mode against the fp32 mode
    (which runs other kernels), intermediates of both levels and eps; reference: models/unet.py:59-65 (widths), models/blocks.py:40-44."""
    sd = synth.synth_state_dict(synth.unet_param_spec(512, 192, (1, 2)))
    g = torch.Generator("cpu").manual_seed(B + H)
    x = torch.randn((B, 3, H, W), generator=g); z = torch.from_numpy(synth.synth_z(B)); t = torch.randint(0, 1000, (B,), generator=g)
    n32, n16 = make_net(sd, 192, (1, 2)), make_net(sd, 192, (1, 2), dtype="bf16")
    e32 = n32(to_dev(x), to_dev(z), to_dev(t)); e16 = n16(to_dev(x), to_dev(z), to_dev(t))
    for name, hh, ww in (("down.0", H, W), ("down.1", H, W), ("down.3", H // 2, W // 2), ("up.5", H, W)):
        a32 = n32.native().read_activation(name, (B, 192, hh, ww)); a16 = n16.native().read_activation(name, (B, 192, hh, ww))
        d = (a16 - a32).abs()
        # a wrong channel / tile mapping is O(1) everywhere; bf16 rounding is O(1e-2) on activations of O(1-5)
        assert float(d.max()) < 0.25 and float(d.mean()) < 1.5e-2, (name, float(d.max()), float(d.mean()))
    d = maxerr(e16, e32)
    assert torch.isfinite(e16).all() and d < 2e-2, d
    n16.native().poll_errors()


@pytest.mark.parametrize("B,H,W", [(1, 256, 256), (3, 72, 104), (8, 40, 48), (5, 136, 200), (2, 264, 136), (16, 128, 128)])
def test_c2_architecture_odd_shapes(synth, c2_sd, B, H, W):
    """Batch / image sizes that change which kernel takes each layer (tile counts decide between the persistent kernel, its
    split-K form, the 4-row kernel and the generic one) and that leave partial tiles everywhere: fp32 mode against the CPU
    oracle on the last sample of the batch, bf16 mode against fp32 on the whole batch."""
    g = torch.Generator("cpu").manual_seed(B * 1000 + H + W)
    x = torch.randn((B, 3, H, W), generator=g); z = torch.from_numpy(synth.synth_z(B)); t = torch.randint(0, 1000, (B,), generator=g)
    e32 = make_net(c2_sd, 128, (1, 2, 2))(to_dev(x), to_dev(z), to_dev(t))
    with torch.no_grad():
        ref = ref_unet.unet_forward(ref_unet.as_torch_sd(c2_sd), x[-1:], z[-1:], t[-1:])
    d32 = maxerr(e32[-1:], ref)
    assert d32 < TOL_EPS_FP32, (B, H, W, d32)
    e16 = make_net(c2_sd, 128, (1, 2, 2), dtype="bf16")(to_dev(x), to_dev(z), to_dev(t))
    d16 = maxerr(e16, e32)
    print(f"B={B} {H}x{W}: fp32 vs oracle {d32:.2e}; bf16 vs fp32 {d16:.3e}")
    assert torch.isfinite(e16).all() and d16 < 2e-2, (B, H, W, d16)



def test_c2_bench_workload_bf16_parity_numbers(golden, synth, c2_sd, tmp_path):
    """The exact bench.py workload (C2: batch 8, 256 px, 50 steps, bf16, fused graph on the persistent kernel) against
    (a) the reference's device='cpu' run of record 0 (golden), (b) this library's fp32 parity mode on all 8 rows, and
    (c) PSNR against the synthetic originals in both modes -- north_star asks for PSNR within 0.1 % (eval/metrics.py:22-29).
    Measured on MI355X (the same numbers are printed in bench.py's `parity` block on every run):
      fp32 mode row 0 vs reference: max-abs 2.9e-4 (gate 1e-3: met);
      bf16 row 0 vs reference: max-abs 0.21, mean-abs 0.032; bf16 vs fp32 mode, all rows: max-abs 0.32, mean-abs 0.032
      (the reference's own bf16-autocast run deviates 0.30 / 0.037 from its fp32 run on these weights);
      PSNR, per-record relative delta of bf16 mode against the fp32 mode ON THE FULL-PRECISION WEIGHTS: round 2 (conv weights
      rounded to bf16 independently, round-to-nearest-even) 0.11 % mean / 0.165 % max -- above the gate; round 3 (error-diffused
      rounding within the output channel and along the DDIM steps, tools/weight_rounding_probe.py) 0.041 % mean / 0.085 % max -- the gate
      is asserted as stated.
      With weight_rounding="nearest" the old shift is reproduced (asserted below: the diffusion is what closes it).
    Other bounds asserted: 2x the measured deviations."""
    from clip_feature_codec.eval.metrics import psnr
    g = golden("c2_sample.npz")
    B, S, T = 8, 256, 50
    z = to_dev(synth.synth_z(B)); xT = to_dev(synth.start_noise(range(B), S, seed_base=100))
    sampler = DDIMSampler(NoiseScheduler(1000, "cosine", DEV), 0.0)
    net16 = make_net(c2_sd, 128, (1, 2, 2), dtype="bf16")
    x16 = sampler.sample(net16, z, (B, 3, S, S), steps=T, x_T=xT)
    x32 = sampler.sample(make_net(c2_sd, 128, (1, 2, 2)), z, (B, 3, S, S), steps=T, x_T=xT)
    d32 = np.abs(x32[0, :, ::4, ::4].cpu().numpy() - g["x_final.sub"])
    d16 = np.abs(x16[0, :, ::4, ::4].cpu().numpy() - g["x_final.sub"])
    dall = (x16 - x32).abs()
    assert d32.max() < TOL_E2E_FP32, d32.max()                  # batch 8 fp32 mode: the headline gate, on the bench batch
    assert d16.mean() < 0.065 and d16.max() < 0.6, (d16.mean(), d16.max())
    assert float(dall.mean()) < 0.065 and float(dall.max()) < 0.8, (float(dall.mean()), float(dall.max()))
    orig = [synth.synth_image(i, S).astype(np.float32).transpose(2, 0, 1) / 127.5 - 1.0 for i in range(B)]
    r16, r32 = x16.clamp(-1, 1).cpu().numpy(), x32.clamp(-1, 1).cpu().numpy()
    p16 = np.array([psnr(orig[k], r16[k]) for k in range(B)]); p32 = np.array([psnr(orig[k], r32[k]) for k in range(B)])
    rel = np.abs(p16 - p32) / np.abs(p32)
    print(f"C2 bench workload: fp32 row0 vs ref {d32.max():.2e}; bf16 row0 vs ref max {d16.max():.3f} mean {d16.mean():.4f}; "
          f"bf16 vs fp32 all rows max {float(dall.max()):.3f} mean {float(dall.mean()):.4f}; "
          f"PSNR fp32 {p32.mean():.4f} dB bf16 {p16.mean():.4f} dB, rel delta mean {rel.mean():.2e} max {rel.max():.2e} "
          f"(0.1 % gate {'met' if rel.max() <= 1e-3 else 'NOT met'} by bf16 mode)")
    assert rel.max() <= 1e-3, rel                               # north_star's 0.1 % PSNR gate, bf16 mode vs fp32 mode on the fp32 weights
    # the same kernels on independently rounded weights: the shift the diffusion removes (measured 0.165 % max)
    net16n = CLIPCondUNet(z_dim=512, base=128, ch_mult=(1, 2, 2), dtype="bf16", weight_rounding="nearest").to(DEV).eval()
    net16n.load_state_dict({k: torch.from_numpy(v) for k, v in c2_sd.items()}, strict=True)
    x16n = sampler.sample(net16n, z, (B, 3, S, S), steps=T, x_T=xT).clamp(-1, 1).cpu().numpy()
    p16n = np.array([psnr(orig[k], x16n[k]) for k in range(B)])
    reln = np.abs(p16n - p32) / np.abs(p32)
    print(f"  weight_rounding='nearest': PSNR {p16n.mean():.4f} dB, rel delta mean {reln.mean():.2e} max {reln.max():.2e}")
    assert reln.mean() > 1.5 * rel.mean(), (reln.mean(), rel.mean())
    # fp32 mode vs the reference on record 0: PSNR within 0.1 % follows from max-abs < 1e-3 (uint8 truncation moves few pixels)
    torch.cuda.synchronize()
    net16.native().poll_errors()                                # no device-side failure (split-K hand-off timeout) was flagged


def test_weight_rounding_modes(synth, c2_sd):
    """ccn_set_weight_rounding (bf16 mode).  Version 0 of 'phases' IS the 'diffused' rounding, and ccn_forward always uses version 0:
    the two modes give bit-equal forwards; the fused sampler cycles the eight versions, so from step 1 on its trajectory differs from
    'diffused' -- by rounding-level amounts; 'nearest' differs from both already in one forward.  All three stay within the bf16 mode's
    per-forward bound against the fp32 mode."""
    B, S = 2, 64
    g = torch.Generator("cpu").manual_seed(3)
    x = to_dev(torch.randn((B, 3, S, S), generator=g)); z = to_dev(synth.synth_z(B)); t = to_dev(torch.tensor([900, 100]))
    nets = {}
    for mode in ("nearest", "diffused", "phases"):
        n = CLIPCondUNet(z_dim=512, base=128, ch_mult=(1, 2, 2), dtype="bf16", weight_rounding=mode).to(DEV).eval()
        n.load_state_dict({k: torch.from_numpy(v) for k, v in c2_sd.items()}, strict=True)
        nets[mode] = n
    e = {m: n(x, z, t) for m, n in nets.items()}
    e32 = make_net(c2_sd, 128, (1, 2, 2))(x, z, t)
    assert torch.equal(e["diffused"], e["phases"])
    assert not torch.equal(e["nearest"], e["diffused"])
    for m in e:
        assert maxerr(e[m], e32) < 2e-2, (m, maxerr(e[m], e32))
    sampler = DDIMSampler(NoiseScheduler(1000, "cosine", DEV), 0.0)
    xs = {m: sampler.sample(nets[m], z, (B, 3, S, S), steps=6, x_T=x) for m in ("diffused", "phases")}
    d = float((xs["diffused"] - xs["phases"]).abs().max())
    assert 0.0 < d < 0.1, d                                        # different versions from step 1 on, same model to rounding level
    one = sampler.sample(nets["phases"], z, (B, 3, S, S), steps=1, x_T=x)
    assert torch.equal(one, sampler.sample(nets["diffused"], z, (B, 3, S, S), steps=1, x_T=x))     # step 0 = version 0
    with pytest.raises(ValueError):
        CLIPCondUNet(z_dim=512, base=128, ch_mult=(1, 2, 2), dtype="bf16", weight_rounding="stochastic").to(DEV).native()
