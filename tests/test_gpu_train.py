"""Training step (BASELINE.json configs[4]) on the GPU through the C ABI: loss, every parameter gradient and the AdamW update
against the CPU oracle (oracle/ref_train.py, pinned to the reference by tests/golden/train_step.npz)."""
import sys
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.nn.functional as F

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "clip-neural-image-conpression_amd")]

from clip_feature_codec import _native  # noqa: E402
from clip_feature_codec.models.unet import CLIPCondUNet  # noqa: E402
from clip_feature_codec.diffusion.scheduler import NoiseScheduler  # noqa: E402
from clip_feature_codec.train.diffusion_train import FusedAdamW, train_step  # noqa: E402
from clip_feature_codec.utils import synth  # noqa: E402
from oracle import ref_unet, ref_train, ref_diffusion  # noqa: E402

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
GOLD = np.load(ROOT / "tests" / "golden" / "train_step.npz")


def make_net(sd, base, ch_mult, dtype="fp32"):
    net = CLIPCondUNet(z_dim=512, base=base, ch_mult=ch_mult, dtype=dtype).to(DEV)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    return net.train()


def rel_err(a, b):
    a = a.detach().cpu().double(); b = b.detach().cpu().double()
    return float((a - b).abs().max() / max(float(b.abs().max()), 1e-12))


def grads_via_autograd(net, x_t, z, t, target):
    net.zero_grad(set_to_none=True)
    eps = net(x_t.to(DEV), z.to(DEV), t.to(DEV))
    loss = F.mse_loss(eps, target.to(DEV))
    loss.backward()
    return loss.detach().cpu(), {k: p.grad.detach().cpu().clone() for k, p in net.named_parameters()}, eps.detach().cpu()


def check_grads(got, ref, tol, what):
    worst = ("", 0.0)
    for k, g in ref.items():
        e = rel_err(got[k], g)
        if e > worst[1]:
            worst = (k, e)
        assert e < tol, f"{what}: {k} relative max error {e:.3e} (|g|max {float(g.abs().max()):.3e})"
    print(f"{what}: worst parameter {worst[0]} rel err {worst[1]:.2e}")


def test_c1_shape_loss_and_all_gradients_fp32_match_reference_fixture_and_oracle():
    """The fixture's inputs (base 32, (1,2), 64 px, B=2): loss.backward() through CLIPCondUNet.forward in fp32 mode."""
    sd = synth.synth_state_dict(synth.unet_param_spec(512, 32, (1, 2)))
    x0, z, t, noise = (torch.from_numpy(GOLD[k]) for k in ("x0", "z", "t", "noise"))
    sch = NoiseScheduler(1000, "cosine", device=DEV)
    assert np.abs(sch.q_sample(x0.to(DEV), t.to(DEV), noise.to(DEV)).cpu().numpy() - GOLD["x_t"]).max() < 1e-6
    x_t = torch.from_numpy(GOLD["x_t"])
    net = make_net(sd, 32, (1, 2))
    loss, grads, eps = grads_via_autograd(net, x_t, z, t, noise)
    assert abs(float(loss) - float(GOLD["loss"])) < 2e-6
    rloss, rgrads, reps = ref_train.loss_and_grads(ref_unet.as_torch_sd(sd), x_t, z, t, noise)
    assert rel_err(eps, reps) < 1e-4
    check_grads(grads, rgrads, 2e-4, "fp32 C1 shape")
    for k in GOLD["names"]:                                    # and straight against what the reference produced
        k = str(k)
        g = grads[k].flatten()
        step = max(1, g.numel() // 64)
        scale = max(float(g.abs().max()), 1e-12)
        assert np.abs(g[::step][:64].numpy() - GOLD[f"gsample/{k}"]).max() <= 2e-4 * scale + 1e-9, k


@pytest.mark.parametrize("base,ch_mult,B,H,W", [(32, (1, 2), 3, 40, 72), (64, (1, 2, 2), 2, 64, 96), (96, (2,), 1, 36, 44),
                                                  (128, (1, 2), 2, 32, 48), (128, (2, 2), 1, 24, 40)])
def test_gradients_fp32_other_shapes(base, ch_mult, B, H, W):
    """Ragged tiles, three levels, channel counts that are not a power of two, batch of one; and the widths of the BASELINE configs
    (128 / 256 / 512 channels: GroupNorm groups of 16 / 32 / 64 channels, where the GroupNorm backward is the single-launch fused
    kernel, gn_bwd_fused_kernel) -- every gradient tensor against the oracle."""
    sd = synth.synth_state_dict(synth.unet_param_spec(512, base, ch_mult))
    g = torch.Generator("cpu").manual_seed(B * 100 + H)
    x_t = torch.randn((B, 3, H, W), generator=g); z = torch.from_numpy(synth.synth_z(B))
    t = torch.randint(0, 1000, (B,), generator=g); target = torch.randn((B, 3, H, W), generator=g)
    net = make_net(sd, base, ch_mult)
    loss, grads, eps = grads_via_autograd(net, x_t, z, t, target)
    rloss, rgrads, reps = ref_train.loss_and_grads(ref_unet.as_torch_sd(sd), x_t, z, t, target)
    assert abs(float(loss) - float(rloss)) < 1e-5 * max(1.0, float(rloss))
    check_grads(grads, rgrads, 3e-4, f"fp32 base={base} {ch_mult} {B}x{H}x{W}")


def test_gradients_bf16_mode_close_to_fp32_oracle():
    """bf16 activations / conv operands (what autocast does at train/diffusion_train.py:121): direction and size of every
    gradient tensor agree with the fp32 oracle."""
    sd = synth.synth_state_dict(synth.unet_param_spec(512, 64, (1, 2)))
    B, S = 2, 64
    g = torch.Generator("cpu").manual_seed(11)
    x_t = torch.randn((B, 3, S, S), generator=g); z = torch.from_numpy(synth.synth_z(B))
    t = torch.tensor([20, 700]); target = torch.randn((B, 3, S, S), generator=g)
    net = make_net(sd, 64, (1, 2), dtype="bf16")
    loss, grads, _ = grads_via_autograd(net, x_t, z, t, target)
    rloss, rgrads, _ = ref_train.loss_and_grads(ref_unet.as_torch_sd(sd), x_t, z, t, target)
    assert abs(float(loss) - float(rloss)) < 2e-2 * float(rloss)
    worst = 1.0
    for k, r in rgrads.items():
        a = grads[k].double().flatten(); b = r.double().flatten()
        cos = float((a @ b) / (a.norm() * b.norm() + 1e-30))
        worst = min(worst, cos)
        assert cos > 0.98, (k, cos)
        assert 0.9 < float(a.norm() / (b.norm() + 1e-30)) < 1.1, k
    print(f"bf16 gradients: worst cosine vs fp32 oracle {worst:.4f}")


def test_fused_loss_and_adamw_match_oracle_over_three_steps():
    """train_step (q_sample -> forward -> fused mse -> backward -> FusedAdamW) for three steps against the oracle loop."""
    sd = synth.synth_state_dict(synth.unet_param_spec(512, 32, (1, 2)))
    tables = ref_diffusion.scheduler_tables(1000, "cosine")
    x0, z, t, noise = (torch.from_numpy(GOLD[k]) for k in ("x0", "z", "t", "noise"))
    net = make_net(sd, 32, (1, 2))
    sch = NoiseScheduler(1000, "cosine", device=DEV)
    opt = FusedAdamW(net, lr=2e-4)
    ref = ref_unet.as_torch_sd(sd)
    m = {k: torch.zeros_like(v) for k, v in ref.items()}; v2 = {k: torch.zeros_like(v) for k, v in ref.items()}
    for step in range(1, 4):
        loss = train_step(net, sch, opt, x0.to(DEV), z.to(DEV), t.to(DEV), noise.to(DEV))
        rloss, rg, _, _ = ref_train.train_step_grads(ref, tables, x0, z, t, noise)
        assert abs(float(loss) - float(rloss)) < 5e-5 * float(rloss), (step, float(loss), float(rloss))
        for k in ref:
            ref[k], m[k], v2[k] = ref_train.adamw_update(ref[k], rg[k], m[k], v2[k], step)
    got = {k: p.detach().cpu() for k, p in net.named_parameters()}
    # Adam's first steps move every weight by ~lr regardless of the gradient's size, so compare the movement
    init = ref_unet.as_torch_sd(sd)
    for k in ref:
        moved = (ref[k] - init[k]).abs().max()
        assert float((got[k] - ref[k]).abs().max()) <= 0.05 * float(moved) + 1e-7, k
    # the state dict still has the reference's keys and the parameters are views of one flat buffer
    assert set(net.state_dict()) == set(sd)
    assert net.train_state().fp.intact()


def test_graph_replay_of_forward_and_backward_matches_plain_launches():
    """train_step(graph=True): captured hipGraphs of the forward and the backward (side-stream branch included) give the same
    parameters as plain launches after three steps."""
    sd = synth.synth_state_dict(synth.unet_param_spec(512, 32, (1, 2)))
    x0, z, t, noise = (torch.from_numpy(GOLD[k]).to(DEV) for k in ("x0", "z", "t", "noise"))
    sch = NoiseScheduler(1000, "cosine", device=DEV)
    res = []
    for graph in (False, True):
        net = make_net(sd, 32, (1, 2))
        opt = FusedAdamW(net, lr=2e-4)
        losses = [float(train_step(net, sch, opt, x0, z, t, noise, graph=graph)) for _ in range(3)]
        res.append((losses, torch.cat([p.detach().flatten().cpu() for p in net.parameters()])))
    assert np.allclose(res[0][0], res[1][0], rtol=1e-5)
    moved = float((res[0][1] - torch.cat([torch.from_numpy(v).flatten() for k, v in sd.items()])).abs().max())
    assert float((res[0][1] - res[1][1]).abs().max()) <= 0.05 * moved


def test_torch_optimizer_drop_in_and_eval_after_training():
    """The reference's three lines with torch.optim.AdamW, then .eval() inference sees the updated weights."""
    sd = synth.synth_state_dict(synth.unet_param_spec(512, 32, (1, 2)))
    net = make_net(sd, 32, (1, 2))
    opt = torch.optim.AdamW(net.parameters(), lr=2e-4)
    x0, z, t, noise = (torch.from_numpy(GOLD[k]).to(DEV) for k in ("x0", "z", "t", "noise"))
    sch = NoiseScheduler(1000, "cosine", device=DEV)
    losses = []
    for _ in range(3):
        x_t = sch.q_sample(x0, t, noise)
        loss = F.mse_loss(net(x_t, z, t), noise)
        loss.backward(); opt.step(); opt.zero_grad(set_to_none=True)
        losses.append(float(loss.detach()))
    assert losses[2] < losses[0]
    net.eval()
    with torch.no_grad():
        e = net(x0, z, t)
    ref = ref_unet.unet_forward(ref_unet.as_torch_sd({k: v.detach().cpu().numpy() for k, v in net.state_dict().items()}), x0.cpu(), z.cpu(), t.cpu())
    assert float((e.cpu() - ref).abs().max()) < 1e-4


def test_backward_requires_matching_forward():
    tr = _native.NativeTrainer(512, 32, (1, 2), 256, 3, device=DEV)
    flat = torch.zeros(tr.total, device=DEV); x = torch.zeros((1, 3, 32, 32), device=DEV); z = torch.zeros((1, 512), device=DEV)
    with pytest.raises(RuntimeError):
        tr.backward(flat, torch.zeros_like(flat), x, z, torch.zeros_like(x))
    with pytest.raises(ValueError):
        tr.forward(flat, torch.zeros((1, 3, 30, 32), device=DEV), z, torch.zeros(1, dtype=torch.int64, device=DEV))


def test_bf16_gradients_at_c2_widths_against_fp32_mode():
    """BASELINE configs[4]'s architecture (base 128, (1,2,2)) at 128 px, batch 2: every kernel choice of the bf16 step (transposed-read
    weight gradients on the warp-specialised kernel, 8-row forward / data-gradient tiles, pre-activated tensors) against the fp32 mode
    of the same library, whose gradients the other tests pin to the oracle."""
    sd = synth.synth_state_dict(synth.unet_param_spec(512, 128, (1, 2, 2)))
    B, S = 2, 128
    g = torch.Generator("cpu").manual_seed(21)
    x_t = torch.randn((B, 3, S, S), generator=g); z = torch.from_numpy(synth.synth_z(B))
    t = torch.tensor([450, 980]); target = torch.randn((B, 3, S, S), generator=g)
    l32, g32, _ = grads_via_autograd(make_net(sd, 128, (1, 2, 2)), x_t, z, t, target)
    l16, g16, _ = grads_via_autograd(make_net(sd, 128, (1, 2, 2), dtype="bf16"), x_t, z, t, target)
    assert abs(float(l16) - float(l32)) < 2e-2 * float(l32)
    worst = ("", 1.0)
    for k, r in g32.items():
        a = g16[k].double().flatten(); b = r.double().flatten()
        cos = float((a @ b) / (a.norm() * b.norm() + 1e-30))
        if cos < worst[1]:
            worst = (k, cos)
        assert cos > 0.97, (k, cos)
        assert 0.9 < float(a.norm() / (b.norm() + 1e-30)) < 1.1, k
    print(f"bf16 vs fp32 mode at base 128: worst cosine {worst[1]:.4f} ({worst[0]})")


def test_backward_accumulates_into_the_gradient_buffer():
    """ccn_train_backward adds to grads_dev (torch's .grad accumulation): two backward passes give twice the gradient."""
    sd = synth.synth_state_dict(synth.unet_param_spec(512, 32, (1, 2)))
    net = make_net(sd, 32, (1, 2))
    st = net.train_state()
    x0, z, t, noise = (torch.from_numpy(GOLD[k]).to(DEV) for k in ("x_t", "z", "t", "noise"))
    eps = st.trainer.forward(st.fp.flat, x0, z, t)
    _, d = _native.mse_loss_grad(eps, noise)
    g1 = torch.zeros_like(st.fp.flat); g2 = torch.zeros_like(st.fp.flat)
    st.trainer.backward(st.fp.flat, g1, x0, z, d)
    st.trainer.backward(st.fp.flat, g2, x0, z, d)
    st.trainer.backward(st.fp.flat, g2, x0, z, d)
    assert float(g1.abs().max()) > 0
    assert float((g2 - 2 * g1).abs().max()) <= 1e-5 * float(g1.abs().max())


def test_mse_loss_grad_and_adamw_kernels_against_torch():
    g = torch.Generator("cpu").manual_seed(5)
    eps = torch.randn((3, 3, 40, 24), generator=g); tgt = torch.randn((3, 3, 40, 24), generator=g)
    loss, d = _native.mse_loss_grad(eps.to(DEV), tgt.to(DEV))
    e = eps.clone().requires_grad_(True)
    ref = F.mse_loss(e, tgt); ref.backward()
    assert abs(float(loss) - float(ref.detach())) < 1e-6 and float((d.cpu() - e.grad).abs().max()) < 1e-9
    n = 10007
    p = torch.randn(n, generator=g); gr = torch.randn(n, generator=g) * 0.01
    pt = torch.nn.Parameter(p.clone()); opt = torch.optim.AdamW([pt], lr=3e-4, betas=(0.9, 0.99), eps=1e-8, weight_decay=0.05)
    pd = p.to(DEV).clone(); m = torch.zeros(n, device=DEV); v = torch.zeros(n, device=DEV)
    for step in range(1, 4):
        pt.grad = gr * step
        opt.step()
        _native.adamw_step(pd, (gr * step).to(DEV), m, v, 3e-4, 0.9, 0.99, 1e-8, 0.05, step)
    assert float((pd.cpu() - pt.detach()).abs().max()) < 2e-6
    # step + zero_grad in one pass (ccn_adamw_step_zero_grad): the same update, bit for bit, and the gradient buffer left at zero
    pa = p.to(DEV).clone(); pb = pa.clone(); ma, va, mb, vb = (torch.zeros(n, device=DEV) for _ in range(4))
    ga = (gr * 2).to(DEV); gb = ga.clone()
    _native.adamw_step(pa, ga, ma, va, 3e-4, 0.9, 0.99, 1e-8, 0.05, 1)
    _native.adamw_step(pb, gb, mb, vb, 3e-4, 0.9, 0.99, 1e-8, 0.05, 1, zero_grad=True)
    assert torch.equal(pa, pb) and torch.equal(ma, mb) and torch.equal(va, vb)
    assert torch.equal(ga, (gr * 2).to(DEV)) and not gb.any()


def test_train_diffusion_entry_point_on_a_synthetic_store(tmp_path):
    """The reference's train_diffusion(store_dir, ...) on a 12-record synthetic store (PNG + .clp + codec_meta), two epochs with
    the L1 and TV extras on: checkpoints carry the reference's key set and load strictly; the loss goes down."""
    from clip_feature_codec.io import bitstream
    from clip_feature_codec.train.diffusion_train import train_diffusion
    store = tmp_path / "store"
    synth.write_synth_store(store, 12, 32, write_clp=bitstream.write_bitstream)
    lines = []
    torch.manual_seed(0)
    final = train_diffusion(store, out_size=32, epochs=3, batch_size=4, lr=1e-3, device=DEV, save_dir=tmp_path / "ckpt", base=32, ch_mult=(1, 2),
                            dtype="fp32", num_workers=0, clip_w=0.1, log=lines.append)
    assert final.name == "diffusion_unet_final.pt" and (tmp_path / "ckpt" / "diffusion_unet_ep3.pt").exists()
    sd = torch.load(final, map_location="cpu", weights_only=True)
    assert set(sd) == {k for k, _ in synth.unet_param_spec(512, 32, (1, 2))}
    net = CLIPCondUNet(512, 32, (1, 2)).to(DEV)
    net.load_state_dict(sd, strict=True)
    losses = [float(ln.split("loss=")[1]) for ln in lines if "epoch" in ln]
    assert len(losses) == 3 and all(np.isfinite(losses)) and losses[-1] < losses[0], lines
    assert any("clip_w" in ln for ln in lines)


def test_bf16_step_on_the_persistent_kernel_with_ragged_tiles():
    """Batch 4 at 136 x 200 with base 128: enough 8-row tiles that the forward and data-gradient 3x3 convs of the first level run on
    the persistent register-weight kernel (fragment layout packed on the device), with partial tiles on every edge.  bf16 mode
    against the fp32 mode of the same library."""
    sd = synth.synth_state_dict(synth.unet_param_spec(512, 128, (1, 2)))
    B, H, W = 4, 136, 200
    g = torch.Generator("cpu").manual_seed(33)
    x_t = torch.randn((B, 3, H, W), generator=g); z = torch.from_numpy(synth.synth_z(B))
    t = torch.tensor([3, 333, 666, 999]); target = torch.randn((B, 3, H, W), generator=g)
    l32, g32, e32 = grads_via_autograd(make_net(sd, 128, (1, 2)), x_t, z, t, target)
    l16, g16, e16 = grads_via_autograd(make_net(sd, 128, (1, 2), dtype="bf16"), x_t, z, t, target)
    assert float((e16 - e32).abs().max()) < 2e-2
    worst = ("", 1.0)
    for k, r in g32.items():
        a = g16[k].double().flatten(); b = r.double().flatten()
        cos = float((a @ b) / (a.norm() * b.norm() + 1e-30))
        if cos < worst[1]:
            worst = (k, cos)
        assert cos > 0.97 and 0.9 < float(a.norm() / (b.norm() + 1e-30)) < 1.1, (k, cos)
    print(f"bf16 on the persistent kernel, ragged tiles: worst cosine {worst[1]:.4f} ({worst[0]})")


@pytest.mark.parametrize("base,ch_mult,B,H,W", [(128, (1, 2), 4, 136, 200), (128, (1, 2, 2), 4, 256, 256), (192, (1, 2, 2, 4), 2, 128, 128)])
def test_stride2_and_convtranspose_steps_on_the_persistent_kernel_match_the_generic_kernels(base, ch_mult, B, H, W):
    """Round 3: with >= 128 eight-row tiles the bf16 step runs its stride-2 convs (five plane passes), its ConvTransposes (four
    parities) and BOTH data gradients -- the stride-2 conv's (a ConvTranspose with the 3x3 kernel padded to 4x4) and the
    ConvTranspose's (a 4x4 stride-2 conv as four plane passes of 2x2 taps, the P4 form) -- on the persistent kernel, from fragment
    operands packed on the device (prs2 / prct / prp4_frag_index).  Same bf16 mode with the persistent kernel switched off (variant 3:
    generic implicit-GEMM / free-running kernels, the [tap][N][K] operands): eps and every gradient must agree to bf16 noise --
    a wrong tap, plane or fragment slot is an O(1) error.  Ragged tiles in the first case, the C5 per-GPU shape in the second, C4's
    architecture (192 / 384 / 768 / 3072 channels: half-padded N tiles, three and six 64-channel chunks) in the third."""
    import ctypes
    sd = synth.synth_state_dict(synth.unet_param_spec(512, base, ch_mult))
    g = torch.Generator("cpu").manual_seed(77)
    x_t = torch.randn((B, 3, H, W), generator=g); z = torch.from_numpy(synth.synth_z(B))
    t = torch.tensor([5, 250, 600, 990][:B]); target = torch.randn((B, 3, H, W), generator=g)
    lib = _native.load_library()
    lib.ccn_internal_set_conv_variant.restype = ctypes.c_int
    lib.ccn_internal_set_conv_variant.argtypes = [ctypes.c_int]
    old = lib.ccn_internal_set_conv_variant(3)
    try:
        l3, g3, e3 = grads_via_autograd(make_net(sd, base, ch_mult, dtype="bf16"), x_t, z, t, target)
        torch.cuda.empty_cache()
        lib.ccn_internal_set_conv_variant(4)
        l4, g4, e4 = grads_via_autograd(make_net(sd, base, ch_mult, dtype="bf16"), x_t, z, t, target)
    finally:
        lib.ccn_internal_set_conv_variant(old)
    assert float((e4 - e3).abs().max()) < 4e-2, float((e4 - e3).abs().max())
    assert float((e4 - e3).abs().mean()) < 2e-3, float((e4 - e3).abs().mean())
    worst = ("", 0.0)
    for k, r in g3.items():
        a = g4[k].double().flatten(); b = r.double().flatten()
        e = float((a - b).norm() / (b.norm() + 1e-30))
        if e > worst[1]:
            worst = (k, e)
        assert e < 0.08, (k, e)
    print(f"persistent vs generic kernels, bf16 step base {base} {ch_mult} {B}x{H}x{W}: eps max diff {float((e4 - e3).abs().max()):.2e}, "
          f"worst gradient relative L2 difference {worst[1]:.3e} ({worst[0]})")


def test_c4_architecture_gradients_fp32():
    """BASELINE configs[3]'s architecture (base 192, (1,2,2,4): 24/48/96/384 channels per group, 3072-channel bottleneck whose
    GroupNorm passes span several channel blocks and which the pre-pass kernel cannot take) at 32 px, batch 1: every gradient
    against the oracle."""
    spec = synth.unet_param_spec(512, 192, (1, 2, 2, 4))
    sd = synth.synth_state_dict(spec)
    g = torch.Generator("cpu").manual_seed(44)
    x_t = torch.randn((1, 3, 32, 32), generator=g); z = torch.from_numpy(synth.synth_z(1)); t = torch.tensor([512])
    target = torch.randn((1, 3, 32, 32), generator=g)
    net = make_net(sd, 192, (1, 2, 2, 4))
    loss, grads, eps = grads_via_autograd(net, x_t, z, t, target)
    del net
    torch.cuda.empty_cache()
    rloss, rgrads, reps = ref_train.loss_and_grads(ref_unet.as_torch_sd(sd), x_t, z, t, target)
    assert abs(float(loss) - float(rloss)) < 1e-5 * max(1.0, float(rloss))
    check_grads(grads, rgrads, 5e-4, "fp32 C4 architecture 1x32x32")


def test_bucketed_backward_hands_out_the_whole_buffer_and_the_same_gradients():
    """ccn_train_backward_bucketed: the gradient-ready ranges are disjoint, descend from the end of the flat buffer, cover it, and the
    gradients equal those of the plain backward (FiLM linears differentiated block by block instead of grouped)."""
    sd = synth.synth_state_dict(synth.unet_param_spec(512, 32, (1, 2)))
    net = make_net(sd, 32, (1, 2))
    st = net.train_state()
    x, z, t, noise = (torch.from_numpy(GOLD[k]).to(DEV) for k in ("x_t", "z", "t", "noise"))
    eps = st.trainer.forward(st.fp.flat, x, z, t)
    _, d = _native.mse_loss_grad(eps, noise)
    g_plain = torch.zeros_like(st.fp.flat); g_bucket = torch.zeros_like(st.fp.flat)
    st.trainer.backward(st.fp.flat, g_plain, x, z, d)
    ranges = []
    st.trainer.backward(st.fp.flat, g_bucket, x, z, d, bucket_cb=lambda lo, hi: ranges.append((lo, hi)), bucket_floats=200_000)
    assert len(ranges) >= 3 and ranges[0][1] == st.trainer.total and ranges[-1][0] == 0
    for (lo, hi), (lo2, hi2) in zip(ranges, ranges[1:]):
        assert lo < hi and hi2 == lo and lo2 < hi2
    assert all(hi - lo >= 200_000 for lo, hi in ranges[:-1])
    scale = float(g_plain.abs().max())
    assert float((g_plain - g_bucket).abs().max()) <= 2e-6 * scale
    with pytest.raises(ZeroDivisionError):                      # an exception inside the callback surfaces after the C call returns
        st.trainer.backward(st.fp.flat, torch.zeros_like(g_plain), x, z, d, bucket_cb=lambda lo, hi: 1 // 0)


def _ddp_rank(rank, world, port, out):
    import os
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.cuda.set_device(0)
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    sd = synth.synth_state_dict(synth.unet_param_spec(512, 32, (1, 2)))
    net = make_net(sd, 32, (1, 2))
    sch = NoiseScheduler(1000, "cosine", device=DEV)
    opt = FusedAdamW(net, lr=2e-4)
    g = torch.Generator("cpu").manual_seed(9)
    x0 = torch.rand((4, 3, 32, 32), generator=g) * 2 - 1; z = torch.from_numpy(synth.synth_z(4))
    t = torch.tensor([10, 400, 700, 990]); noise = torch.randn((4, 3, 32, 32), generator=g)
    lo, hi = rank * 4 // world, (rank + 1) * 4 // world
    for _ in range(2):
        train_step(net, sch, opt, x0[lo:hi].to(DEV), z[lo:hi].to(DEV), t[lo:hi].to(DEV), noise[lo:hi].to(DEV), ddp=world > 1)
    if rank == 0:
        np.save(out, torch.cat([p.detach().flatten().cpu() for p in net.parameters()]).numpy())
    if world > 1:
        dist.destroy_process_group()


def test_two_rank_training_with_overlapped_allreduce_equals_single_process(tmp_path):
    """Two ranks (gloo, both on this GPU) train on half batches with the bucketed backward + asynchronous all-reduce per bucket; after
    two steps their parameters equal those of one process training on the whole batch."""
    import torch.multiprocessing as mp
    _ddp_rank(0, 1, 0, str(tmp_path / "p1.npy"))
    mp.spawn(_ddp_rank, args=(2, 29571, str(tmp_path / "p2.npy")), nprocs=2, join=True)
    p1, p2 = np.load(tmp_path / "p1.npy"), np.load(tmp_path / "p2.npy")
    init = np.concatenate([v.reshape(-1) for v in synth.synth_state_dict(synth.unet_param_spec(512, 32, (1, 2))).values()])
    moved = np.abs(p1 - init).max()
    assert moved > 0 and np.abs(p1 - p2).max() <= 0.05 * moved


# ---- round-2 additions: advisor findings on the drop-in route ---------------------------------------------------------------
def test_eval_between_fused_adamw_steps_uses_the_current_weights():
    """FusedAdamW writes the parameters through the flat buffer (the views' own version counters do not move): an eval forward /
    preview sample between training steps must still see the updated weights -- each compared with the oracle on the CURRENT
    state_dict."""
    sd = synth.synth_state_dict(synth.unet_param_spec(512, 32, (1, 2)))
    net = make_net(sd, 32, (1, 2))
    sch = NoiseScheduler(1000, "cosine", device=DEV)
    opt = FusedAdamW(net, lr=5e-3)
    g = torch.Generator("cpu").manual_seed(21)
    x0 = torch.rand((2, 3, 32, 32), generator=g) * 2 - 1; z = torch.from_numpy(synth.synth_z(2))
    xe = torch.randn((2, 3, 32, 32), generator=g); te = torch.tensor([700, 20])

    def eval_err():
        net.eval()
        with torch.no_grad():
            got = net(xe.to(DEV), z.to(DEV), te.to(DEV)).cpu()
            ref = ref_unet.unet_forward({k: v.detach().cpu().clone() for k, v in net.state_dict().items()}, xe, z, te)
        net.train()
        return got, float((got - ref).abs().max())

    e0, err0 = eval_err()
    assert err0 < 2e-5, err0
    train_step(net, sch, opt, x0.to(DEV), z.to(DEV))
    e1, err1 = eval_err()                                      # handle built BEFORE this step must not be reused
    assert err1 < 2e-5, err1
    assert float((e1 - e0).abs().max()) > 1e-4                 # the step really moved the output
    for _ in range(2):
        train_step(net, sch, opt, x0.to(DEV), z.to(DEV))
    e2, err2 = eval_err()
    assert err2 < 2e-5 and float((e2 - e1).abs().max()) > 1e-4, (err2,)


def test_two_training_forwards_before_a_backward_raise():
    """The trainer holds one forward's activations: a backward of an older forward must raise, not differentiate the newer one."""
    sd = synth.synth_state_dict(synth.unet_param_spec(512, 32, (1, 2)))
    net = make_net(sd, 32, (1, 2))
    g = torch.Generator("cpu").manual_seed(22)
    xa, xb = (torch.randn((1, 3, 32, 32), generator=g).to(DEV) for _ in range(2))
    z = torch.from_numpy(synth.synth_z(1)).to(DEV); t = torch.tensor([300], device=DEV)
    la = net(xa, z, t).square().mean()
    lb = net(xb, z, t).square().mean()
    with pytest.raises(RuntimeError, match="another training forward"):
        (la + lb).backward()
    # forward -> backward pairs (micro-batches) work and accumulate in the flat gradient buffer
    net.zero_grad(set_to_none=True)
    net(xa, z, t).square().mean().backward()
    ga = net.train_state().fp.grad.clone()
    net(xb, z, t).square().mean().backward()
    gab = net.train_state().fp.grad.clone()
    net.zero_grad(set_to_none=True)
    net(xb, z, t).square().mean().backward()
    gb = net.train_state().fp.grad.clone()
    scale = float(gab.abs().max())
    assert float((gab - (ga + gb)).abs().max()) <= 1e-5 * scale
    p = next(net.parameters())
    assert p.grad.data_ptr() == net.train_state().fp.grad.data_ptr() + 4 * net.train_state().fp.views[0][1]


def _ddp_grad_rank(rank, world, port, out):
    import os
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.cuda.set_device(0)
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    sd = synth.synth_state_dict(synth.unet_param_spec(512, 32, (1, 2)))
    net = make_net(sd, 32, (1, 2))
    st = net.train_state()
    x, z, t, noise = (torch.from_numpy(GOLD[k]) for k in ("x_t", "z", "t", "noise"))
    B = x.shape[0]
    lo, hi = rank * B // world, (rank + 1) * B // world
    x, z, t, noise = (v[lo:hi].to(DEV) for v in (x, z, t, noise))
    eps = st.trainer.forward(st.fp.flat, x, z, t)
    _, d = _native.mse_loss_grad(eps, noise)
    d.mul_(1.0 / world)
    st.fp.grad.zero_()
    works = []
    st.trainer.backward(st.fp.flat, st.fp.grad, x, z, d, bucket_floats=200_000,
                        bucket_cb=(lambda a, b: works.append(dist.all_reduce(st.fp.grad[a:b], async_op=True))) if world > 1 else None)
    for w in works:
        w.wait()
    torch.cuda.synchronize()
    if rank == 0:
        np.save(out, st.fp.grad.cpu().numpy())
    if world > 1:
        dist.destroy_process_group()


def test_two_rank_bucketed_allreduce_gives_the_full_batch_gradient(tmp_path):
    """One bucketed backward per rank on half of the fixture's batch, every bucket all-reduced as it is handed out: the reduced flat
    gradient equals the single-process full-batch gradient to 1e-5 of its max (catches a wrong 1/world scale or a missed bucket,
    which parameter positions after Adam steps cannot).  gloo on one card: the RCCL stream-ordering of the callback is exercised
    only on a multi-GPU node (bench_train.py --gpus N)."""
    import torch.multiprocessing as mp
    _ddp_grad_rank(0, 1, 0, str(tmp_path / "g1.npy"))
    mp.spawn(_ddp_grad_rank, args=(2, 29573, str(tmp_path / "g2.npy")), nprocs=2, join=True)
    g1, g2 = np.load(tmp_path / "g1.npy"), np.load(tmp_path / "g2.npy")
    assert np.abs(g1).max() > 0
    assert np.abs(g1 - g2).max() <= 1e-5 * np.abs(g1).max(), np.abs(g1 - g2).max() / np.abs(g1).max()


def _ddp_accum_rank(rank, world, port, out, foreign):
    """Autograd route (UNetFunction.backward) under data parallelism with two micro-batches per rank: the first inside no_sync()."""
    import os
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.cuda.set_device(0)
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    sd = synth.synth_state_dict(synth.unet_param_spec(512, 32, (1, 2)))
    net = make_net(sd, 32, (1, 2)).train()
    st = net.train_state()
    st.ddp_bucketed = world > 1
    gen = torch.Generator("cpu").manual_seed(77)
    B, S = 4, 32
    x = torch.randn((B, 3, S, S), generator=gen); noise = torch.randn((B, 3, S, S), generator=gen)
    z = torch.from_numpy(synth.synth_z(B)); t = torch.tensor([5, 250, 600, 990])
    per = B // world
    mine = [v[rank * per:(rank + 1) * per].to(DEV) for v in (x, z, t, noise)]
    half = per // 2
    if foreign:                                    # .grad tensors that are NOT views of the flat buffer: the fallback route
        for p in net.parameters():
            p.grad = torch.zeros_like(p)
    def mb(lo, hi):
        xx, zz, tt, nn_ = (v[lo:hi] for v in mine)
        # mean over the GLOBAL batch: every micro-batch contributes sum / B (the 1/world factor is the library's)
        return ((net(xx, zz, tt) - nn_) ** 2).sum() * (world / (B * 3 * xx.shape[2] * xx.shape[3]))
    with st.no_sync():
        mb(0, half).backward()
    mb(half, per).backward()
    st.wait_grad_sync()
    torch.cuda.synchronize()
    g = torch.cat([p.grad.flatten() for p in net.parameters()])
    if rank == 0:
        np.save(out, g.cpu().numpy())
    if world > 1:
        dist.destroy_process_group()


@pytest.mark.parametrize("foreign", [False, True])
def test_two_rank_micro_batches_with_no_sync_give_the_full_batch_gradient(tmp_path, foreign):
    """Gradient accumulation on the drop-in autograd route under data parallelism: two micro-batches per rank, the first inside
    TrainState.no_sync(), on two gloo ranks (one card) against four micro-batches of one process -- the mean-over-the-global-batch
    gradient must agree to 1e-5 of its max.  Without no_sync the first micro-batch would be reduced twice (x world); with foreign
    .grad tensors (not views of the flat buffer) the gradients go through autograd and are averaged by the fallback all-reduce."""
    import torch.multiprocessing as mp
    _ddp_accum_rank(0, 1, 0, str(tmp_path / "g1.npy"), foreign)
    mp.spawn(_ddp_accum_rank, args=(2, 29577 + int(foreign), str(tmp_path / "g2.npy"), foreign), nprocs=2, join=True)
    g1, g2 = np.load(tmp_path / "g1.npy"), np.load(tmp_path / "g2.npy")
    assert np.abs(g1).max() > 0
    assert np.abs(g1 - g2).max() <= 1e-5 * np.abs(g1).max(), np.abs(g1 - g2).max() / np.abs(g1).max()


def test_c5_per_gpu_shape_bf16_vs_fp32_mode():
    """BASELINE configs[4]'s per-GPU shape (256 px, batch 4, base 128, (1,2,2)): one loss + backward in bf16 mode against the fp32
    parity mode of the same library (itself checked against the oracle at the sizes above) -- loss, eps and every gradient
    tensor's direction and norm; reference loop body: train/diffusion_train.py:119-124,137."""
    sd = synth.synth_state_dict(synth.unet_param_spec(512, 128, (1, 2, 2)))
    B, S = 4, 256
    g = torch.Generator("cpu").manual_seed(55)
    x_t = torch.randn((B, 3, S, S), generator=g); z = torch.from_numpy(synth.synth_z(B))
    t = torch.tensor([7, 321, 654, 987]); target = torch.randn((B, 3, S, S), generator=g)
    l32, g32, e32 = grads_via_autograd(make_net(sd, 128, (1, 2, 2)), x_t, z, t, target)
    torch.cuda.empty_cache()
    l16, g16, e16 = grads_via_autograd(make_net(sd, 128, (1, 2, 2), dtype="bf16"), x_t, z, t, target)
    assert abs(float(l16) - float(l32)) < 2e-3 * float(l32), (float(l16), float(l32))
    assert float((e16 - e32).abs().max()) < 2e-2
    worst = ("", 1.0)
    for k, r in g32.items():
        a = g16[k].double().flatten(); b = r.double().flatten()
        cos = float((a @ b) / (a.norm() * b.norm() + 1e-30))
        if cos < worst[1]:
            worst = (k, cos)
        assert cos > 0.97 and 0.9 < float(a.norm() / (b.norm() + 1e-30)) < 1.1, (k, cos)
    print(f"C5 shape 4x256x256: loss fp32 {float(l32):.5f} bf16 {float(l16):.5f}; worst gradient cosine {worst[1]:.4f} ({worst[0]})")
