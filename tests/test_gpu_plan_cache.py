"""Plan / workspace caches under churn: the Python side keeps at most 8 workspaces per handle and frees the least
recently used one to torch's allocator; the library caches plans keyed by (B, H, W, steps, workspace address) and a plan keeps state
in its workspace (uploaded timestep table, zeroed split-K flags and arrival counters).  A later workspace of the same shape may land
on the recycled address: ccn_release_workspace drops the stale plans first.  Cycling more shapes than either cache holds, with other
allocations in between, must reproduce the first pass bit for bit (the kernels are run-to-run deterministic)."""
import numpy as np
import pytest
import torch

from clip_feature_codec import _native
from clip_feature_codec.models.unet import CLIPCondUNet
from clip_feature_codec.diffusion.scheduler import NoiseScheduler
from clip_feature_codec.diffusion.ddim import DDIMSampler

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_cycling_more_shapes_than_the_caches_hold_reproduces_first_pass(synth):
    base, cm = 128, (1, 2, 2)                         # bf16 mode at this width uses the persistent kernel incl. its split-K hand-off
    sd = synth.synth_state_dict(synth.unet_param_spec(512, base, cm))
    net = CLIPCondUNet(512, base, cm, dtype="bf16").to(DEV).eval()
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    sm = DDIMSampler(NoiseScheduler(1000, "cosine", DEV), 0.0)
    shapes = [(1, 32, 3), (2, 32, 3), (1, 64, 2), (2, 64, 3), (3, 32, 2), (1, 96, 2), (2, 96, 2), (4, 32, 3), (1, 128, 2), (3, 64, 2),
              (2, 128, 2), (1, 32, 4)]            # 12 keys > 8 workspaces / plans
    assert len(shapes) > _native.NativeUNet.MAX_WORKSPACES

    def inputs(i, B, S):
        z = torch.from_numpy(synth.synth_z(B, seed=50 + i)).to(DEV)
        xT = torch.from_numpy(synth.start_noise(range(B), S, 200 + i)).to(DEV)
        return z, xT

    def one_pass(junk_scale):
        outs = []
        for i, (B, S, steps) in enumerate(shapes):
            z, xT = inputs(i, B, S)
            outs.append(sm.sample(net, z, (B, 3, S, S), steps=steps, x_T=xT).cpu().numpy())
            # other traffic through torch's allocator: blocks of workspace-like sizes are taken and returned, filled with garbage
            junk = [torch.full(((junk_scale + k) * (1 << 20),), float("nan"), device=DEV) for k in range(3)]
            del junk
        return outs

    first = one_pass(3)
    torch.cuda.empty_cache()                          # returns the cached blocks: the next workspaces come from fresh segments
    second = one_pass(5)
    third = one_pass(2)
    net.native().poll_errors()
    for i, (a, b, c) in enumerate(zip(first, second, third)):
        assert np.isfinite(a).all()
        assert np.array_equal(a, b), (shapes[i], float(np.abs(a - b).max()))
        assert np.array_equal(a, c), (shapes[i], float(np.abs(a - c).max()))


def test_release_workspace_drops_the_plans_living_in_it(synth, tiny_sd):
    net = CLIPCondUNet(512, 32, (1, 2), dtype="fp32").to(DEV).eval()
    net.load_state_dict({k: torch.from_numpy(v) for k, v in tiny_sd.items()}, strict=True)
    nat = net.native()
    B, S = 2, 32
    x = torch.from_numpy(synth.start_noise(range(B), S, 9)).to(DEV)
    z = torch.from_numpy(synth.synth_z(B)).to(DEV)
    t = torch.full((B,), 500, dtype=torch.int64, device=DEV)
    e0 = nat.forward(x, z, t).cpu()
    ws = nat.workspace(B, S, S, 1)
    # garbage in the whole workspace, then a forward through the SAME address: only a plan rebuilt from scratch (zeroed counters and
    # flags) gives the same result; the stale plan would run on poisoned hand-off state
    nat._release(ws)
    torch.cuda.synchronize()
    ws.buf.fill_(0xFF)
    e1 = nat.forward(x, z, t).cpu()
    assert torch.equal(e0, e1)
    assert nat.lib.ccn_release_workspace(nat.h, None) == 0            # unknown address: nothing to drop, not an error
