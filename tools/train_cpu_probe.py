#!/usr/bin/env python3
"""Is the training step bound by the host?  Per step: the time the host needs to ENQUEUE it (no synchronisation) next to the
wall time per step of the whole loop; and the split of the enqueue time over the step's parts.

    python tools/train_cpu_probe.py [--steps 30]"""
import argparse, sys, time
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "clip-neural-image-conpression_amd")]
from clip_feature_codec.models.unet import CLIPCondUNet  # noqa: E402
from clip_feature_codec.diffusion.scheduler import NoiseScheduler  # noqa: E402
from clip_feature_codec.train.diffusion_train import FusedAdamW, train_step  # noqa: E402
from clip_feature_codec import _native  # noqa: E402
from clip_feature_codec.utils import synth  # noqa: E402

ap = argparse.ArgumentParser(); ap.add_argument("--steps", type=int, default=30); a = ap.parse_args()
dev = "cuda:0"; B, S = 4, 256
sd = synth.synth_state_dict(synth.unet_param_spec(512, 128, (1, 2, 2)))
net = CLIPCondUNet(512, 128, (1, 2, 2), dtype="bf16").to(dev)
net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True); net.train()
sch = NoiseScheduler(1000, "cosine", device=dev); opt = FusedAdamW(net, lr=2e-4)
g = torch.Generator("cpu").manual_seed(0)
x0 = (torch.rand((B, 3, S, S), generator=g) * 2 - 1).to(dev); z = torch.from_numpy(synth.synth_z(B)).to(dev)
for _ in range(5): train_step(net, sch, opt, x0, z)
torch.cuda.synchronize()
enq = []
t0 = time.perf_counter()
for _ in range(a.steps):
    t = time.perf_counter(); train_step(net, sch, opt, x0, z); enq.append(time.perf_counter() - t)
t_enq = time.perf_counter() - t0
torch.cuda.synchronize()
wall = time.perf_counter() - t0
print(f"wall {wall / a.steps * 1e3:.2f} ms/step; host enqueue {t_enq / a.steps * 1e3:.2f} ms/step (min {min(enq) * 1e3:.2f}, median {sorted(enq)[len(enq) // 2] * 1e3:.2f})")
# split of the host time with a synchronisation after every part (so nothing blocks on a full queue)
state = net.train_state(); fp = state.fp; sb = state.static_buffers(x0, z)
parts = {"prep": 0.0, "forward": 0.0, "loss": 0.0, "backward": 0.0, "opt": 0.0}
for _ in range(10):
    torch.cuda.synchronize(); t = time.perf_counter()
    fp.rebind_grads(); sb["x0"].copy_(x0); sb["z"].copy_(z); sb["t"].random_(0, sch.timesteps); sb["noise"].normal_()
    sch.q_sample(sb["x0"], sb["t"], sb["noise"], out=sb["x_t"]); parts["prep"] += time.perf_counter() - t
    torch.cuda.synchronize(); t = time.perf_counter()
    state.trainer.forward(fp.flat, sb["x_t"], sb["z"], sb["t"], out=sb["eps"]); parts["forward"] += time.perf_counter() - t
    torch.cuda.synchronize(); t = time.perf_counter()
    loss, d_eps = _native.mse_loss_grad(sb["eps"], sb["noise"], bufs=(sb["loss"], sb["d_eps"], sb["scratch"])); parts["loss"] += time.perf_counter() - t
    torch.cuda.synchronize(); t = time.perf_counter()
    state.trainer.backward(fp.flat, fp.grad, sb["x_t"], sb["z"], d_eps); parts["backward"] += time.perf_counter() - t
    torch.cuda.synchronize(); t = time.perf_counter()
    opt.step(); opt.zero_grad(); parts["opt"] += time.perf_counter() - t
print("host enqueue time per part (ms): " + ", ".join(f"{k} {v / 10 * 1e3:.3f}" for k, v in parts.items()))
