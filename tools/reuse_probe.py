import sys
from pathlib import Path
import numpy as np, torch
REPO = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(REPO), str(REPO / "clip-neural-image-conpression_amd")]
from clip_feature_codec.utils import synth
from clip_feature_codec.models.unet import CLIPCondUNet
dev = "cuda:0"
sd = synth.synth_state_dict(synth.unet_param_spec(512, 128, (1, 2, 2)))
def mk(dt):
    n = CLIPCondUNet(512, 128, (1, 2, 2), dtype=dt).to(dev).eval(); n.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}); return n
n16, n32 = mk("bf16"), mk("fp32")
for (B, H, W) in [(8, 256, 256), (8, 200, 168), (8, 200, 160), (8, 208, 168), (8, 256, 168), (5, 136, 200), (8, 264, 256)]:
    g = torch.Generator("cpu").manual_seed(3)
    x = torch.randn((B, 3, H, W), generator=g).to(dev); z = torch.from_numpy(synth.synth_z(B)).to(dev)
    t = torch.randint(0, 1000, (B,), generator=g).to(dev)
    e16, e32 = n16(x, z, t), n32(x, z, t)
    d = (e16 - e32).abs()
    print(B, H, W, "max err", float(d.max()), "per-sample", [round(float(v), 4) for v in d.flatten(1).max(1).values])
    for name in ("in_conv", "down.0", "down.1", "down.2", "down.3", "down.4", "up.6", "up.7", "up.8"):
        sh = (B, 128, H, W) if name in ("in_conv", "down.0", "down.1", "up.8") else (B, 128, H // 2, W // 2)
        a16 = n16.read_activation(name, sh); a32 = n32.read_activation(name, sh)
        dd = (a16 - a32).abs()
        w = torch.nonzero(dd == dd.max())[0].tolist()
        print("   ", name, "max err %.4f at %s" % (float(dd.max()), w))
