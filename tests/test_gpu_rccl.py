"""RCCL itself, executed on the one GPU a single-GPU box has: a one-rank process group with ``backend="nccl"`` in a fresh child
process.  What the N > 1 route does with the backend -- the final all-gather of cli.eval's metric rows from GPU-resident tensors
(cli/eval.py:56-86 of the reference is the loop that is sharded), bench.py's barrier / all-gather / all-reduce, the training step's
bucketed ``dist.all_reduce(..., async_op=True)`` on RCCL's stream joined before the optimiser step -- runs for real; results must
equal the run without a process group.  What stays unverified here is N > 1 (xGMI links, several devices): the driver's scaling run.
"""
import json
import os
import subprocess
import sys
import textwrap
from pathlib import Path

import numpy as np
import pytest
import torch

from clip_feature_codec.io import bitstream
from clip_feature_codec.utils import synth

pytestmark = pytest.mark.gpu
REPO = Path(__file__).resolve().parent.parent
PKG = REPO / "clip-neural-image-conpression_amd"


def _env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "CCN_LAUNCH_DEPTH",
                                                            "CCN_DIST_BACKEND")}
    env["PYTHONPATH"] = os.pathsep.join([str(PKG), str(REPO)] + ([env["PYTHONPATH"]] if env.get("PYTHONPATH") else []))
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    env.update(kw)
    return env


def _json_line(out: str) -> dict:
    return json.loads([ln for ln in out.splitlines() if ln.startswith("{")][-1])


def test_cli_eval_one_rank_rccl_group_equals_no_group(tmp_path):
    """cli.eval with --force-process-group: gather_metric_rows all-gathers a GPU-resident block over RCCL (world 1); every row equals
    the run without a group."""
    n, size, steps, batch = 12, 64, 5, 8
    store = tmp_path / "store"
    synth.write_synth_store(store, n, size, write_clp=bitstream.write_bitstream)
    sd = synth.synth_state_dict(synth.unet_param_spec(512, 128, (1, 2, 2)))
    ckpt = tmp_path / "ckpt.pt"
    torch.save({k: torch.from_numpy(v) for k, v in sd.items()}, ckpt)

    def run(out, extra):
        cmd = [sys.executable, "-m", "clip_feature_codec.cli.eval", "--store_dir", str(store), "--weights", str(ckpt), "--size", str(size),
               "--steps", str(steps), "--batch", str(batch), "--seed", "7", "--device", "cuda", "--dtype", "bf16", "--out_json", str(out)] + extra
        r = subprocess.run(cmd, capture_output=True, text=True, env=_env(), timeout=900)
        assert r.returncode == 0, r.stderr[-3000:]
        return json.loads(Path(out).read_text()), r.stderr
    plain, _ = run(tmp_path / "a.json", [])
    grouped, err = run(tmp_path / "b.json", ["--force-process-group"])
    assert "[eval] process group: backend nccl, world 1, collective on cuda" in err
    assert len(plain) == len(grouped) == n
    for a, b in zip(plain, grouped):
        assert a["image"] == b["image"] and a["psnr"] == b["psnr"] and a["ssim"] == b["ssim"]


def test_bench_one_rank_rccl_group():
    """bench.py --force-process-group: init_process_group('nccl'), the all-reduce of ones, both barriers of the timed region and the
    all-gather of the per-rank times run on RCCL; the line says so."""
    r = subprocess.run([sys.executable, str(REPO / "bench.py"), "--gpus", "1", "--force-process-group", "--steps", "2", "--warmup", "1", "--size", "64",
                        "--ddim-steps", "4", "--no-cpu-baseline", "--no-roofline", "--no-parity"], capture_output=True, text=True, env=_env(), timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = _json_line(r.stdout)
    assert line["n_gpus"] == 1 and line["rccl_ranks"] == 1 and line["backend"] == "nccl" and line["value"] > 0


TRAIN_CHILD = textwrap.dedent("""
    import os, sys, json
    import numpy as np, torch
    import torch.distributed as dist
    from clip_feature_codec.utils import synth
    from clip_feature_codec.utils.launch import single_rank_env, init_process_group
    from clip_feature_codec.models.unet import CLIPCondUNet
    from clip_feature_codec.diffusion.scheduler import NoiseScheduler
    from clip_feature_codec.train.diffusion_train import FusedAdamW, train_step
    grouped = sys.argv[1] == "1"
    dev = "cuda:0"; torch.cuda.set_device(dev)
    if grouped:
        single_rank_env(); init_process_group(dev)
        assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
    base, cm, B, S = 32, (1, 2), 4, 32
    sd = synth.synth_state_dict(synth.unet_param_spec(512, base, cm))
    net = CLIPCondUNet(512, base, cm, dtype="fp32").to(dev).train()
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    sch = NoiseScheduler(1000, "cosine", dev)
    opt = FusedAdamW(net, lr=2e-4)
    g = torch.Generator("cpu").manual_seed(3)
    x0 = torch.rand(B, 3, S, S, generator=g).mul(2).sub(1).to(dev); z = torch.from_numpy(synth.synth_z(B)).to(dev)
    losses = []
    for i in range(3):
        t = torch.randint(0, 1000, (B,), generator=g).to(dev); noise = torch.randn(B, 3, S, S, generator=g).to(dev)
        losses.append(float(train_step(net, sch, opt, x0, z, t=t, noise=noise, ddp="always" if grouped else False)))
    torch.cuda.synchronize()
    flat = net.train_state().fp.flat.detach().cpu().numpy()
    np.save(sys.argv[2], flat)
    print(json.dumps({"losses": losses}))
    if grouped:
        dist.barrier(); dist.destroy_process_group()
""")


def test_train_step_bucketed_allreduce_on_rccl_one_rank(tmp_path):
    """train_step(ddp="always") in a one-rank nccl group: ccn_train_backward_bucketed hands finished gradient ranges to
    dist.all_reduce(async_op=True) -- RCCL's own stream -- and wait() joins them before AdamW.  Three steps must leave the same
    parameters as three plain steps (the all-reduce over one rank is the identity; a missing join or a clobbered bucket is not)."""
    outs = []
    for grouped in ("0", "1"):
        out = tmp_path / f"flat{grouped}.npy"
        r = subprocess.run([sys.executable, "-c", TRAIN_CHILD, grouped, str(out)], capture_output=True, text=True, env=_env(), timeout=900)
        assert r.returncode == 0, r.stderr[-3000:]
        outs.append((np.load(out), _json_line(r.stdout)))
    (f0, l0), (f1, l1) = outs
    assert np.allclose(l0["losses"], l1["losses"], rtol=1e-5, atol=1e-7), (l0, l1)
    # the bucketed backward differentiates the FiLM linears block by block (other summation order of a few fp32 sums)
    assert np.abs(f0 - f1).max() <= 2e-6 * max(1.0, np.abs(f0).max()), np.abs(f0 - f1).max()
