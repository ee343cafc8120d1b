"""N>1 path on CPU: two gloo ranks shard a synthetic store, reconstruct with an injected stand-in,
all-gather once, and must reproduce the single-process rows exactly."""
import json
import os
from pathlib import Path

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from clip_feature_codec.cli import eval as cli_eval
from clip_feature_codec.cli._common import start_noise, load_codec_meta, load_embedding
from clip_feature_codec.io import bitstream
from clip_feature_codec.utils import synth

SIZE, N = 16, 11   # ragged: 11 records over 2 ranks, batch 4


def _recon_fn():
    """Deterministic stand-in for the GPU sampler (tests only): the oracle on a tiny model."""
    from oracle import ref_unet, ref_diffusion
    sd = ref_unet.as_torch_sd(synth.synth_state_dict(synth.unet_param_spec(512, 8, (1,))))
    model = ref_unet.make_model(sd)

    def recon(z, x_T):
        x = ref_diffusion.ddim_sample(model, torch.from_numpy(z), x_T, steps=3)
        return x.clamp(-1, 1).numpy()
    return recon


def _run(rank, world, store, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(1)
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    manifest = json.loads((Path(store) / "manifest.json").read_text())
    scale, zero = load_codec_meta(Path(store))
    rows = cli_eval.evaluate(manifest, lambda r: load_embedding(Path(r["bitstream"]), scale, zero), _recon_fn(),
                             SIZE, 4, 5, rank, world, "cpu", start_noise)
    if rank == 0:
        np.save(out, rows)
    if world > 1:
        dist.destroy_process_group()


def test_two_rank_eval_matches_single_process(tmp_path):
    store = tmp_path / "store"
    synth.write_synth_store(store, N, SIZE, write_clp=bitstream.write_bitstream)
    _run(0, 1, str(store), 0, str(tmp_path / "one.npy"))
    mp.spawn(_run, args=(2, str(store), 29541, str(tmp_path / "two.npy")), nprocs=2, join=True)
    one, two = np.load(tmp_path / "one.npy"), np.load(tmp_path / "two.npy")
    assert one.shape == (N, 4) and np.isfinite(one[:, 0]).all()
    assert np.array_equal(one, two, equal_nan=True)


# ---- data-parallel training step: per-rank gradients of half batches, one all-reduce --------------
def _train_rank(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(1)
    from oracle import ref_unet, ref_train
    from clip_feature_codec.train.diffusion_train import average_gradients
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    spec = synth.unet_param_spec(512, 8, (1,))
    sd = ref_unet.as_torch_sd(synth.synth_state_dict(spec))
    g = torch.Generator("cpu").manual_seed(3)
    B = 4
    x = torch.randn((B, 3, 16, 16), generator=g); z = torch.from_numpy(synth.synth_z(B)); t = torch.tensor([5, 300, 650, 999])
    target = torch.randn((B, 3, 16, 16), generator=g)
    lo, hi = (rank * B // world, (rank + 1) * B // world)
    _, grads, _ = ref_train.loss_and_grads(sd, x[lo:hi], z[lo:hi], t[lo:hi], target[lo:hi])
    flat = torch.cat([grads[k].flatten() for k, _ in spec])           # the flat gradient buffer of this rank
    average_gradients(flat)
    if rank == 0:
        np.save(out, flat.numpy())
    if world > 1:
        dist.destroy_process_group()


def test_two_rank_gradient_average_equals_full_batch_gradient(tmp_path):
    _train_rank(0, 1, 0, str(tmp_path / "g1.npy"))
    mp.spawn(_train_rank, args=(2, 29547, str(tmp_path / "g2.npy")), nprocs=2, join=True)
    g1, g2 = np.load(tmp_path / "g1.npy"), np.load(tmp_path / "g2.npy")
    assert g1.shape == g2.shape and np.abs(g1).max() > 0
    assert np.abs(g1 - g2).max() <= 1e-5 * np.abs(g1).max()
