"""``python -m clip_feature_codec.cli.eval`` -- store evaluation, batched and sharded over the GPUs of a node.

Reference surface kept (cli/eval.py:33-86): flags ``--store_dir --weights --size --steps --eta --device
--out_json``; the four ``Average ...`` lines; the JSON list of ``{image, psnr, ssim, lpips, clip_sim}``
in manifest order; NaN-filtered means.  What differs, because the reference reconstructs one image
at a time on one device:

* records are processed ``--batch`` (8) at a time through the fused DDIM graph;
* under ``torchrun`` (one process per GPU) rank r takes manifest records ``r::world``; there is no
  communication inside the loop, and ONE all-gather (RCCL over xGMI, ``backend='nccl'``) of the
  per-record metric rows at the end; rank 0 prints / writes;
* ``--seed`` makes start noise reproducible and independent of the sharding (record i always gets
  the CPU-generator stream ``seed+i``); without it noise is unseeded like the reference;
* originals are decoded / resized on host threads while the GPU samples.
"""
from __future__ import annotations

import argparse
import json
import os
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path
from typing import Callable, List, Optional, Sequence

import numpy as np
import torch
from PIL import Image

from ..eval.metrics import psnr, ssim, lpips_distance, clip_similarity

METRIC_KEYS = ("psnr", "ssim", "lpips", "clip_sim")


def shard_indices(n: int, rank: int, world: int) -> List[int]:
    """Records of rank ``rank``: the strided slice rank::world (independent units, no exchange)."""
    return list(range(rank, n, world))


def load_original(path: str, size: int) -> np.ndarray:
    img = Image.open(path).convert("RGB").resize((size, size), Image.BICUBIC)
    return (np.array(img).astype(np.float32) / 127.5 - 1.0).transpose(2, 0, 1)


def metric_row(orig: np.ndarray, recon: np.ndarray, device: str) -> List[float]:
    return [psnr(orig, recon), ssim(orig, recon), lpips_distance(orig, recon, device=device),
            clip_similarity(orig, recon, device=device)]


def gather_metric_rows(local_idx: Sequence[int], local_rows: np.ndarray, n_total: int, device: str) -> np.ndarray:
    """One all-gather of [n_pad, 1+4] fp32 blocks (index, 4 metrics); returns (n_total, 4) in manifest order."""
    import torch.distributed as dist
    world = dist.get_world_size() if dist.is_initialized() else 1
    out = np.full((n_total, len(METRIC_KEYS)), np.nan, dtype=np.float64)
    if not dist.is_initialized():                                  # (a one-rank GROUP still goes through the collective below)
        for i, row in zip(local_idx, local_rows):
            out[i] = row
        return out
    n_pad = (n_total + world - 1) // world
    block = torch.full((n_pad, 1 + len(METRIC_KEYS)), float("nan"), dtype=torch.float64)
    block[:, 0] = -1
    for k, (i, row) in enumerate(zip(local_idx, local_rows)):
        block[k, 0] = i
        block[k, 1:] = torch.as_tensor(row, dtype=torch.float64)
    from ..utils.launch import collective_device
    block = block.to(collective_device(device))
    blocks = [torch.empty_like(block) for _ in range(world)]
    dist.all_gather(blocks, block)
    for blk in blocks:
        for row in blk.cpu().numpy():
            if row[0] >= 0:
                out[int(row[0])] = row[1:]
    return out


def aggregate(rows: np.ndarray) -> dict:
    """NaN-filtered mean per metric (cli/eval.py:77-79)."""
    res = {}
    for k, key in enumerate(METRIC_KEYS):
        vals = rows[:, k][~np.isnan(rows[:, k])]
        res[key] = float(np.mean(vals)) if vals.size else float("nan")
    return res


def default_workers(world: int) -> int:
    """Host threads of one rank: this process's CPU share divided over the ranks of the node, 2..16."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 2)
    return max(2, min(16, n // max(world, 1)))


def original_u8(path: str, size: int) -> np.ndarray:
    """The uint8 image the reference's PSNR / SSIM see for an original: its float form ([-1, 1], cli/eval.py:66-67) pushed
    through ``_to_uint8`` again (eval/metrics.py:16-19) -- NOT the decoded pixels: v / 127.5 - 1 + 1 does not round-trip in
    fp32 and the truncation then loses a level on some values, which the reference's numbers include."""
    from ..eval.metrics import _to_uint8
    return _to_uint8(load_original(path, size))


def metric_row_u8(orig_u8: np.ndarray, recon_u8: np.ndarray) -> List[float]:
    from ..eval.metrics import psnr_u8, ssim_u8
    return [psnr_u8(orig_u8, recon_u8), ssim_u8(orig_u8, recon_u8), float("nan"), float("nan")]


def evaluate(manifest: List[dict], z_of: Callable[[dict], np.ndarray], reconstruct: Callable, size: int, batch: int,
             seed: Optional[int], rank: int, world: int, device: str, start_noise_fn: Callable, submit: Optional[Callable] = None,
             fetch: Optional[Callable] = None, workers: Optional[int] = None, u8: bool = False, marks: Optional[list] = None) -> np.ndarray:
    """Shard, reconstruct in batches, score on the host, gather.  ``reconstruct(z, x_T) -> (b,3,S,S)`` numpy in [-1,1].

    With ``submit(z, x_T, slot) -> handle`` / ``fetch(handle) -> numpy`` (the GPU path) two batches are kept in flight: batch k+1 is
    enqueued on the other stream before batch k's result is waited for -- consecutive batches are independent, and one batch's kernel
    tails and launch gaps fill with the other's work.  Everything the host does per record -- PNG decode + resize, the .clp read and z
    decode, the per-record start noise, PSNR / SSIM -- runs on a pool of ``workers`` threads (numpy / scipy / PIL / torch release the
    GIL in their kernels), inputs prepared one batch AHEAD of the GPU and metrics collected only at the end, so the submitting
    thread never waits for host work.  ``u8``: ``fetch`` returns the reconstruction already converted to uint8 on the device
    (a quarter of the D2H bytes; same arithmetic as ``_to_uint8``) and the originals are converted once."""
    mine = shard_indices(len(manifest), rank, world)
    nw = workers or default_workers(world)
    batches = [mine[lo:lo + batch] for lo in range(0, len(mine), batch)]
    rows_f: List = []                                              # one future per record, in `mine` order

    with ThreadPoolExecutor(max_workers=nw) as pool:
        def prepare(idx):
            """Futures of one batch's inputs: originals (float or uint8), z rows, start-noise rows (None when unseeded)."""
            load = original_u8 if u8 else load_original
            return dict(idx=idx,
                        orig=[pool.submit(load, manifest[i]["image"], size) for i in idx],
                        z=[pool.submit(z_of, manifest[i]) for i in idx],
                        noise=None if seed is None else [pool.submit(start_noise_fn, [i], size, seed) for i in idx])

        def inputs(p):
            z = np.concatenate([f.result() for f in p["z"]], 0)
            x_T = None if p["noise"] is None else torch.cat([f.result() for f in p["noise"]], 0)
            return z, x_T

        def score(p, recon):
            for k, o in enumerate(p["orig"]):
                if u8:
                    rows_f.append(pool.submit(lambda of, r: metric_row_u8(of.result(), r), o, recon[k]))
                else:
                    rows_f.append(pool.submit(lambda of, r: metric_row(of.result(), r, device), o, recon[k]))

        ahead = prepare(batches[0]) if batches else None
        pending = None                                             # (handle, prepared batch) still on the GPU
        for bi in range(len(batches)):
            cur = ahead
            ahead = prepare(batches[bi + 1]) if bi + 1 < len(batches) else None
            z, x_T = inputs(cur)
            if marks is not None:
                import time
                marks.append((time.perf_counter(), sum(len(b) for b in batches[:bi])))   # (when batch bi is handed over, records before it)
            if submit is not None and fetch is not None:
                handle = submit(z, x_T, bi & 1)
                if pending is not None:
                    score(pending[1], fetch(pending[0]))
                pending = (handle, cur)
            else:
                score(cur, reconstruct(z, x_T))
        if pending is not None:
            score(pending[1], fetch(pending[0]))
        rows = [f.result() for f in rows_f]
    local = np.asarray(rows, dtype=np.float64).reshape(len(mine), len(METRIC_KEYS))
    return gather_metric_rows(mine, local, len(manifest), device)


def main(argv=None) -> None:
    ap = argparse.ArgumentParser(description="Evaluate reconstruction quality on a store of images (MI355X build).")
    ap.add_argument("--store_dir", type=str, required=True)
    ap.add_argument("--weights", type=str, required=True)
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--eta", type=float, default=0.0)
    ap.add_argument("--device", type=str, default=None)
    ap.add_argument("--out_json", type=str, default=None)
    ap.add_argument("--batch", type=int, default=8, help="records per fused DDIM launch per GPU")
    ap.add_argument("--seed", type=int, default=None)
    ap.add_argument("--dtype", choices=["fp32", "bf16"], default="fp32")
    ap.add_argument("--timing", action="store_true", help="print set-up and loop wall time (records/s of the loop) to stderr")
    ap.add_argument("--workers", type=int, default=None, help="host threads for decode / metrics (default: this rank's share of the CPUs, 2..16)")
    ap.add_argument("--force-process-group", action="store_true",
                    help="initialise torch.distributed even for one rank, so that the final all-gather runs through the backend "
                         "(RCCL on a one-GPU box)")
    ap.add_argument("--gpus", type=int, default=None,
                    help="shard the store over this many GPUs of the node: one process per GPU is started here unless a launcher "
                         "(torchrun) already did; default: the launcher's WORLD_SIZE, else 1")
    args = ap.parse_args(argv)

    from ..utils.launch import ensure_ranks, init_process_group, rank_env, single_rank_env
    if args.gpus is not None:
        # starts the ranks (fresh child processes) and exits with their code unless this process already is one of them
        ensure_ranks(args.gpus, "clip_feature_codec.cli.eval", argv, module=True)

    from ._common import pick_device, load_codec_meta, load_embedding, build_model, build_sampler, start_noise
    import torch.distributed as dist

    rank, world, _ = rank_env()
    device = pick_device(args.device)
    use_pg = world > 1 or args.force_process_group
    if use_pg:
        if world == 1:
            single_rank_env()
        init_process_group(device)                                 # RCCL; CCN_DIST_BACKEND=gloo for several ranks on one card
        if args.force_process_group and rank == 0:
            import sys
            from ..utils.launch import collective_device
            print(f"[eval] process group: backend {dist.get_backend()}, world {dist.get_world_size()}, collective on "
                  f"{collective_device(device).split(':')[0]}", file=sys.stderr)

    import sys
    import time
    t_start = time.perf_counter()
    store_dir = Path(args.store_dir)
    manifest = json.loads((store_dir / "manifest.json").read_text(encoding="utf-8"))
    scale, zero = load_codec_meta(store_dir)
    net = build_model(args.weights, device, scale.shape[0], args.dtype)
    sampler = build_sampler(args.eta, device)

    def reconstruct(z: np.ndarray, x_T: Optional[torch.Tensor]) -> np.ndarray:
        zt = torch.from_numpy(z).to(device)
        with torch.no_grad():
            x = sampler.sample(net, zt, shape=(zt.shape[0], 3, args.size, args.size), steps=args.steps,
                               x_T=None if x_T is None else x_T.to(device))
        return x.clamp(-1, 1).cpu().numpy()

    streams = [torch.cuda.Stream(device=device), torch.cuda.Stream(device=device)]
    pinned: dict = {}
    # parallelism on the host comes from the worker threads; torch's own intra-op pool only gets in their way (a 196k-element
    # randn went from 1 to 60 ms per record when eight threads fought over it)
    torch.set_num_threads(1)
    from ..eval.metrics import learned_metrics_available
    # LPIPS / CLIP similarity need the float images (and packages that are absent offline); without them only uint8 leaves the GPU
    u8 = not learned_metrics_available()

    def submit(z: np.ndarray, x_T: Optional[torch.Tensor], slot: int):
        zt = torch.from_numpy(z).to(device)
        xt = None if x_T is None else x_T.to(device)
        st = streams[slot]
        st.wait_stream(torch.cuda.current_stream(device))
        with torch.no_grad(), torch.cuda.stream(st):
            x = sampler.sample(net, zt, shape=(zt.shape[0], 3, args.size, args.size), steps=args.steps, x_T=xt, slot=slot).clamp(-1, 1)
            if u8:
                # _to_uint8 (eval/metrics.py:16-19) on the device: the same two separately rounded fp32 ops, clip, truncation
                x = ((x + 1.0) * 127.5).clamp(0, 255).to(torch.uint8)
            key = (slot, tuple(x.shape), x.dtype)
            host = pinned.get(key)
            if host is None:
                host = pinned[key] = torch.empty(x.shape, dtype=x.dtype, pin_memory=True)
            host.copy_(x, non_blocking=True)
        ev = torch.cuda.Event(); ev.record(st)
        return host, ev, zt, xt, x                                 # device tensors kept alive until the result is fetched

    def fetch(handle) -> np.ndarray:
        handle[1].synchronize()
        net.native().poll_errors()                                 # a device-side failure flagged during this batch raises here
        return handle[0].numpy().copy()                            # the pinned buffer of this slot is reused two batches later

    pipelined = args.eta == 0                                      # the fused sampler; eta > 0 draws noise step by step
    t_loop = time.perf_counter()
    marks: list = []
    rows = evaluate(manifest, lambda rec: load_embedding(Path(rec["bitstream"]), scale, zero), reconstruct,
                    args.size, args.batch, args.seed, rank, world, device, start_noise,
                    submit if pipelined else None, fetch if pipelined else None, workers=args.workers, u8=u8 and pipelined,
                    marks=marks)
    if args.timing and rank == 0:
        t_end = time.perf_counter()
        print(f"[eval] set-up {t_loop - t_start:.2f} s (checkpoint load + weight repack), loop {t_end - t_loop:.2f} s for {len(manifest)} records = "
              f"{len(manifest) / (t_end - t_loop):.1f} records/s over {world} rank(s) (the first two batches include plan + graph capture)", file=sys.stderr)
        if len(marks) > 3:                                         # this rank's steady state: from the hand-over of its third batch on
            n_rank = len(shard_indices(len(manifest), rank, world))
            print(f"[eval] steady state of rank 0: {(n_rank - marks[2][1]) / (t_end - marks[2][0]):.1f} records/s per rank", file=sys.stderr)
    if rank == 0:
        agg = aggregate(rows)
        print(f"Average PSNR: {agg['psnr']:.2f} dB")
        print(f"Average SSIM: {agg['ssim']:.4f}")
        print(f"Average LPIPS: {agg['lpips']:.4f}")
        print(f"Average CLIP similarity: {agg['clip_sim']:.4f}")
        if args.out_json:
            recs = [dict(image=manifest[i]["image"], **{k: float(rows[i, j]) for j, k in enumerate(METRIC_KEYS)})
                    for i in range(len(manifest))]
            with open(args.out_json, "w", encoding="utf-8") as f:
                json.dump(recs, f, ensure_ascii=False, indent=2)
    if use_pg:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
