#!/usr/bin/env python3
"""Training-step benchmark (BASELINE.json configs[4]): images/sec of the epsilon-MSE step at 256 px, batch 4 per GPU.

    python bench_train.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench_train.py --gpus N --steps K --warmup W

Same contract as bench.py.  One "step" = the loop body of the reference's train/diffusion_train.py:119-124,137-140 on one
batch resident in HBM: draw t and noise, q_sample, UNet forward, MSE loss, backward, (N > 1: one all-reduce of the flat
gradient buffer over RCCL), AdamW.  Data-parallel: every rank holds its own batch of 4 (weak scaling; global batch 4 N).
Rank 0 prints ONE JSON line with `roofline` for the dominant kernel family (HIP events inside the library, on the stream the
kernels run on) and, at N = 1, `cpu_baseline` (the oracle's step -- torch-CPU autograd + AdamW -- on the host cores).
"""
from __future__ import annotations

import argparse
import os

# the pool's host driver only supports dmabuf IPC: without this RCCL fails with "hipIpcGetMemHandle: invalid argument";
# it is read when the HIP runtime starts, so it is set before torch is imported
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import json
import os
import sys
import time
from pathlib import Path

REPO = Path(__file__).resolve().parent
for p in (str(REPO), str(REPO / "clip-neural-image-conpression_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch

PEAK = {"bf16": 2500.0, "fp32": 157.3}      # dense MFMA TFLOP/s, /opt/skills/guides/MI355X_MICROARCH.md
PMC_TRAFFIC_TRAIN_FILE = "r03_pmc_traffic_train.json"


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--dtype", choices=["bf16", "fp32"], default="bf16")
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--base", type=int, default=128)
    ap.add_argument("--ch-mult", type=str, default="1,2,2")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    args = ap.parse_args()

    # --gpus N without a launcher: start the N ranks ourselves (child processes, before this process touches the GPU)
    from clip_feature_codec.utils.launch import ensure_ranks, init_process_group, rank_env
    ensure_ranks(args.gpus, str(Path(__file__).resolve()))

    import torch.distributed as dist
    from clip_feature_codec import _native
    from clip_feature_codec.utils import synth
    from clip_feature_codec.models.unet import CLIPCondUNet
    from clip_feature_codec.diffusion.scheduler import NoiseScheduler
    from clip_feature_codec.train.diffusion_train import FusedAdamW, train_step

    rank, world, local = rank_env()
    assert world == args.gpus, (world, args.gpus)                    # ensure_ranks guarantees it
    if not torch.cuda.is_available():
        raise SystemExit("bench_train.py needs an MI355X; the HIP path has no CPU fallback")
    dev = f"cuda:{local % torch.cuda.device_count()}"
    torch.cuda.set_device(dev)
    ranks_seen = 1
    if world > 1:
        # RCCL ("nccl") on a GPU node; CCN_DIST_BACKEND=gloo rehearses the N > 1 path with several ranks on one card
        init_process_group(dev)
        ones = torch.ones(1, device=dev)
        dist.all_reduce(ones)
        ranks_seen = int(round(float(ones.item())))
        if ranks_seen != args.gpus or dist.get_world_size() != args.gpus:
            raise SystemExit(f"--gpus {args.gpus} but the process group has {ranks_seen} ranks")
    _native.load_library()

    ch_mult = tuple(int(v) for v in args.ch_mult.split(","))
    B, S = args.batch, args.size
    sd = synth.synth_state_dict(synth.unet_param_spec(512, args.base, ch_mult))
    net = CLIPCondUNet(512, args.base, ch_mult, dtype=args.dtype).to(dev)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    net.train()
    sch = NoiseScheduler(1000, "cosine", dev)
    opt = FusedAdamW(net, lr=2e-4)
    g = torch.Generator("cpu").manual_seed(1000 + rank)
    x0 = (torch.rand((B, 3, S, S), generator=g) * 2 - 1).to(dev)
    z = torch.from_numpy(synth.synth_z(world * B)[rank * B:rank * B + B]).to(dev)

    def step():
        return train_step(net, sch, opt, x0, z, ddp=world > 1, graph=os.environ.get("CCN_TRAIN_GRAPH", "0") == "1")

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        loss = step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    assert torch.isfinite(loss) or os.environ.get("CCN_WG_DBG")
    value = world * B * args.steps / dt

    roofline = None
    if rank == 0 and not args.no_roofline:
        tr = net.train_state().trainer
        tr.profile(True)
        nprof = 3
        for _ in range(nprof):
            step()
        fams = [f for f in tr.profile_read() if f["calls"]]
        tr.profile(False)
        dom = max(fams, key=lambda f: f["ms"])
        mfma = [f for f in fams if f["flops"] > 0]
        total_flops = sum(f["flops"] for f in mfma) / nprof
        if dom["flops"] > 0:
            ach = dom["flops"] / (dom["ms"] * 1e-3) / 1e12
            head = {"bound": "mfma", "kernel": dom["name"], "achieved": round(ach, 2), "peak": PEAK[args.dtype], "unit": "TFLOP/s",
                    "frac": round(ach / PEAK[args.dtype], 4), "traffic": None,
                    "algorithmic_gflop_per_launch": round(dom["flops"] / dom["calls"] / 1e9, 3)}
        else:                               # a bandwidth-bound family leads: algorithmic HBM bytes / time against the 8 TB/s peak
            ach = dom["bytes"] / (dom["ms"] * 1e-3) / 1e9
            head = {"bound": "hbm", "kernel": dom["name"], "achieved": round(ach, 1), "peak": 8000.0, "unit": "GB/s",
                    "frac": round(ach / 8000.0, 4), "traffic": None,
                    "algorithmic_mb_per_launch": round(dom["bytes"] / dom["calls"] / 1e6, 2)}
        # HBM bytes per launch of the leading family from the --pmc passes of this workload (tools/profile_train.sh; a STORED figure)
        pmc = REPO / "profiles" / PMC_TRAFFIC_TRAIN_FILE
        if pmc.exists() and args.dtype == "bf16" and (B, S, args.base, ch_mult) == (4, 256, 128, (1, 2, 2)):
            fam = json.loads(pmc.read_text()).get("families", {}).get(dom["name"])
            if fam:
                head["traffic"] = round((fam["read_mb_per_launch_corrected_x2"] + fam["write_mb_per_launch"]) * 1e6)
                head["traffic_source"] = (f"stored: profiles/{PMC_TRAFFIC_TRAIN_FILE} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this workload, "
                                          "average over every launch of the family; not this run)")
        roofline = {
            **head,
            "launches": dom["calls"], "avg_launch_us": round(dom["ms"] * 1e3 / dom["calls"], 2),
            "families_ms_per_step": {f["name"]: round(f["ms"] / nprof, 3) for f in fams},
            "families_tflops": {f["name"]: round(f["flops"] / (f["ms"] * 1e-3) / 1e12, 1) for f in mfma},
            "families_hbm_gbs": {f["name"]: round(f["bytes"] / (f["ms"] * 1e-3) / 1e9, 1) for f in fams if f["bytes"] > 0},
            "event_pass_ms_per_step": round(sum(f["ms"] for f in fams) / nprof, 3),
            "whole_step": {"gflop_per_image": round(total_flops / B / 1e9, 2), "tflops": round(total_flops * args.steps / dt / 1e12, 2),
                           "mfma_frac": round(total_flops * args.steps / dt / 1e12 / PEAK[args.dtype], 4)},
        }

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import ref_unet, ref_train, ref_diffusion
        ncpu = min(len(os.sched_getaffinity(0)), int(os.environ.get("CCN_CPU_THREADS", "16")))
        torch.set_num_threads(max(1, ncpu))
        ref = ref_unet.as_torch_sd(sd)
        tab = ref_diffusion.scheduler_tables()
        x1, z1 = x0[:1].cpu(), z[:1].cpu()
        m = {k: torch.zeros_like(v) for k, v in ref.items()}; v2 = {k: torch.zeros_like(v) for k, v in ref.items()}
        gg = torch.Generator("cpu").manual_seed(7)
        n, c0 = 0, time.perf_counter()
        while n < 2 or (time.perf_counter() - c0 < 15.0 and n < 32):
            t1 = torch.randint(0, 1000, (1,), generator=gg); nz = torch.randn(x1.shape, generator=gg)
            _, grads, _, _ = ref_train.train_step_grads(ref, tab, x1, z1, t1, nz)
            for k in ref:
                ref[k], m[k], v2[k] = ref_train.adamw_update(ref[k], grads[k], m[k], v2[k], n + 1)
            n += 1
        cdt = time.perf_counter() - c0
        cpu = {"value": round(n / cdt, 5), "unit": "images/sec", "cores": torch.get_num_threads(), "kind": "port",
               "sample": f"{n} steps of batch 1 at {S}px (q_sample, forward, MSE, autograd backward, AdamW), fp32 torch-CPU oracle, {cdt:.1f}s measured"}

    if rank == 0:
        line = {
            "metric": f"train images/sec @{S}px eps-MSE step (fwd+bwd+AdamW), batch={B}/GPU", "value": round(value, 3), "unit": "images/sec",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"{S}px base={args.base} ch_mult={ch_mult} eps-MSE training step, batch={B}/GPU, key-seeded synthetic "
                                   "weights, synthetic x0 / z, t and noise drawn per step",
                       "global_batch": world * B,
                       "parallelism": f"dp{world}" + (" (one all-reduce of the flat fp32 gradient buffer per step, RCCL)" if world > 1 else "")},
            "rccl_ranks": ranks_seen, "backend": (dist.get_backend() if world > 1 else None),
            "final_loss": round(float(loss), 5), "roofline": roofline, "cpu_baseline": cpu,
        }
        print(json.dumps(line))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
