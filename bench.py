#!/usr/bin/env python3
"""Headline benchmark: images/sec of the 256 px / 50-step DDIM reconstruction, batch 8 per GPU.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch: ccn_sample of 8 images (256x256, base=128,
ch_mult=(1,2,2)) through 50 DDIM steps, replayed as one hipGraph, inputs resident in HBM.  Consecutive steps are
independent batches (like consecutive batches of cli.eval); they are timed one after the other.  `--inflight 2` keeps two
in flight per GPU, each on its own stream with its own workspace and graph (every launch still works on a batch of 8);
with the default that figure is measured after the timed region and reported as config.value_with_two_steps_in_flight.  Ranks hold
independent batches (weak scaling, no data-path collective).  Rank 0 prints ONE JSON line with the
contract fields plus `roofline` (dominant kernel, HIP-event timed in a launch-by-launch pass of the same
workload) and, at N=1, `cpu_baseline` (the oracle timed on the host cores on a bounded sample).
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

import numpy as np
import torch

PEAK = {"bf16": 2500.0, "fp32": 157.3}      # dense MFMA TFLOP/s, /opt/skills/guides/MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0
PMC_TRAFFIC_FILE = "r03_pmc_traffic.json"


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--dtype", choices=["bf16", "fp32"], default="bf16")
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--ddim-steps", type=int, default=50)
    ap.add_argument("--base", type=int, default=128)
    ap.add_argument("--ch-mult", type=str, default="1,2,2")
    ap.add_argument("--inflight", type=int, default=1, choices=[1, 2],
                    help="bench steps (independent batches) kept in flight per GPU, each on its own stream / workspace / graph, as cli.eval does")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--cpu-steps", type=int, default=50, help="most DDIM steps of the CPU-baseline sample (batch 1)")
    ap.add_argument("--force-process-group", action="store_true",
                    help="initialise torch.distributed (RCCL) even at --gpus 1: barrier / all-gather / all-reduce of the N > 1 route run on one GPU")
    ap.add_argument("--weight-rounding", choices=["phases", "diffused", "nearest"], default="phases",
                    help="bf16 mode: how ccn_commit_params rounds the conv weights (A/B runs; the default is the library's)")
    ap.add_argument("--no-parity", action="store_true", help="skip the parity block (bf16 vs the reference's golden / fp32 mode, PSNR delta)")
    args = ap.parse_args()

    # --gpus N without a launcher: start the N ranks ourselves (child processes, before this process touches the GPU)
    from clip_feature_codec.utils.launch import ensure_ranks, init_process_group, rank_env, collective_device, single_rank_env
    ensure_ranks(args.gpus, str(Path(__file__).resolve()))

    import torch.distributed as dist
    from clip_feature_codec import _native
    from clip_feature_codec.utils import synth
    from clip_feature_codec.models.unet import CLIPCondUNet
    from clip_feature_codec.diffusion.scheduler import NoiseScheduler
    from clip_feature_codec.diffusion.ddim import DDIMSampler

    rank, world, local = rank_env()
    assert world == args.gpus, (world, args.gpus)                    # ensure_ranks guarantees it
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X; the HIP path has no CPU fallback")
    dev = f"cuda:{local % torch.cuda.device_count()}"
    torch.cuda.set_device(dev)
    ranks_seen = 1
    use_pg = world > 1 or args.force_process_group
    if use_pg:
        if world == 1:
            single_rank_env()
        # RCCL ("nccl") on a GPU node; CCN_DIST_BACKEND=gloo rehearses the N > 1 path with several ranks on one card
        init_process_group(dev)
        ones = torch.ones(1, device=dev)
        dist.all_reduce(ones)                                        # every rank really is there: the sum of ones over the group
        ranks_seen = int(round(float(ones.item())))
        if ranks_seen != args.gpus or dist.get_world_size() != args.gpus:
            raise SystemExit(f"--gpus {args.gpus} but the process group has {ranks_seen} ranks")
    _native.load_library()

    ch_mult = tuple(int(v) for v in args.ch_mult.split(","))
    B, S, T = args.batch, args.size, args.ddim_steps
    sd = synth.synth_state_dict(synth.unet_param_spec(512, args.base, ch_mult))
    net = CLIPCondUNet(512, args.base, ch_mult, dtype=args.dtype, weight_rounding=args.weight_rounding).to(dev).eval()
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    idx = list(range(rank * B, rank * B + B))
    z = torch.from_numpy(synth.synth_z(world * B)[rank * B:rank * B + B]).to(dev)
    x_T = torch.from_numpy(synth.start_noise(idx, S, seed_base=100)).to(dev)
    sampler = DDIMSampler(NoiseScheduler(1000, "cosine", dev), eta=0.0)

    # The timed steps run one after the other by default.  Consecutive steps are independent batches (as consecutive batches of cli.eval
    # are), so they can also be kept in flight two at a time, step i on stream i % 2 with its own workspace and captured graph: one
    # batch's kernel tails and launch gaps then fill with the other's work.  `--inflight 2` times that; with the default the figure is
    # measured after the timed region and reported as config.value_with_two_steps_in_flight.
    nfl = args.inflight
    streams = [torch.cuda.Stream(device=dev) for _ in range(2)]

    def step(i=0, lanes=None):
        lanes = nfl if lanes is None else lanes
        st = streams[i % lanes]
        st.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(st):
            return sampler.sample(net, z, (B, 3, S, S), steps=T, x_T=x_T, slot=i % lanes)

    for i in range(2):                                               # plan + graph capture of both slots, outside warm-up and timing
        step(i, 2)

    def fence():
        torch.cuda.synchronize()
        if use_pg:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        x = step(i)
    fence()
    t0 = time.perf_counter()
    outs = [step(i) for i in range(args.steps)]
    torch.cuda.synchronize()
    dt_own = time.perf_counter() - t0                                # this rank's own work drained, before waiting for the others
    fence()
    x = outs[-1]
    dt = time.perf_counter() - t0
    per_rank = [B * args.steps / dt_own]
    if use_pg:
        mine = torch.tensor([dt, dt_own], dtype=torch.float64, device=collective_device(dev))
        every = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        dt = max(float(v[0].item()) for v in every)                  # MAX over ranks of barrier-to-barrier time
        per_rank = [B * args.steps / float(v[1].item()) for v in every]
    assert torch.isfinite(x).all()
    value = world * B * args.steps / dt
    # the other way of running the same steps (see above): a short second measurement, outside the timed region
    def timed(lanes, n):
        fence()
        t1 = time.perf_counter()
        keep = [step(i, lanes) for i in range(n)]
        fence()
        d = time.perf_counter() - t1
        if use_pg:
            tt = torch.tensor([d], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            d = float(tt.item())
        del keep
        return round(world * B * n / d, 3)
    other_lanes = 1 if nfl == 2 else 2
    other_value = timed(other_lanes, min(4, max(2, args.steps - args.steps % 2)))

    roofline = None
    if rank == 0 and not args.no_roofline:
        nat = net.native()
        nat.profile(True)
        sampler.sample(net, z, (B, 3, S, S), steps=T, x_T=x_T)       # launch by launch, HIP events around every kernel
        fams = nat.profile_read()
        nat.profile(False)
        total_ms = sum(f["ms"] for f in fams)
        dom = max(fams, key=lambda f: f["ms"])
        tfs = dom["flops"] / (dom["ms"] * 1e-3) / 1e12
        flops_fwd, bytes_fwd = nat.algorithmic_work(B, S, S)
        # HBM bytes per launch: a STORED figure from the rocprofv3 --pmc passes committed under profiles/ (FETCH_SIZE + WRITE_SIZE
        # in separate passes, tools/pmc_traffic.py) -- counters cannot be read from inside this process; only quoted for the
        # workload and library build it was collected on (the file names the library version), else null
        traffic = traffic_source = None
        pmc = REPO / "profiles" / PMC_TRAFFIC_FILE
        if pmc.exists() and args.dtype == "bf16" and (B, S, args.base, ch_mult) == (8, 256, 128, (1, 2, 2)):
            pj = json.loads(pmc.read_text())
            same_build = pj.get("library_version") in (None, _native.load_library().ccn_version().decode())
            if pj.get("kernel_family") == dom["name"] and same_build:
                traffic = round(pj["hbm_bytes_per_launch"])
                traffic_source = f"stored: profiles/{PMC_TRAFFIC_FILE} (rocprofv3 --pmc passes of this workload, not this run)"
        roofline = {
            "bound": "mfma", "kernel": dom["name"], "achieved": round(tfs, 2), "peak": PEAK[args.dtype], "unit": "TFLOP/s",
            "frac": round(tfs / PEAK[args.dtype], 4), "traffic": traffic, "traffic_source": traffic_source,
            "algorithmic_bytes_per_launch": round(dom["bytes"] / dom["calls"]),
            "launches": dom["calls"], "avg_launch_us": round(dom["ms"] * 1e3 / dom["calls"], 2),
            "algorithmic_gflop_per_launch": round(dom["flops"] / dom["calls"] / 1e9, 3),
            "kernel_hbm_gbs_algorithmic": round(dom["bytes"] / (dom["ms"] * 1e-3) / 1e9, 1),
            # informational, STORED from tools/power_probe_ubench.sh (finding 13): the dense bf16 rate the chip sustains on
            # random operands -- 1.77 GHz register-resident, 1.55-1.57 GHz with this kernel's LDS / L2 operand traffic at 99.6 % MFMA issue
            "sustained_peak_random_operands": ({"tflops": 1630.0, "frac": round(tfs / 1630.0, 4),
                                                "source": "stored: tools/ubench/consumer_loop.hip loop_rnd on all CUs, 1.55-1.57 GHz (not this run)"}
                                               if args.dtype == "bf16" else None),
            "families_ms": {f["name"]: round(f["ms"], 3) for f in fams},
            "event_pass_ms": round(total_ms, 2),
            "whole_forward": {"gflop_per_image": round(flops_fwd / B / 1e9, 2), "mb_per_image": round(bytes_fwd / B / 1e6, 1),
                              "tflops": round(flops_fwd * T * args.steps / dt / 1e12, 2),
                              "hbm_gbs_algorithmic": round(bytes_fwd * T * args.steps / dt / 1e9, 1),
                              "hbm_frac": round(bytes_fwd * T * args.steps / dt / 1e9 / HBM_PEAK_GBS, 4)},
        }

    # ---- parity of the timed mode, outside the timed region (rank 0): the bench batch itself, 50 steps, against
    # (a) the reference's device='cpu' run of record 0 (tests/golden/c2_sample.npz, made by tests/golden/make_golden.py),
    # (b) this library's fp32 parity mode on all rows, (c) PSNR against synthetic originals in both modes (north_star: 0.1 %)
    parity = parity_mode = None
    headline = (S, args.base, ch_mult, T) == (256, 128, (1, 2, 2), 50)
    if rank == 0 and not args.no_parity:
        from clip_feature_codec.eval.metrics import psnr
        x16 = sampler.sample(net, z, (B, 3, S, S), steps=T, x_T=x_T) if args.dtype == "bf16" else None
        net32 = CLIPCondUNet(512, args.base, ch_mult, dtype="fp32").to(dev).eval()
        net32.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
        x32 = sampler.sample(net32, z, (B, 3, S, S), steps=T, x_T=x_T)            # plan + capture
        torch.cuda.synchronize()
        p0 = time.perf_counter()
        x32 = sampler.sample(net32, z, (B, 3, S, S), steps=T, x_T=x_T)
        torch.cuda.synchronize()
        p_dt = time.perf_counter() - p0
        flops32, _ = net32.native().algorithmic_work(B, S, S)
        parity_mode = {"dtype": "fp32", "value": round(B / p_dt, 3), "unit": "images/sec", "ms_per_step": round(p_dt * 1e3, 2),
                       "tflops_whole_forward": round(flops32 * T / p_dt / 1e12, 2),
                       "frac": round(flops32 * T / p_dt / 1e12 / PEAK["fp32"], 4),
                       "note": "the mode that meets the 1e-3 max-abs gate; one timed step of the same batch, whole forward against the fp32 MFMA peak"}
        parity = {"workload": "the timed batch: rank 0's records, all DDIM steps"}
        gold = REPO / "tests" / "golden" / "c2_sample.npz"
        if headline and gold.exists():
            g = np.load(gold, allow_pickle=False)["x_final.sub"]                  # 1/16 strided sample of the reference's x_final
            d32 = np.abs(x32[0, :, ::4, ::4].cpu().numpy() - g)
            parity["fp32_mode_row0_vs_reference_cpu"] = {"max_abs": float(d32.max()), "mean_abs": float(d32.mean()), "gate_max_abs": 1e-3,
                                                         "pass": bool(d32.max() < 1e-3)}
            if x16 is not None:
                d16 = np.abs(x16[0, :, ::4, ::4].cpu().numpy() - g)
                parity["bf16_row0_vs_reference_cpu"] = {"max_abs": float(d16.max()), "mean_abs": float(d16.mean())}
        if x16 is not None:
            d = (x16 - x32).abs()
            parity["bf16_vs_fp32_mode_all_rows"] = {"max_abs": float(d.max()), "mean_abs": float(d.mean()),
                                                    "per_row_mean_abs": [round(float(v), 5) for v in d.mean((1, 2, 3))]}
            orig = [synth.synth_image(i, S).astype(np.float32).transpose(2, 0, 1) / 127.5 - 1.0 for i in idx]
            r16, r32 = x16.clamp(-1, 1).cpu().numpy(), x32.clamp(-1, 1).cpu().numpy()
            p16 = [psnr(orig[k], r16[k]) for k in range(B)]
            p32 = [psnr(orig[k], r32[k]) for k in range(B)]
            rel = [abs(a - b) / abs(b) for a, b in zip(p16, p32)]
            del net32
            parity["psnr_vs_synthetic_originals"] = {
                "records": B, "psnr_fp32_mode_mean_db": round(float(np.mean(p32)), 4), "psnr_bf16_mean_db": round(float(np.mean(p16)), 4),
                "max_rel_delta": float(max(rel)), "mean_rel_delta": float(np.mean(rel)),
                "rel_delta_of_means": float(abs(np.mean(p16) - np.mean(p32)) / abs(np.mean(p32))),
                "gate_rel": 1e-3, "pass": bool(max(rel) <= 1e-3),
                "weight_rounding": "error-diffused bf16 (ccn_set_weight_rounding default; independent rounding measures 0.165 % max)"}
        else:
            del net32
        del x32

    quality_gate_met = None
    if parity is not None:
        if "psnr_vs_synthetic_originals" in parity:
            quality_gate_met = bool(parity["psnr_vs_synthetic_originals"]["pass"])
        elif "fp32_mode_row0_vs_reference_cpu" in parity and args.dtype == "fp32":
            quality_gate_met = bool(parity["fp32_mode_row0_vs_reference_cpu"]["pass"])

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import ref_unet, ref_diffusion
        # the box's CPU share, not the host's core count: oversubscribed torch threads crawl under a cgroup quota
        ncpu = min(len(os.sched_getaffinity(0)), int(os.environ.get("CCN_CPU_THREADS", "16")))
        torch.set_num_threads(max(1, ncpu))
        osd = ref_unet.as_torch_sd(sd)
        model = ref_unet.make_model(osd)
        z1, x1 = z[:1].cpu(), x_T[:1].cpu()
        with torch.no_grad():
            w0 = time.perf_counter()
            model(x1, z1, torch.tensor([999]))
            t_fwd = time.perf_counter() - w0
        # bounded sample: about 20 s of CPU work (10-30 s), at least 2 and at most --cpu-steps DDIM steps
        n = max(2, min(args.cpu_steps, T, int(20.0 / max(t_fwd, 1e-3))))
        tab = ref_diffusion.scheduler_tables()
        ts = ref_diffusion.ddim_timesteps(1000, T)
        coefs = ref_diffusion.ddim_coefficients(tab, T)
        with torch.no_grad():
            model(x1, z1, torch.tensor([int(ts[0])]))              # warm the CPU caches / thread pool
            c0 = time.perf_counter()
            xc = x1
            for i in range(n):
                e = model(xc, z1, torch.tensor([int(ts[i])]))
                xc = ref_diffusion.ddim_update(xc, e, coefs[i])
            cdt = time.perf_counter() - c0
        cpu = {"value": round(1.0 / (cdt / n * T), 5), "unit": "images/sec", "cores": torch.get_num_threads(), "kind": "port",
               "sample": f"first {n} of {T} DDIM steps (UNet forward + update), batch 1, {S}px, fp32 torch-CPU oracle, "
                         f"{cdt:.1f}s measured" + ("" if n == T else f", extrapolated x{T / n:.2f}")}

    if rank == 0:
        line = {
            "metric": f"images/sec @{S}px {T}-step DDIM, batch={B}/GPU", "value": round(value, 3), "unit": "images/sec",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"{S}px base={args.base} ch_mult={ch_mult} {T}-step DDIM (eta=0), batch={B}/GPU, "
                                   "key-seeded synthetic weights (out.* x0.1), synthetic z / x_T",
                       "global_batch": world * B, "parallelism": f"dp{world} (independent batches, no collective in the loop)",
                       "graph": "hipGraph, one replay per step",
                       "bf16_weight_rounding": (args.weight_rounding + " (ccn_set_weight_rounding: error diffusion within the output channel"
                                                " and along the DDIM steps, eight bf16 versions of every conv weight used in turn)"
                                                if args.weight_rounding == "phases" else args.weight_rounding) if args.dtype == "bf16" else None,
                       "steps_in_flight": nfl,
                       ("value_with_two_steps_in_flight" if nfl == 1 else "value_with_one_step_in_flight"): other_value},
            "rccl_ranks": ranks_seen, "backend": (dist.get_backend() if use_pg else None),
            "per_rank_images_per_sec": {"min": round(min(per_rank), 3), "max": round(max(per_rank), 3)},
            # north_star's quality gate for the timed mode, on the timed batch: PSNR within 0.1 % of the reference-equivalent fp32 path per
            # record (bf16), or the 1e-3 max-abs gate against the reference's CPU run (fp32); null when the parity block was skipped
            "quality_gate_met": quality_gate_met,
            "roofline": roofline, "parity": parity, "parity_mode": parity_mode, "cpu_baseline": cpu,
        }
        print(json.dumps(line))
    if use_pg:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
