"""N > 1 through the HIP path, on one card: fresh ranks (gloo between them, every rank on cuda:0) started by the scripts' own
`--gpus N` handling.  RCCL itself needs a multi-GPU node (the driver's scaling run); what is checked here is everything else on that
route: the launcher, the rank-strided sharding of cli.eval with the REAL sampler, the single all-gather, the bench line's fields."""
import json
import os
import subprocess
import sys
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
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "CCN_LAUNCH_DEPTH")}
    env["PYTHONPATH"] = os.pathsep.join([str(PKG), str(REPO)] + ([env["PYTHONPATH"]] if env.get("PYTHONPATH") else []))
    env["CCN_DIST_BACKEND"] = "gloo"
    env.update(kw)
    return env


def _json_line(out: str) -> dict:
    return json.loads([ln for ln in out.splitlines() if ln.startswith("{")][-1])


def test_bench_gpus_2_alone_reports_two_ranks():
    """`python bench.py --gpus 2` with no launcher around it: n_gpus == 2, both ranks counted by an all-reduce of ones."""
    r = subprocess.run([sys.executable, str(REPO / "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--size", "64", "--ddim-steps", "4",
                        "--no-cpu-baseline", "--no-roofline", "--no-parity"], capture_output=True, text=True, env=_env(), timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = _json_line(r.stdout)
    assert line["n_gpus"] == 2 and line["rccl_ranks"] == 2 and line["backend"] == "gloo"
    assert line["config"]["global_batch"] == 16 and line["value"] > 0
    assert 0 < line["per_rank_images_per_sec"]["min"] <= line["per_rank_images_per_sec"]["max"]


def test_bench_refuses_a_disagreeing_world_size():
    r = subprocess.run([sys.executable, str(REPO / "bench.py"), "--gpus", "8", "--no-cpu-baseline"], capture_output=True, text=True,
                       env=_env(RANK="0", WORLD_SIZE="2", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999"), timeout=300)
    assert r.returncode != 0 and "refusing" in (r.stderr + r.stdout)
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_bench_train_gpus_2_alone_reports_two_ranks():
    r = subprocess.run([sys.executable, str(REPO / "bench_train.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--size", "64", "--batch", "2",
                        "--no-cpu-baseline", "--no-roofline"], capture_output=True, text=True, env=_env(), timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = _json_line(r.stdout)
    assert line["n_gpus"] == 2 and line["rccl_ranks"] == 2 and line["config"]["global_batch"] == 4 and np.isfinite(line["final_loss"])


def _run_eval(store, ckpt, out_json, gpus, batch, dtype, size, steps, extra=(), env=None):
    cmd = [sys.executable, "-m", "clip_feature_codec.cli.eval", "--store_dir", str(store), "--weights", str(ckpt), "--size", str(size),
           "--steps", str(steps), "--batch", str(batch), "--seed", "7", "--device", "cuda", "--dtype", dtype, "--out_json", str(out_json)]
    if gpus is not None:
        cmd += ["--gpus", str(gpus)]
    cmd += list(extra)
    r = subprocess.run(cmd, capture_output=True, text=True, env=env or _env(), timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    assert r.stdout.count("Average PSNR:") == 1                   # rank 0 only
    return json.loads(Path(out_json).read_text())


@pytest.mark.parametrize("n,batch,dtype,base,ch_mult,world", [
    (16, 8, "bf16", 128, (1, 2, 2), 2),     # C3's shape in small: C2 architecture, one full batch of 8 per rank, throughput mode
    (11, 4, "fp32", 32, (1, 2), 2),         # ragged: rank 0 gets 6 records (4 + 2), rank 1 gets 5 (4 + 1); single process 4 + 4 + 3
    (13, 2, "fp32", 32, (1, 2), 4),         # four ranks on the card, ragged: 4 / 3 / 3 / 3 records in batches of 2 (2+2, 2+1, 2+1, 2+1)
])
def test_cli_eval_ranks_equal_single_process(tmp_path, n, batch, dtype, base, ch_mult, world):
    """cli.eval (reference loop: cli/eval.py:56-86) with the real fused sampler on `world` fresh ranks sharding the store r::world, one
    all-gather at the end, against the single-process run of the same command: same records in manifest order, every metric equal.
    Records are independent units and start noise is seeded per record, so sharding must not change any row.  Rows are bit-equal when a
    record sits in a batch of the same size in both runs; in the ragged case the tail batches differ in size (3 vs 2 and 1), which can
    change the kernel chosen per layer and the summation order of the GroupNorm partial sums, so those rows carry a 1e-6 relative
    tolerance (fp32 mode)."""
    size, steps = 64, 5
    store = tmp_path / "store"
    synth.write_synth_store(store, n, size, write_clp=bitstream.write_bitstream)
    sd = synth.synth_state_dict(synth.unet_param_spec(512, base, ch_mult))
    ckpt = tmp_path / "ckpt.pt"
    torch.save({k: torch.from_numpy(v) for k, v in sd.items()}, ckpt)
    one = _run_eval(store, ckpt, tmp_path / "one.json", None, batch, dtype, size, steps)
    two = _run_eval(store, ckpt, tmp_path / "two.json", world, batch, dtype, size, steps)
    assert len(one) == len(two) == n and [r["image"] for r in one] == [r["image"] for r in two]
    same_batch_size = n % (world * batch) == 0
    for i, (a, b) in enumerate(zip(one, two)):
        assert np.isfinite(a["psnr"]) and np.isfinite(a["ssim"])
        for k in ("psnr", "ssim"):
            if same_batch_size:
                assert a[k] == b[k], (i, k, a[k], b[k])
            else:
                assert abs(a[k] - b[k]) <= 1e-6 * abs(a[k]), (i, k, a[k], b[k])
        for k in ("lpips", "clip_sim"):
            assert (np.isnan(a[k]) and np.isnan(b[k])) or a[k] == b[k]


def test_c3_workload_at_full_size_sharded_over_four_ranks_and_over_rccl(tmp_path):
    """BASELINE configs[2] at its real size -- 64 records, 256 px, base 128, (1,2,2), 50 DDIM steps, bf16, batch 8 -- through cli.eval
    (reference loop: cli/eval.py:56-86): (a) one process; (b) four fresh ranks on this one card (gloo between them; the process guard of
    the box allows six), rank r taking records r::4 in two batches of 8, one all-gather; (c) one rank in an RCCL group (backend nccl).
    Every record's PSNR / SSIM must be bit-equal in all three (independent units, per-record seeded start noise, batches of 8
    everywhere).  What this cannot show is eight devices and xGMI: the driver's scaling run."""
    n, size, steps, batch = 64, 256, 50, 8
    store = tmp_path / "store"
    synth.write_synth_store(store, n, size, write_clp=bitstream.write_bitstream)
    sd = synth.synth_state_dict(synth.unet_param_spec(512, 128, (1, 2, 2)))
    ckpt = tmp_path / "ckpt.pt"
    torch.save({k: torch.from_numpy(v) for k, v in sd.items()}, ckpt)
    one = _run_eval(store, ckpt, tmp_path / "one.json", None, batch, "bf16", size, steps)
    four = _run_eval(store, ckpt, tmp_path / "four.json", 4, batch, "bf16", size, steps)
    env = {k: v for k, v in _env().items() if k != "CCN_DIST_BACKEND"}
    rccl = _run_eval(store, ckpt, tmp_path / "rccl.json", None, batch, "bf16", size, steps, extra=["--force-process-group"], env=env)
    assert len(one) == len(four) == len(rccl) == n
    for a, b, c in zip(one, four, rccl):
        assert a["image"] == b["image"] == c["image"]
        assert np.isfinite(a["psnr"]) and a["psnr"] == b["psnr"] == c["psnr"] and a["ssim"] == b["ssim"] == c["ssim"]
