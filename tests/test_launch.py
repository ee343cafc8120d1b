"""`--gpus N` without a launcher starts N ranks itself (utils/launch.py); a rank count that disagrees with --gpus is an error,
never a silent single-rank run (the reference has no launcher: cli/eval.py:33 is one process)."""
import json
import os
import subprocess
import sys
import textwrap
from pathlib import Path

import pytest

from clip_feature_codec.utils import launch

REPO = Path(__file__).resolve().parent.parent
PKG = REPO / "clip-neural-image-conpression_amd"

SCRIPT = textwrap.dedent("""
    import argparse, json, os, sys
    sys.path[:0] = [{pkg!r}]
    from clip_feature_codec.utils.launch import ensure_ranks, rank_env
    ap = argparse.ArgumentParser(); ap.add_argument("--gpus", type=int, default=1); ap.add_argument("--tag", default="")
    a = ap.parse_args()
    ensure_ranks(a.gpus, os.path.abspath(__file__))
    import torch, torch.distributed as dist
    rank, world, local = rank_env()
    if world > 1:
        dist.init_process_group("gloo")
        ones = torch.ones(1); dist.all_reduce(ones); seen = int(ones.item())
    else:
        seen = 1
    if rank == 0:
        print(json.dumps({{"n_gpus": world, "ranks_seen": seen, "tag": a.tag}}))
    if world > 1:
        dist.destroy_process_group()
""")


def _clean_env():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "CCN_LAUNCH_DEPTH")}
    return env


def test_gpus_2_without_launcher_runs_two_ranks(tmp_path):
    script = tmp_path / "probe.py"
    script.write_text(SCRIPT.format(pkg=str(PKG)))
    r = subprocess.run([sys.executable, str(script), "--gpus", "2", "--tag", "x y"], capture_output=True, text=True, env=_clean_env(), timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line == {"n_gpus": 2, "ranks_seen": 2, "tag": "x y"}


def test_gpus_1_runs_in_place(tmp_path):
    script = tmp_path / "probe.py"
    script.write_text(SCRIPT.format(pkg=str(PKG)))
    r = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, env=_clean_env(), timeout=120)
    assert r.returncode == 0 and json.loads(r.stdout.strip().splitlines()[-1])["n_gpus"] == 1, r.stderr[-2000:]


def test_world_size_mismatch_is_an_error(monkeypatch):
    monkeypatch.setenv("RANK", "0"); monkeypatch.setenv("WORLD_SIZE", "2")
    with pytest.raises(SystemExit, match="refusing"):
        launch.ensure_ranks(8, "bench.py", [])
    launch.ensure_ranks(2, "bench.py", [])                     # agrees: returns
    monkeypatch.delenv("RANK"); monkeypatch.delenv("WORLD_SIZE")
    launch.ensure_ranks(1, "bench.py", [])                     # single process: returns
    monkeypatch.setenv("CCN_LAUNCH_DEPTH", "1")
    with pytest.raises(SystemExit, match="recursion"):
        launch.ensure_ranks(2, "bench.py", [])


def test_bench_scripts_call_the_launcher_before_touching_the_gpu():
    """bench.py / bench_train.py: ensure_ranks comes before the first torch.cuda / library call in main()."""
    for name in ("bench.py", "bench_train.py"):
        src = (REPO / name).read_text()
        body = src[src.index("def main()"):]
        assert "ensure_ranks(args.gpus" in body
        assert body.index("ensure_ranks(args.gpus") < body.index("torch.cuda."), name
        assert body.index("ensure_ranks(args.gpus") < body.index("load_library"), name
