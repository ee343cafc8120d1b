"""One process per GPU: turn ``script --gpus N`` into N ranks when nobody launched them for us.

The reference has no launcher (``cli/eval.py:33`` is a single process).  The benchmarks and ``cli.eval`` accept being
started either under ``python -m torch.distributed.run`` (RANK / WORLD_SIZE in the environment) or plainly as
``python script.py --gpus N``; in the second case :func:`ensure_ranks` re-runs the script as N fresh ranks through
``torch.distributed.run`` **before anything in this process touches the GPU** (a child process, never an exec), forwards
their output and exits with their code.  A rank count that disagrees with ``--gpus`` is an error, never a silent 1-GPU run.
"""
from __future__ import annotations

import os
import socket
import subprocess
import sys
from pathlib import Path
from typing import Optional, Sequence, Tuple


def free_port() -> int:
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return int(s.getsockname()[1])


def rank_env() -> Tuple[int, int, int]:
    """(rank, world, local_rank) from the torchrun environment; (0, 1, 0) when not launched."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0")))


def ensure_ranks(gpus: int, script: str, argv: Optional[Sequence[str]] = None, module: bool = False) -> None:
    """Return in a process that is one of ``gpus`` ranks (or the only one when ``gpus == 1``); otherwise launch them and exit.

    ``script`` is a file path, or a module name with ``module=True`` (``python -m torch.distributed.run -m pkg.mod``)."""
    if gpus < 1:
        raise SystemExit(f"--gpus {gpus}: need at least one")
    launched = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if launched:
        if world != gpus:
            raise SystemExit(f"--gpus {gpus} but launched with WORLD_SIZE={world}: refusing to report a {world}-rank run as {gpus} GPUs")
        return
    if gpus == 1:
        return
    if os.environ.get("CCN_LAUNCH_DEPTH"):
        raise SystemExit("launcher recursion: child started without RANK/WORLD_SIZE")
    argv = list(sys.argv[1:] if argv is None else argv)
    env = dict(os.environ)
    env["CCN_LAUNCH_DEPTH"] = "1"
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    pkg_root = str(Path(__file__).resolve().parents[2])                # so that `-m clip_feature_codec...` resolves in the children
    env["PYTHONPATH"] = os.pathsep.join([pkg_root] + ([env["PYTHONPATH"]] if env.get("PYTHONPATH") else []))
    env.setdefault("OMP_NUM_THREADS", str(max(1, len(os.sched_getaffinity(0)) // gpus)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port())]
    cmd += (["-m", script] if module else [script]) + argv
    rc = subprocess.run(cmd, env=env).returncode
    raise SystemExit(rc)


def single_rank_env() -> None:
    """Rendezvous variables of a one-rank group for a process no launcher started (``--force-process-group``: the collectives of
    the N > 1 route -- RCCL included -- then run on the one GPU a single-GPU box has)."""
    os.environ.setdefault("RANK", "0")
    os.environ.setdefault("WORLD_SIZE", "1")
    os.environ.setdefault("LOCAL_RANK", "0")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if "MASTER_PORT" not in os.environ:
        os.environ["MASTER_PORT"] = str(free_port())


def init_process_group(device: str):
    """RCCL (``backend='nccl'``) on a GPU node; ``CCN_DIST_BACKEND=gloo`` rehearses N > 1 with several ranks on one card."""
    import torch
    import torch.distributed as dist
    if dist.is_initialized():
        return
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # the pool's driver only supports dmabuf IPC (RCCL needs it)
    backend = os.environ.get("CCN_DIST_BACKEND", "nccl")
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=torch.device(device))
    else:
        dist.init_process_group(backend)


def collective_device(device: str) -> str:
    """Where tensors handed to a collective must live: the GPU under RCCL, the host under gloo."""
    import torch.distributed as dist
    return device if (dist.is_initialized() and dist.get_backend() == "nccl") else "cpu"
