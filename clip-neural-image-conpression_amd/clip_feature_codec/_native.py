"""ctypes binding of libccn_hip.so (include/ccn_hip.h) -- the only way this package computes.

There is no CPU or eager-PyTorch fallback: if the library is missing, or a tensor is not
on a HIP device, the call raises.  Build the library with ``python __graft_entry__.py``
(or ``make -C clip-neural-image-conpression_amd/csrc``).
"""
from __future__ import annotations

import ctypes
import os
from pathlib import Path
from typing import Dict, List, Optional, Sequence, Tuple

import torch

CSRC = Path(__file__).resolve().parent.parent / "csrc"
LIB_PATH = Path(os.environ.get("CCN_HIP_LIB", CSRC / "libccn_hip.so"))

DTYPE_F32, DTYPE_BF16 = 0, 1
_DTYPES = {"fp32": DTYPE_F32, "f32": DTYPE_F32, "float32": DTYPE_F32, "bf16": DTYPE_BF16, "bfloat16": DTYPE_BF16}

c_i32, c_i64, c_f32, c_vp, c_sz = ctypes.c_int32, ctypes.c_int64, ctypes.c_float, ctypes.c_void_p, ctypes.c_size_t


class CcnConfig(ctypes.Structure):
    _fields_ = [("z_dim", c_i32), ("base", c_i32), ("n_mult", c_i32), ("ch_mult", c_i32 * 8),
                ("time_dim", c_i32), ("img_ch", c_i32), ("groups", c_i32), ("dtype", c_i32)]


# name -> (restype, argtypes); must list every symbol include/ccn_hip.h declares
SIGNATURES = {
    "ccn_create": (c_i32, [ctypes.POINTER(CcnConfig), ctypes.POINTER(c_vp)]),
    "ccn_destroy": (c_i32, [c_vp]),
    "ccn_num_params": (c_i32, [c_vp, ctypes.POINTER(c_i32)]),
    "ccn_param_info": (c_i32, [c_vp, c_i32, ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(c_i64), ctypes.POINTER(c_i32)]),
    "ccn_load_param": (c_i32, [c_vp, ctypes.c_char_p, c_vp, ctypes.POINTER(c_i64), c_i32]),
    "ccn_commit_params": (c_i32, [c_vp]),
    "ccn_set_weight_rounding": (c_i32, [c_vp, c_i32]),
    "ccn_workspace_bytes": (c_i32, [c_vp, c_i32, c_i32, c_i32, c_i32, ctypes.POINTER(c_sz)]),
    "ccn_release_workspace": (c_i32, [c_vp, c_vp]),
    "ccn_forward": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_vp, c_sz, c_vp]),
    "ccn_sample": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp, c_sz, c_vp, c_i32]),
    "ccn_sample_eta": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_sz, c_vp, c_i32]),
    "ccn_ddim_step": (c_i32, [c_vp, c_vp, c_vp, c_f32, c_f32, c_f32, c_f32, c_f32, c_i64, c_vp]),
    "ccn_q_sample": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i64, c_vp]),
    "ccn_predict_x0": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i64, c_vp]),
    "ccn_film_forward": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp]),
    "ccn_resblock_forward": (c_i32, [c_vp, ctypes.c_char_p, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_vp, c_sz, c_vp]),
    "ccn_timestep_embedding": (c_i32, [c_vp, c_vp, c_i32, c_i32, c_vp]),
    "ccn_read_activation": (c_i32, [c_vp, ctypes.c_char_p, c_vp, c_sz, c_vp]),
    "ccn_poll_errors": (c_i32, [c_vp]),
    "ccn_profile_enable": (c_i32, [c_vp, c_i32]),
    "ccn_profile_read": (c_i32, [c_vp, ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(c_f32), ctypes.POINTER(c_i32),
                                 ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double), c_i32, ctypes.POINTER(c_i32)]),
    "ccn_algorithmic_work": (c_i32, [c_vp, c_i32, c_i32, c_i32, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]),
    "ccn_train_create": (c_i32, [ctypes.POINTER(CcnConfig), ctypes.POINTER(c_vp)]),
    "ccn_train_destroy": (c_i32, [c_vp]),
    "ccn_train_num_params": (c_i32, [c_vp, ctypes.POINTER(c_i32), ctypes.POINTER(c_i64)]),
    "ccn_train_param_info": (c_i32, [c_vp, c_i32, ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(c_i64), ctypes.POINTER(c_i32),
                                     ctypes.POINTER(c_i64)]),
    "ccn_train_workspace_bytes": (c_i32, [c_vp, c_i32, c_i32, c_i32, ctypes.POINTER(c_sz)]),
    "ccn_train_forward": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_vp, c_sz, c_vp]),
    "ccn_train_backward": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_vp, c_sz, c_vp]),
    "ccn_train_backward_bucketed": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_vp, c_sz, c_vp, c_i64, c_vp, c_vp]),
    "ccn_train_set_graph": (c_i32, [c_vp, c_i32]),
    "ccn_train_profile_enable": (c_i32, [c_vp, c_i32]),
    "ccn_train_profile_read": (c_i32, [c_vp, ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(c_f32), ctypes.POINTER(c_i32),
                                       ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double), c_i32, ctypes.POINTER(c_i32)]),
    "ccn_mse_loss_grad": (c_i32, [c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp]),
    "ccn_adamw_step": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_i64, c_f32, c_f32, c_f32, c_f32, c_f32, c_i32, c_vp]),
    "ccn_adamw_step_zero_grad": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_i64, c_f32, c_f32, c_f32, c_f32, c_f32, c_i32, c_vp]),
    "ccn_last_error": (ctypes.c_char_p, []),
    "ccn_version": (ctypes.c_char_p, []),
}

GRAD_READY_CB = ctypes.CFUNCTYPE(None, c_vp, c_i64, c_i64)

_lib = None


def load_library(path: Optional[os.PathLike] = None) -> ctypes.CDLL:
    """dlopen libccn_hip.so and set every prototype.  Raises if it is not built."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = Path(path) if path is not None else LIB_PATH
    if not p.exists():
        raise RuntimeError(
            f"HIP library {p} is not built; this package has no CPU fallback. "
            "Run `python __graft_entry__.py` at the repository root to compile it for gfx950.")
    lib = ctypes.CDLL(str(p))
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the .so lacks a declared symbol
        fn.restype, fn.argtypes = res, args
    if path is None:
        _lib = lib
    return lib


def dtype_code(dtype) -> int:
    if isinstance(dtype, int):
        return dtype
    if isinstance(dtype, torch.dtype):
        dtype = {torch.float32: "fp32", torch.bfloat16: "bf16"}[dtype]
    return _DTYPES[str(dtype).lower()]


def check(rc: int) -> None:
    if rc == 0:
        return
    msg = load_library().ccn_last_error().decode("utf-8", "replace")
    if rc in (1, 3, 4):          # EINVAL / EWEIGHTS / EWORKSPACE
        raise (RuntimeError if rc != 1 else ValueError)(f"ccn_hip: {msg}")
    raise RuntimeError(f"ccn_hip (code {rc}): {msg}")


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def require_dev(t: torch.Tensor, name: str, dtype=torch.float32) -> torch.Tensor:
    """The tensor as the C ABI wants it: on a HIP device, contiguous, of `dtype`."""
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name} must be a torch.Tensor")
    if not t.is_cuda:
        raise RuntimeError(f"{name} is on {t.device}: the MI355X path has no CPU fallback; move it to a HIP device ('cuda')")
    if t.dtype != dtype:
        t = t.to(dtype)
    return t.contiguous()


def current_stream(device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


class Workspace:
    """Caller-owned scratch for one (B,H,W,steps); allocated through torch's caching allocator."""

    def __init__(self, nbytes: int, device) -> None:
        self.buf = torch.empty(nbytes + 512, dtype=torch.uint8, device=device)
        base = self.buf.data_ptr()
        self.ptr = (base + 255) // 256 * 256
        self.nbytes = nbytes


class NativeUNet:
    """One ccn_handle_t: repacked weights + cached plans/graphs for a CLIPCondUNet."""

    MAX_WORKSPACES = 8

    def __init__(self, z_dim: int, base: int, ch_mult: Sequence[int], time_dim: int, img_ch: int,
                 groups: int = 8, dtype="fp32", device="cuda", weight_rounding: str = "phases") -> None:
        self.lib = load_library()
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("NativeUNet needs a HIP device ('cuda'); there is no CPU fallback")
        self.dtype = dtype_code(dtype)
        cfg = CcnConfig(z_dim, base, len(ch_mult), (c_i32 * 8)(*list(ch_mult)), time_dim, img_ch, groups, self.dtype)
        self.cfg = cfg
        self.z_dim, self.img_ch, self.time_dim = z_dim, img_ch, time_dim
        h = c_vp()
        with torch.cuda.device(self.device):
            check(self.lib.ccn_create(ctypes.byref(cfg), ctypes.byref(h)))
        self.h = h
        modes = {"nearest": 0, "diffused": 1, "phases": 2}
        if weight_rounding not in modes:
            raise ValueError(f"weight_rounding must be one of {sorted(modes)}")
        check(self.lib.ccn_set_weight_rounding(self.h, modes[weight_rounding]))
        self._ws: Dict[Tuple[int, int, int, int, int], Workspace] = {}

    def close(self) -> None:
        if getattr(self, "h", None):
            self.lib.ccn_destroy(self.h)
            self.h = None
            self._ws.clear()

    def __del__(self) -> None:  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    # -- parameters --------------------------------------------------------------------------
    def param_spec(self) -> List[Tuple[str, Tuple[int, ...]]]:
        n = c_i32()
        check(self.lib.ccn_num_params(self.h, ctypes.byref(n)))
        out = []
        for i in range(n.value):
            name, shape, nd = ctypes.c_char_p(), (c_i64 * 4)(), c_i32()
            check(self.lib.ccn_param_info(self.h, i, ctypes.byref(name), shape, ctypes.byref(nd)))
            out.append((name.value.decode(), tuple(int(shape[k]) for k in range(nd.value))))
        return out

    def load_state_dict(self, sd) -> None:
        """strict load: every key once, right shape (errors come from the library)."""
        with torch.cuda.device(self.device):
            for name, v in sd.items():
                t = v if isinstance(v, torch.Tensor) else torch.as_tensor(v)
                t = t.detach().to("cpu", torch.float32).contiguous()
                shape = (c_i64 * max(t.dim(), 1))(*t.shape)
                check(self.lib.ccn_load_param(self.h, name.encode(), t.data_ptr(), shape, t.dim()))
            check(self.lib.ccn_commit_params(self.h))
        self._ws.clear()

    # -- scratch -----------------------------------------------------------------------------
    def workspace(self, B: int, H: int, W: int, steps: int, slot: int = 0) -> Workspace:
        """``slot``: independent scratch (and, inside the library, plan + captured graph) for callers that keep several batches in flight."""
        key = (B, H, W, steps, slot)
        ws = self._ws.pop(key, None)
        if ws is None:
            # as many live workspaces as the library caches plans (8): the least recently used one goes back to torch's
            # allocator, after draining the device -- a replay on a side stream may still be writing it
            while len(self._ws) >= self.MAX_WORKSPACES:
                self._release(self._ws.pop(next(iter(self._ws))))
            n = c_sz()
            check(self.lib.ccn_workspace_bytes(self.h, B, H, W, steps, ctypes.byref(n)))
            ws = Workspace(n.value, self.device)
        self._ws[key] = ws                        # (re)insert as most recently used
        return ws

    def _release(self, ws: "Workspace") -> None:
        """Before a workspace goes back to torch's allocator: the library drains the device and drops every plan / captured graph
        that lives in it (a later workspace of the same shape may land on the same address and must not find the stale plan)."""
        with torch.cuda.device(self.device):
            check(self.lib.ccn_release_workspace(self.h, ws.ptr))

    # -- hot path ----------------------------------------------------------------------------
    def forward(self, x: torch.Tensor, z: torch.Tensor, t: torch.Tensor) -> torch.Tensor:
        x = require_dev(x, "x_t"); z = require_dev(z, "z_clip"); t = require_dev(t, "t", torch.int64)
        B, C, H, W = x.shape
        if C != self.img_ch or z.shape != (B, self.z_dim) or t.shape != (B,):
            raise ValueError(f"shape mismatch: x {tuple(x.shape)}, z {tuple(z.shape)}, t {tuple(t.shape)}")
        ws = self.workspace(B, H, W, 1)
        out = torch.empty_like(x)
        with torch.cuda.device(x.device):
            check(self.lib.ccn_forward(self.h, x.data_ptr(), z.data_ptr(), t.data_ptr(), out.data_ptr(), B, H, W,
                                       ws.ptr, ws.nbytes, current_stream(x.device)))
        return out

    def sample(self, z: torch.Tensor, x_T: torch.Tensor, ts, coef, use_graph: bool = True, slot: int = 0,
               sigma=None, noise: Optional[torch.Tensor] = None) -> torch.Tensor:
        """``sigma`` (steps,) + ``noise`` (steps, B, C, H, W) fp32 on the device: the eta > 0 loop (ccn_sample_eta)."""
        import numpy as np
        z = require_dev(z, "z_clip"); x_T = require_dev(x_T, "x_T")
        B, C, H, W = x_T.shape
        ts = np.ascontiguousarray(ts, dtype=np.int32)
        coef = np.ascontiguousarray(coef, dtype=np.float32)
        steps = int(ts.shape[0])
        if coef.shape != (steps, 4):
            raise ValueError("coef must be (steps, 4)")
        if C != self.img_ch or z.shape != (B, self.z_dim):
            raise ValueError(f"shape mismatch: x_T {tuple(x_T.shape)}, z {tuple(z.shape)}")
        ws = self.workspace(B, H, W, steps, slot)
        out = torch.empty_like(x_T)
        with torch.cuda.device(x_T.device):
            if sigma is not None:
                sigma = np.ascontiguousarray(sigma, dtype=np.float32)
                noise = require_dev(noise, "noise")
                if sigma.shape != (steps,) or tuple(noise.shape) != (steps, B, C, H, W):
                    raise ValueError(f"sigma must be ({steps},) and noise ({steps}, {B}, {C}, {H}, {W})")
                check(self.lib.ccn_sample_eta(self.h, z.data_ptr(), x_T.data_ptr(), out.data_ptr(), B, H, W, steps,
                                              ts.ctypes.data, coef.ctypes.data, sigma.ctypes.data, noise.data_ptr(), ws.ptr, ws.nbytes,
                                              current_stream(x_T.device), 1 if use_graph else 0))
            else:
                check(self.lib.ccn_sample(self.h, z.data_ptr(), x_T.data_ptr(), out.data_ptr(), B, H, W, steps,
                                          ts.ctypes.data, coef.ctypes.data, ws.ptr, ws.nbytes,
                                          current_stream(x_T.device), 1 if use_graph else 0))
        return out

    def resblock(self, prefix: str, x: torch.Tensor, h: torch.Tensor) -> torch.Tensor:
        x = require_dev(x, "x"); h = require_dev(h, "h")
        B, C, H, W = x.shape
        out = torch.empty_like(x)
        nbytes = 64 * B * H * W * C + (1 << 20)
        ws = Workspace(nbytes, x.device)
        with torch.cuda.device(x.device):
            check(self.lib.ccn_resblock_forward(self.h, prefix.encode(), x.data_ptr(), h.data_ptr(), out.data_ptr(),
                                                B, H, W, ws.ptr, ws.nbytes, current_stream(x.device)))
            torch.cuda.current_stream(x.device).synchronize()   # ws is freed on return
        return out

    def read_activation(self, name: str, shape: Tuple[int, int, int, int]) -> torch.Tensor:
        out = torch.empty(shape, dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            check(self.lib.ccn_read_activation(self.h, name.encode(), out.data_ptr(), out.numel(), current_stream(self.device)))
        return out

    def poll_errors(self) -> None:
        """Raise if a kernel reported a device-side failure since the last check (call after synchronising the stream)."""
        check(self.lib.ccn_poll_errors(self.h))

    def profile(self, on: bool) -> None:
        check(self.lib.ccn_profile_enable(self.h, 1 if on else 0))

    def profile_read(self) -> List[dict]:
        cap = 16
        names = (ctypes.c_char_p * cap)(); ms = (c_f32 * cap)(); calls = (c_i32 * cap)()
        fl = (ctypes.c_double * cap)(); by = (ctypes.c_double * cap)(); n = c_i32()
        with torch.cuda.device(self.device):
            check(self.lib.ccn_profile_read(self.h, names, ms, calls, fl, by, cap, ctypes.byref(n)))
        return [dict(name=names[i].decode(), ms=float(ms[i]), calls=int(calls[i]), flops=float(fl[i]), bytes=float(by[i]))
                for i in range(n.value)]

    def algorithmic_work(self, B: int, H: int, W: int) -> Tuple[float, float]:
        f, b = ctypes.c_double(), ctypes.c_double()
        check(self.lib.ccn_algorithmic_work(self.h, B, H, W, ctypes.byref(f), ctypes.byref(b)))
        return f.value, b.value


class NativeTrainer:
    """One ccn_trainer_t: the training forward + backward of a CLIPCondUNet over a flat fp32 parameter buffer."""

    def __init__(self, z_dim: int, base: int, ch_mult: Sequence[int], time_dim: int, img_ch: int,
                 groups: int = 8, dtype="fp32", device="cuda") -> None:
        self.lib = load_library()
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("NativeTrainer needs a HIP device ('cuda'); there is no CPU fallback")
        self.dtype = dtype_code(dtype)
        cfg = CcnConfig(z_dim, base, len(ch_mult), (c_i32 * 8)(*list(ch_mult)), time_dim, img_ch, groups, self.dtype)
        self.z_dim, self.img_ch = z_dim, img_ch
        h = c_vp()
        with torch.cuda.device(self.device):
            check(self.lib.ccn_train_create(ctypes.byref(cfg), ctypes.byref(h)))
        self.h = h
        self._ws: Dict[Tuple[int, int, int], Workspace] = {}
        n, total = c_i32(), c_i64()
        check(self.lib.ccn_train_num_params(self.h, ctypes.byref(n), ctypes.byref(total)))
        self.total = int(total.value)
        self.serial = 0                       # forwards run so far: the workspace holds the activations of forward number `serial`
        self.layout: List[Tuple[str, Tuple[int, ...], int]] = []
        for i in range(n.value):
            name, shape, nd, off = ctypes.c_char_p(), (c_i64 * 4)(), c_i32(), c_i64()
            check(self.lib.ccn_train_param_info(self.h, i, ctypes.byref(name), shape, ctypes.byref(nd), ctypes.byref(off)))
            self.layout.append((name.value.decode(), tuple(int(shape[k]) for k in range(nd.value)), int(off.value)))

    def close(self) -> None:
        if getattr(self, "h", None):
            self.lib.ccn_train_destroy(self.h)
            self.h = None
            self._ws.clear()

    def __del__(self) -> None:  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    def workspace(self, B: int, H: int, W: int) -> Workspace:
        key = (B, H, W)
        ws = self._ws.get(key)
        if ws is None:
            n = c_sz()
            check(self.lib.ccn_train_workspace_bytes(self.h, B, H, W, ctypes.byref(n)))
            self._ws.clear()                      # one live shape at a time: the activations of a step are large
            ws = Workspace(n.value, self.device)
            self._ws[key] = ws
        return ws

    def set_graph(self, on: bool) -> None:
        """Capture / replay forward and backward as hipGraphs (callers with fixed buffer addresses only)."""
        check(self.lib.ccn_train_set_graph(self.h, 1 if on else 0))

    def forward(self, flat: torch.Tensor, x: torch.Tensor, z: torch.Tensor, t: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        B, C, H, W = x.shape
        if C != self.img_ch or z.shape != (B, self.z_dim) or t.shape != (B,) or flat.numel() != self.total:
            raise ValueError(f"shape mismatch: x {tuple(x.shape)}, z {tuple(z.shape)}, t {tuple(t.shape)}, params {flat.numel()}")
        ws = self.workspace(B, H, W)
        if out is None:
            out = torch.empty_like(x)
        self.serial += 1
        with torch.cuda.device(x.device):
            check(self.lib.ccn_train_forward(self.h, flat.data_ptr(), x.data_ptr(), z.data_ptr(), t.data_ptr(), out.data_ptr(),
                                             B, H, W, ws.ptr, ws.nbytes, current_stream(x.device)))
        return out

    def backward(self, flat: torch.Tensor, gflat: torch.Tensor, x: torch.Tensor, z: torch.Tensor, d_eps: torch.Tensor,
                 bucket_cb=None, bucket_floats: int = 6 << 20) -> None:
        """``bucket_cb(lo, hi)``: called as soon as ``gflat[lo:hi]`` is complete in stream order (ccn_train_backward_bucketed)."""
        B, C, H, W = x.shape
        ws = self.workspace(B, H, W)
        if bucket_cb is not None:
            errs = []

            def tramp(_user, lo, hi):
                try:
                    bucket_cb(int(lo), int(hi))
                except BaseException as exc:          # never unwind through the C frame
                    errs.append(exc)
            cfn = GRAD_READY_CB(tramp)
            with torch.cuda.device(x.device):
                check(self.lib.ccn_train_backward_bucketed(self.h, flat.data_ptr(), gflat.data_ptr(), x.data_ptr(), z.data_ptr(), d_eps.data_ptr(),
                                                           B, H, W, ws.ptr, ws.nbytes, current_stream(x.device), int(bucket_floats), cfn, None))
            if errs:
                raise errs[0]
            return
        with torch.cuda.device(x.device):
            check(self.lib.ccn_train_backward(self.h, flat.data_ptr(), gflat.data_ptr(), x.data_ptr(), z.data_ptr(), d_eps.data_ptr(),
                                              B, H, W, ws.ptr, ws.nbytes, current_stream(x.device)))


    def profile(self, on: bool) -> None:
        check(self.lib.ccn_train_profile_enable(self.h, 1 if on else 0))

    def profile_read(self) -> List[dict]:
        cap = 16
        names = (ctypes.c_char_p * cap)(); ms = (c_f32 * cap)(); calls = (c_i32 * cap)(); n = c_i32()
        fl = (ctypes.c_double * cap)(); by = (ctypes.c_double * cap)()
        with torch.cuda.device(self.device):
            check(self.lib.ccn_train_profile_read(self.h, names, ms, calls, fl, by, cap, ctypes.byref(n)))
        return [dict(name=names[i].decode(), ms=float(ms[i]), calls=int(calls[i]), flops=float(fl[i]), bytes=float(by[i])) for i in range(n.value)]


def mse_loss_grad(eps: torch.Tensor, target: torch.Tensor, want_grad: bool = True, bufs=None):
    """F.mse_loss(eps, target) and d loss / d eps in one pass (train/diffusion_train.py:124)."""
    lib = load_library()
    eps = require_dev(eps, "eps"); target = require_dev(target, "target")
    loss, d, scratch = (bufs if bufs is not None else (None, None, None))
    if loss is None:
        loss = torch.empty((), dtype=torch.float32, device=eps.device)
        d = torch.empty_like(eps) if want_grad else None
        scratch = torch.empty(1024, dtype=torch.float32, device=eps.device)
    with torch.cuda.device(eps.device):
        check(lib.ccn_mse_loss_grad(eps.data_ptr(), target.data_ptr(), eps.numel(), loss.data_ptr(), ptr(d), scratch.data_ptr(),
                                    current_stream(eps.device)))
    return loss, d


def adamw_step(p: torch.Tensor, g: torch.Tensor, m: torch.Tensor, v: torch.Tensor, lr: float, beta1: float, beta2: float,
               eps: float, weight_decay: float, step: int, zero_grad: bool = False) -> None:
    """One torch.optim.AdamW update over flat fp32 buffers, in place; ``zero_grad``: the gradients are left at zero by the same pass."""
    lib = load_library()
    for name, tns in (("params", p), ("grads", g), ("exp_avg", m), ("exp_avg_sq", v)):
        if not (tns.is_cuda and tns.dtype == torch.float32 and tns.is_contiguous() and tns.numel() == p.numel()):
            raise ValueError(f"{name} must be a contiguous fp32 HIP tensor of {p.numel()} elements")
    with torch.cuda.device(p.device):
        fn = lib.ccn_adamw_step_zero_grad if zero_grad else lib.ccn_adamw_step
        check(fn(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(), float(lr), float(beta1),
                 float(beta2), float(eps), float(weight_decay), int(step), current_stream(p.device)))


# ---- stateless ops --------------------------------------------------------------------------------

def ddim_step(x: torch.Tensor, eps: torch.Tensor, coef: Sequence[float], sigma: float = 0.0,
              noise: Optional[torch.Tensor] = None) -> torch.Tensor:
    """In-place DDIM update of ``x`` (diffusion/ddim.py:34-45)."""
    lib = load_library()
    x = require_dev(x, "x"); eps = require_dev(eps, "eps")
    nz = require_dev(noise, "noise") if noise is not None else None
    with torch.cuda.device(x.device):
        check(lib.ccn_ddim_step(x.data_ptr(), eps.data_ptr(), ptr(nz), float(coef[0]), float(coef[1]), float(coef[2]),
                                float(coef[3]), float(sigma), x.numel(), current_stream(x.device)))
    return x


def _per_sample(fn_name: str, a0: torch.Tensor, a1: torch.Tensor, ca: torch.Tensor, cs: torch.Tensor, out=None) -> torch.Tensor:
    lib = load_library()
    a0 = require_dev(a0, "x"); a1 = require_dev(a1, "y"); ca = require_dev(ca, "a"); cs = require_dev(cs, "s")
    B = a0.shape[0]
    if out is None:
        out = torch.empty_like(a0)
    with torch.cuda.device(a0.device):
        check(getattr(lib, fn_name)(out.data_ptr(), a0.data_ptr(), a1.data_ptr(), ca.data_ptr(), cs.data_ptr(), B,
                                    a0.numel() // B, current_stream(a0.device)))
    return out


def q_sample(x0, noise, a, s, out=None):
    return _per_sample("ccn_q_sample", x0, noise, a, s, out)


def predict_x0(x_t, eps, a, s):
    return _per_sample("ccn_predict_x0", x_t, eps, a, s)


def timestep_embedding(t: torch.Tensor, dim: int) -> torch.Tensor:
    lib = load_library()
    t = require_dev(t, "t", torch.int64)
    out = torch.empty((t.shape[0], dim), dtype=torch.float32, device=t.device)
    with torch.cuda.device(t.device):
        check(lib.ccn_timestep_embedding(t.data_ptr(), out.data_ptr(), t.shape[0], dim, current_stream(t.device)))
    return out


def film_forward(x, h, ws_, bs_, wh_, bh_):
    lib = load_library()
    x = require_dev(x, "x"); h = require_dev(h, "h")
    ws_, bs_, wh_, bh_ = (require_dev(v, "film parameter") for v in (ws_, bs_, wh_, bh_))
    B, C, H, W = x.shape
    D = h.shape[1]
    y = torch.empty_like(x)
    scratch = torch.empty(2 * B * C, dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        check(lib.ccn_film_forward(x.data_ptr(), h.data_ptr(), ws_.data_ptr(), bs_.data_ptr(), wh_.data_ptr(), bh_.data_ptr(),
                                   y.data_ptr(), B, C, H, W, D, scratch.data_ptr(), current_stream(x.device)))
    return y
