"""Host-side reconstruction metrics with the reference's function names (eval/metrics.py:16-85).

``psnr`` / ``_to_uint8`` follow the reference exactly (uint8 conversion by truncation, PSNR on
the uint8 images, ``inf`` for identical images) -- PSNR is the parity metric of this build.
``ssim`` uses scikit-image when it is importable (as the reference does) and otherwise an own scipy
implementation of ``skimage.metrics.structural_similarity``'s defaults (7x7 uniform window,
K1=0.01, K2=0.03, sample covariance, data_range=255, mean over channels and the valid interior);
scikit-image is absent from the build image and the reference holds no SSIM fixture, so that branch is
*parity unpinned*.  ``lpips_distance`` / ``clip_similarity`` need pretrained networks fetched by name;
offline they return NaN (the reference returns NaN for LPIPS and raises ImportError for CLIP) and
the aggregate skips NaNs like ``cli/eval.py:77-79``.
"""
from __future__ import annotations

import numpy as np


def _to_uint8(img: np.ndarray) -> np.ndarray:
    return ((img + 1.0) * 127.5).clip(0, 255).astype(np.uint8)


def psnr_u8(x1: np.ndarray, x2: np.ndarray) -> float:
    """PSNR of two uint8 images -- what ``psnr`` computes after its uint8 conversion (eval/metrics.py:22-29)."""
    d = x1.astype(np.float32) - x2.astype(np.float32)
    mse = np.mean(d ** 2)
    if mse == 0:
        return float("inf")
    return 20.0 * np.log10(255.0 / np.sqrt(mse))


def psnr(img1: np.ndarray, img2: np.ndarray) -> float:
    return psnr_u8(_to_uint8(img1), _to_uint8(img2))


def _ssim_plane(a: np.ndarray, b: np.ndarray, data_range: float = 255.0, win: int = 7) -> float:
    from scipy.ndimage import uniform_filter
    a = a.astype(np.float64); b = b.astype(np.float64)
    npx = win * win
    cov_norm = npx / (npx - 1.0)
    ua, ub = uniform_filter(a, win), uniform_filter(b, win)
    uaa, ubb, uab = uniform_filter(a * a, win), uniform_filter(b * b, win), uniform_filter(a * b, win)
    va, vb, vab = cov_norm * (uaa - ua * ua), cov_norm * (ubb - ub * ub), cov_norm * (uab - ua * ub)
    c1, c2 = (0.01 * data_range) ** 2, (0.03 * data_range) ** 2
    s = ((2 * ua * ub + c1) * (2 * vab + c2)) / ((ua ** 2 + ub ** 2 + c1) * (va + vb + c2))
    pad = (win - 1) // 2
    return float(s[pad:-pad, pad:-pad].mean())


def ssim(img1: np.ndarray, img2: np.ndarray) -> float:
    return ssim_u8(_to_uint8(img1), _to_uint8(img2))


def ssim_u8(x1: np.ndarray, x2: np.ndarray) -> float:
    """SSIM of two uint8 images (CHW or HWC) -- the part of ``ssim`` after its uint8 conversion."""
    if x1.ndim == 3 and x1.shape[0] in (1, 3):
        x1, x2 = x1.transpose(1, 2, 0), x2.transpose(1, 2, 0)
    try:
        from skimage.metrics import structural_similarity
        multi = x1.ndim == 3 and x1.shape[2] > 1
        return float(structural_similarity(x1, x2, data_range=255, channel_axis=-1 if multi else None))
    except ImportError:
        pass
    try:
        if x1.ndim == 2:
            return _ssim_plane(x1, x2)
        return float(np.mean([_ssim_plane(x1[..., c], x2[..., c]) for c in range(x1.shape[2])]))
    except Exception:
        return float("nan")


def learned_metrics_available() -> bool:
    """True if LPIPS or CLIP similarity could produce a number here (their packages import); offline both are NaN."""
    for mod in ("lpips", "open_clip"):
        try:
            __import__(mod)
            return True
        except Exception:
            pass
    return False


def lpips_distance(img1: np.ndarray, img2: np.ndarray, device: str = "cpu") -> float:
    try:
        import lpips  # noqa: F401  (needs downloaded VGG weights)
    except Exception:
        return float("nan")
    import torch
    t1 = torch.from_numpy(img1).float().unsqueeze(0)
    t2 = torch.from_numpy(img2).float().unsqueeze(0)
    if t1.shape[1] != 3:
        raise ValueError("LPIPS expects 3-channel images")
    try:
        fn = lpips.LPIPS(net="vgg").to(device)
        return float(fn(t1.to(device), t2.to(device)).item())
    except Exception:
        return float("nan")


def clip_similarity(img1: np.ndarray, img2: np.ndarray, device: str = "cpu") -> float:
    try:
        import open_clip  # noqa: F401  (needs downloaded ViT-B-32 weights)
    except Exception:
        return float("nan")
    import torch
    from PIL import Image
    try:
        model, _, preprocess = open_clip.create_model_and_transforms("ViT-B-32", pretrained="openai")
        model = model.to(device).eval()
        feats = []
        with torch.no_grad():
            for img in (img1, img2):
                vis = _to_uint8(img)
                if vis.ndim == 3 and vis.shape[0] in (1, 3):
                    vis = vis.transpose(1, 2, 0)
                f = model.encode_image(preprocess(Image.fromarray(vis)).unsqueeze(0).to(device)).float()
                feats.append(f / f.norm(dim=-1, keepdim=True))
        return float((feats[0] * feats[1]).sum().item())
    except Exception:
        return float("nan")
