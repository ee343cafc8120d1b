"""clip_feature_codec -- MI355X-native build of the DDIM reconstruction path.

Same package / module / class names as the reference's hot path, so
``python -m clip_feature_codec.cli.eval`` and ``from clip_feature_codec.models.unet import CLIPCondUNet``
work unchanged with this directory first on ``PYTHONPATH``.  All arithmetic runs in libccn_hip.so
(hand-written gfx950 kernels, C ABI in include/ccn_hip.h); there is no CPU fallback.
"""
__version__ = "0.3.0+mi355x.r1"

import os as _os

# RCCL across processes needs dmabuf IPC on this pool's host driver (else "hipIpcGetMemHandle: invalid argument"); the variable is
# read when the HIP runtime starts, i.e. before any entry point of this package touches the GPU
_os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
