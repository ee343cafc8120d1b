"""clip_feature_codec -- MI355X-native build of the DDIM reconstruction path.

Same package / module / class names as the reference's hot path, so
``python -m clip_feature_codec.cli.eval`` and ``from clip_feature_codec.models.unet import CLIPCondUNet``
work unchanged with this directory first on ``PYTHONPATH``.  All arithmetic runs in libccn_hip.so
(hand-written gfx950 kernels, C ABI in include/ccn_hip.h); there is no CPU fallback.
"""
__version__ = "0.3.0+mi355x.r1"
