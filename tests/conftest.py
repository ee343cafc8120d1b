"""Test configuration: path setup, the ``gpu`` marker, shared fixtures.

``-m "not gpu"`` : oracle vs golden vectors, host logic, C-ABI symbol export, gloo world_size-2 sharding.
``-m gpu``       : parity of the HIP path (through the C ABI) against the oracle and the golden vectors.
"""
import sys
from pathlib import Path

import numpy as np
import pytest
import torch

REPO = Path(__file__).resolve().parent.parent
PKG = REPO / "clip-neural-image-conpression_amd"
GOLDEN = Path(__file__).resolve().parent / "golden"
for p in (str(REPO), str(PKG)):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no HIP device in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden():
    def load(name):
        return np.load(GOLDEN / name, allow_pickle=False)
    return load


@pytest.fixture(scope="session")
def synth():
    from clip_feature_codec.utils import synth as s
    return s


@pytest.fixture(scope="session")
def tiny_sd(synth):
    """Key-seeded weights of the C1 architecture (base 32, ch_mult (1,2))."""
    return synth.synth_state_dict(synth.unet_param_spec(512, 32, (1, 2)))


def bits_or_close(a, b, rtol=1e-4, atol=1e-6, what=""):
    """Bit equality where this host reproduces the build container's torch-CPU rounding (it does there: the
    fixtures were generated on it); on another CPU model torch's vectorised cos/exp/conv kernels may round
    differently in the last bits, so fall back to a tight tolerance and say so."""
    import warnings
    a, b = np.asarray(a), np.asarray(b)
    if np.array_equal(a, b, equal_nan=True):
        return
    assert a.shape == b.shape and np.allclose(a, b, rtol=rtol, atol=atol, equal_nan=True), f"{what}: max abs diff {np.abs(a - b).max()}"
    warnings.warn(f"{what}: equal to the golden vector only within rtol={rtol} on this host CPU (not bit-identical)")
