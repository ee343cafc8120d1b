"""The C-ABI library loads without a GPU and exports every symbol include/ccn_hip.h declares."""
import ctypes
import re
from pathlib import Path

import pytest

from clip_feature_codec import _native

REPO = Path(__file__).resolve().parent.parent
HEADER = REPO / "include" / "ccn_hip.h"


def declared_functions():
    text = re.sub(r"/\*.*?\*/", "", HEADER.read_text(), flags=re.S)
    return sorted(set(re.findall(r"\b(ccn_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_documented_surface():
    names = declared_functions()
    for must in ("ccn_create", "ccn_destroy", "ccn_load_param", "ccn_commit_params", "ccn_workspace_bytes", "ccn_forward",
                 "ccn_sample", "ccn_ddim_step", "ccn_q_sample", "ccn_last_error"):
        assert must in names


def test_library_exports_every_declared_symbol():
    if not _native.LIB_PATH.exists():
        pytest.fail(f"{_native.LIB_PATH} is not built: run `python __graft_entry__.py`")
    lib = ctypes.CDLL(str(_native.LIB_PATH))
    for name in declared_functions():
        assert hasattr(lib, name), f"{name} declared in ccn_hip.h but not exported"
        assert name in _native.SIGNATURES, f"{name} has no ctypes prototype in _native.SIGNATURES"
    assert set(_native.SIGNATURES) == set(declared_functions())
    _native.load_library()
    lib.ccn_version.restype = ctypes.c_char_p
    assert b"gfx950" in lib.ccn_version()


def test_config_struct_matches_header():
    assert ctypes.sizeof(_native.CcnConfig) == 4 * (3 + 8 + 4)
