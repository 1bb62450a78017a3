"""CPU-only checks of the drop-in boundary: the C-ABI library builds/loads here (hipcc cross-compiles
gfx950 without a GPU) and exports exactly the symbols include/kws_hip.h declares."""
import ctypes
import os
import re

import pytest

from speech_recognition_amd import _lib


def _header_functions(root):
    src = open(os.path.join(root, "include", "kws_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(kws_[a-z0-9_]+)\s*\(", src)))


def test_header_and_binding_agree(repo_root):
    names = _header_functions(repo_root)
    assert len(names) >= 30
    assert sorted(_lib.SIGNATURES.keys()) == names


def test_library_loads_and_exports_every_symbol(repo_root):
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in _header_functions(repo_root):
        assert hasattr(lib, name), name
    assert _lib.load().kws_abi_version() == _lib.ABI_VERSION == 2


def test_product_path_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from speech_recognition_amd.net import DeviceNet
    with pytest.raises(_lib.KwsError):
        DeviceNet(_lib.KWS_NET_TS_ATTENTION, 12)
