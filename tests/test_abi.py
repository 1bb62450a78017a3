"""CPU-only checks of the drop-in boundary: the C-ABI library builds/loads here (hipcc cross-compiles
gfx950 without a GPU) and exports exactly the symbols include/kws_hip.h declares."""
import ctypes
import os
import re
import subprocess

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
    names = _header_functions(repo_root)
    for name in names:
        assert hasattr(lib, name), name
    # ... and nothing else: the library is built with -fvisibility=hidden, the header pushes default visibility
    # around its declarations, so cross-TU helpers (csrc/internal.h) stay out of the dynamic symbol table
    out = subprocess.check_output(["nm", "-D", "--defined-only", _lib.LIB_PATH], text=True)
    exported = sorted(line.split()[-1] for line in out.splitlines() if line.split()[-1].startswith("kws_"))
    assert exported == names, sorted(set(exported) ^ set(names))
    assert _lib.load().kws_abi_version() == _lib.ABI_VERSION == 5


def test_product_path_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from speech_recognition_amd.net import DeviceNet
    with pytest.raises(_lib.KwsError):
        DeviceNet(_lib.KWS_NET_TS_ATTENTION, 12)


def test_profiler_handle_refuses_destroy_while_another_thread_is_attached():
    """a handle destroyed under an attached thread left that thread a dangling pointer.  Host-side only."""
    import threading
    lib = _lib.load()
    prof = _lib.Profiler()
    attached, release = threading.Event(), threading.Event()

    def worker():
        prof.attach()
        attached.set()
        release.wait(10)
        _lib.Profiler.detach()

    t = threading.Thread(target=worker)
    t.start()
    assert attached.wait(10)
    prof.attach()                                   # this thread too: destroy ends the caller's own attachment
    with pytest.raises(_lib.KwsError):
        prof.close()
    assert prof.handle is not None
    release.set()
    t.join()
    prof.close()
    assert prof.handle is None
    # a thread that EXITS while attached detaches itself
    prof2 = _lib.Profiler()
    t2 = threading.Thread(target=prof2.attach)
    t2.start()
    t2.join()
    # (join() returns when the Python thread state is gone; the C++ thread_local destructor runs a moment later, at pthread exit)
    import time
    for _ in range(200):
        try:
            prof2.close()
            break
        except _lib.KwsError:
            time.sleep(0.01)
    assert prof2.handle is None
    assert lib.kws_profiler_attach(None) == 0
