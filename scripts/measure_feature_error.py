"""Measured error of the STFT -> |X| -> mel -> log -> DCT kernels against the float64 oracle on the inputs of
tests/test_kernels_gpu.py::test_stft_mel_features (the four plan shapes): the numbers its tolerances are set from (2 x)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import features as OF
from speech_recognition_amd import _lib
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_kernels_gpu import _plan, dev, S

lib = _lib.load()
for path, n_mel, n_out, win, step in [("B", 80, 60, 480, 160), ("B", 40, 40, 480, 160), ("A", 40, 40, 480, 160), ("B", 80, 60, 400, 240)]:
    rng = np.random.RandomState(n_mel + win)
    B, L = 5, 16000
    t = np.arange(L) / 16000.0
    x = (rng.randn(B, L) * 0.0774 + 0.05 * np.sin(2 * np.pi * 440 * t)[None]).astype(np.float32)
    x[1] = 0.0
    tables = OF.tables_path_b(win, n_mel, n_out) if path == "B" else OF.tables_path_a(win, 16000, n_out, n_mel)
    plan = _plan(tables, step, n_mel, n_out)
    F = lib.kws_stft_num_frames(plan, L)
    mag_ref, logmel_ref, feat_ref = OF.features(x, tables, step, dtype=np.float64, return_all=True)
    dx = dev(x)
    errs = {}
    for name, kind, width, ref in (("spectrogram", 1, 257, mag_ref), ("log_mel", 2, n_mel, logmel_ref), ("mfcc", 0, n_out, feat_ref)):
        o = torch.full((B, F, width), float("nan"), device="cuda")
        _lib.call("kws_stft_mel_f32", plan, _lib.ptr(dx), B, L, _lib.ptr(o), kind, S())
        torch.cuda.synchronize()
        errs[name] = (float(np.abs(o.cpu().numpy() - ref).max()), float(np.abs(ref).max()))
    print("path %s mel %d out %d win %d step %d: " % (path, n_mel, n_out, win, step) +
          "  ".join("%s max_abs_err %.3g (max |ref| %.3g)" % (k, v[0], v[1]) for k, v in errs.items()), flush=True)
    lib.kws_stft_plan_destroy(plan)
