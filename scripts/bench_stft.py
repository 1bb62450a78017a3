"""Standalone timing of the STFT+mel(+DCT) stage (a3-a5): algorithmic GB/s vs the HBM roof."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from speech_recognition_amd import _lib
from speech_recognition_amd.features import path_b_tables
lib = _lib.load()
S = _lib.stream_ptr()
cases = [(1024, 80, 60), (2048, 40, 40), (8192, 80, 60)]
if len(sys.argv) > 1:                     # e.g. `bench_stft.py 1024,80,60`: one case only (the PMC passes)
    cases = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]]
for (B, n_mel, n_out) in cases:
    t = path_b_tables(480, n_mel, n_out)
    plan = ctypes.c_void_p()
    _lib.check(lib.kws_stft_plan_create(480, 160, 512, n_mel, n_out, t['window'].ctypes.data_as(ctypes.c_void_p),
               t['mel'].ctypes.data_as(ctypes.c_void_p), t['dct'].ctypes.data_as(ctypes.c_void_p), 1e-6, 0.0, ctypes.byref(plan)), "plan")
    x = torch.randn(B, 16000, device='cuda') * 0.08
    out = torch.empty(B, 98 * n_out, device='cuda')
    f = lambda: _lib.call("kws_stft_mel_f32", plan, _lib.ptr(x), B, 16000, _lib.ptr(out), 0, S)
    for _ in range(3): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20): f()
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 20
    byts = B * (16000 * 4 + 98 * n_out * 4)
    print("B=%d M=%d K=%d: %.1f us  %.2f M clips/s  %.0f GB/s algorithmic = %.1f%% of 8 TB/s" % (B, n_mel, n_out, ms * 1e3, B / ms / 1e3, byts / ms / 1e6, byts / ms / 1e6 / 80))
