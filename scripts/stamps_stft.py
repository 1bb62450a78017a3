"""Reads the in-kernel phase stamps of stft4_kernel (library built with -DKWS_STFT_STAMP):
scripts/build_variant.sh stftstamp "-DKWS_STFT_STAMP" stft4 && KWS_LIB_PATH=variants/libkws_stftstamp.so python scripts/stamps_stft.py"""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from speech_recognition_amd import _lib
from speech_recognition_amd.features import path_b_tables
lib = _lib.load()
S = _lib.stream_ptr()
B, n_mel, n_out = 1024, 80, 60
t = path_b_tables(480, n_mel, n_out)
plan = ctypes.c_void_p()
_lib.check(lib.kws_stft_plan_create(480, 160, 512, n_mel, n_out, t['window'].ctypes.data_as(ctypes.c_void_p),
           t['mel'].ctypes.data_as(ctypes.c_void_p), t['dct'].ctypes.data_as(ctypes.c_void_p), 1e-6, 0.0, ctypes.byref(plan)), "plan")
x = torch.randn(B, 16000, device='cuda') * 0.08
out = torch.empty(B, 98 * n_out, device='cuda')
for _ in range(5):
    _lib.call("kws_stft_mel_f32", plan, _lib.ptr(x), B, 16000, _lib.ptr(out), 0, S)
torch.cuda.synchronize()
buf = np.zeros((256, 12), dtype=np.uint64)
raw = ctypes.CDLL(_lib.LIB_PATH)
raw.kws_debug_read_stft_stamps(buf.ctypes.data_as(ctypes.c_void_p))
t = buf.astype(np.float64)
t = t[t[:, 5] > 0]
n = t[:, 5:6]
names = ["load+MFMA pass 1", "twiddle+FFT16", "split+|X|", "mel+log", "DCT+store (per quad)"]
per = np.median(t[:, :5] / n, axis=0)
print("workgroups %d, passes/wave %.1f, total %.0f cycles = %.1f us, clock %.2f GHz" % (len(t), n.mean(), np.median(t[:, 6]), np.median(t[:, 7]) / 100.0, np.median(t[:, 6] / t[:, 7]) * 0.1))
for nm, v in zip(names, per):
    print("  %-24s %8.0f cycles per pass" % (nm, v))
print("  sum %.0f of %.0f cycles per pass" % (per.sum(), np.median(t[:, 6] / n[:, 0])))
e, l0, l1, lw = t[:, 8], t[:, 9], t[:, 10], t[:, 11]
print("absolute (100 MHz ticks -> us): first entry 0, last entry %.1f, median prologue %.1f, wave-0 loop ends %.1f .. %.1f, last wave of any workgroup ends %.1f"
      % ((e.max() - e.min()) / 100.0, np.median(l0 - e) / 100.0, (l1.min() - e.min()) / 100.0, (l1.max() - e.min()) / 100.0, (lw.max() - e.min()) / 100.0))
