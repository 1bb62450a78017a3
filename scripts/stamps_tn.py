"""Reads the in-kernel stamps of the wave-specialised TN (weight-gradient) GEMM (library built with -DKWS_GEMM_STAMP):
scripts/build_variant.sh wsstamp "-DKWS_GEMM_STAMP" gemm && KWS_LIB_PATH=variants/libkws_wsstamp.so python scripts/stamps_tn.py M K N"""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from speech_recognition_amd import _lib
lib = _lib.load()
M, K, N = [int(v) for v in sys.argv[1:4]]
S = _lib.stream_ptr()
A = torch.randn(M, K, device='cuda'); G = torch.randn(M, N, device='cuda') * 1e-3; dW = torch.empty(K, N, device='cuda')
ws = torch.empty(int(lib.kws_gemm_tn_workspace_floats(M, K, N)), device='cuda')
for _ in range(3):
    _lib.call("kws_gemm_tn_f32", _lib.ptr(A), _lib.ptr(G), _lib.ptr(dW), M, K, N, _lib.ptr(ws), S)
torch.cuda.synchronize()
buf = np.zeros((8192, 8), dtype=np.uint64)
raw = ctypes.CDLL(_lib.LIB_PATH)
raw.kws_debug_read_stamps(buf.ctypes.data_as(ctypes.c_void_p))
t = buf[:4096].astype(np.float64)
t = t[t[:, 3] > 0]
g = t[:, 3]
print("TN M=%d K=%d N=%d: %d work items, %.1f stages each | per stage (cycles): MFMA wave 0 issue %.0f, barrier wait %.0f | loader wave 0: work %.0f, barrier wait %.0f | loop %.0f cycles per stage, epilogue %.0f cycles per item" % (
    M, K, N, len(t), g.mean(), np.median(t[:, 0] / g), np.median(t[:, 1] / g), np.median(t[:, 6] / g), np.median(t[:, 7] / g),
    np.median(t[:, 4] / g), np.median(t[:, 5])))
if t[:, 2].max() > 0:
    tot = t[:, 4] + t[:, 5]
    print("   item lifetime %.0f k cycles = %.1f us (median; min %.1f max %.1f us) -> clock %.2f GHz" % (
        np.median(tot) / 1e3, np.median(t[:, 2]) / 100.0, t[:, 2].min() / 100.0, t[:, 2].max() / 100.0, np.median(tot / t[:, 2]) * 0.1))
