"""Reads the in-kernel stamps of the wave-specialised NN GEMM (library built with -DKWS_GEMM_STAMP):
KWS_LIB_PATH=variants/libkws_wsstamp.so KWS_GEMM_WS=1 python scripts/stamps_ws.py M K N"""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from speech_recognition_amd import _lib
lib = _lib.load()
M, K, N = [int(v) for v in sys.argv[1:4]]
S = _lib.stream_ptr()
A = torch.randn(M, K, device='cuda'); W = torch.randn(K, N, device='cuda') * 0.1; C = torch.empty(M, N, device='cuda')
part = torch.empty(lib.kws_gemm_num_row_tiles(M) * 2 * N, device='cuda')
for _ in range(3):
    _lib.call("kws_gemm_nn_f32", _lib.ptr(A), _lib.ptr(W), _lib.ptr(C), M, K, N, None if os.environ.get("NOSTATS") else _lib.ptr(part), S)
torch.cuda.synchronize()
buf = np.zeros((8192, 8), dtype=np.uint64)
raw = ctypes.CDLL(_lib.LIB_PATH)
raw.kws_debug_read_stamps(buf.ctypes.data_as(ctypes.c_void_p))
t = buf[:256].astype(np.float64)
st = buf[256:512, 0].astype(np.float64)
t = t[t[:, 3] > 0]
it = t[:, 3:4]
print("%s M=%d K=%d N=%d WGs %d iters/WG %.1f | per iteration (cycles): mma %.0f barrier %.0f staging %.0f | loader (its turns only, per iteration of the WG): wait+write %.0f issue %.0f | total/iter %.0f cycles, %.3f us -> clock %.2f GHz" % (
    os.environ.get("KWS_LIB_PATH", ""), M, K, N, len(t), it.mean(), np.median(t[:, 0:1] / it), np.median(t[:, 1:2] / it), np.median(t[:, 2:3] / it),
    np.median(t[:, 6:7] / it), np.median(t[:, 7:8] / it), np.median(t[:, 4:5] / it), np.median(t[:, 5:6] / it) / 100.0,
    np.median(t[:, 4] / t[:, 5]) * 0.1))
tiles = it[:, 0] / (K // 32)
print("   storer: %.0f cycles per store_c (tiles/WG %.1f)" % (np.median(st[:len(t)] / np.maximum(tiles - 1, 1)), tiles.mean()))
sb = buf[256:512, 1].astype(np.float64)[:len(t)]; lb = buf[256:512, 2].astype(np.float64)[:len(t)]
print("   waiting at the barrier, cycles per iteration: MFMA wave 0 %.0f, loader wave 0 %.0f, storer wave 0 %.0f" % (
    np.median(t[:, 1] / it[:, 0]), np.median(lb / it[:, 0]), np.median(sb / it[:, 0])))
wb = buf[512:768].astype(np.float64)[:len(t)]
print("   barrier wait per iteration by wave (0-3 MFMA, 4-5 loaders, 6-7 storers): " + " ".join("%.0f" % np.median(wb[:, w] / it[:, 0]) for w in range(8)))
