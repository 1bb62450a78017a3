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
    _lib.call("kws_gemm_nn_f32", _lib.ptr(A), _lib.ptr(W), _lib.ptr(C), M, K, N, _lib.ptr(part), S)
torch.cuda.synchronize()
buf = np.zeros((8192, 8), dtype=np.uint64)
raw = ctypes.CDLL(_lib.LIB_PATH)
raw.kws_debug_read_stamps(buf.ctypes.data_as(ctypes.c_void_p))
t = buf[:512].astype(np.int64)
t = t[t[:, 5] > 0]
per = t[:, :5] / t[:, 5:6]
pk = t[:, 7]; e = np.stack([pk & 0xFFFFF, (pk >> 20) & 0xFFFFF, (pk >> 40) & 0xFFFFF], 1) / t[:, 5:6]
print("  epilogue split per tile: reg->LDS(+stats regs) %.0f | LDS->global stores %.0f | stats sync+store %.0f  (then sync+store_lds(0)+sync = rest of epilogue)" % tuple(np.median(e, axis=0)))
print("WGs %d tiles/WG %.1f | per tile cycles: load-issue %.0f compute %.0f store_lds(vmcnt wait) %.0f barrier %.0f epilogue %.0f | total/tile %.0f" % (
    len(t), t[:, 5].mean(), *np.median(per, axis=0), np.median(t[:, 6] / t[:, 5])))
