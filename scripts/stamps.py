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
nb = min(8192, ((M + 127) // 128 + 7) // 8 * 8 * ((N + 127) // 128))
t = buf[:nb, :5].astype(np.int64)
ok = t[:, 4] > t[:, 0]
t = t[ok]
d = np.diff(t, axis=1)
print("blocks", len(t), "median cycles: prologue %d  mainloop %d  store %d  stats %d  total %d" % tuple(list(np.median(d, axis=0)) + [np.median(t[:, 4] - t[:, 0])]))
t0 = t[:, 0].min()
print("kernel span (memtime ticks)", t[:, 4].max() - t0, " first-round start spread", np.percentile(t[:, 0] - t0, [50, 90, 99]))
