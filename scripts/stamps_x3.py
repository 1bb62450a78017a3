"""Reads the in-kernel phase stamps of gemm_nn_bf16x3p_kernel / gemm_nn_f16x2_kernel (library built with -DKWS_X3_STAMP):
scripts/build_variant.sh x3stamp "-DKWS_X3_STAMP" gemm_bf16x3 && KWS_LIB_PATH=variants/libkws_x3stamp.so python scripts/stamps_x3.py
scripts/build_variant.sh h2stamp "-DKWS_X3_STAMP" gemm_f16x2 && KWS_LIB_PATH=variants/libkws_h2stamp.so python scripts/stamps_x3.py f16x2"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from speech_recognition_amd import _lib
lib = _lib.load()
S = _lib.stream_ptr()
raw = ctypes.CDLL(_lib.LIB_PATH)
H2 = len(sys.argv) > 1 and sys.argv[1] == "f16x2"
names = ["entry -> first slab staged", "products (all slabs)", "barrier after products", "wait + split + store", "barrier after store", "epilogue"]
for L, K, N in [(397, 128, 128), (197, 192, 192), (97, 256, 256), (47, 320, 320), (9, 512, 512)]:
    M = 1024 * L
    A = torch.randn(M, K, device='cuda'); W = torch.randn(K, N, device='cuda') * 0.1
    C = torch.empty(M, N, device='cuda')
    P, I = ctypes.c_void_p * 1, ctypes.c_int * 1
    buf = np.zeros((2048, 8), dtype=np.uint64)
    if H2:
        slots = torch.zeros((2, 256), dtype=torch.int32, device='cuda')
        P2, L2 = ctypes.c_void_p * 2, ctypes.c_int64 * 2
        _lib.call("kws_absmax_batch_f32", P2(A.data_ptr(), W.data_ptr()), L2(A.numel(), W.numel()), _lib.ptr(slots), 2, S)
        Wp = torch.empty((2, N, K), dtype=torch.float16, device='cuda')
        _lib.call("kws_f16x2_split_batch", P(W.data_ptr()), P(Wp.data_ptr()), I(K), I(N), I(1), P(slots[1].data_ptr()), 1, S)
        for _ in range(20):
            _lib.call("kws_gemm_nn_f16x2_f32", _lib.ptr(A), _lib.ptr(Wp), _lib.ptr(C), M, K, N, _lib.ptr(slots[0]), _lib.ptr(slots[1]), None, S)
        torch.cuda.synchronize()
        raw.kws_debug_read_h2_stamps(buf.ctypes.data_as(ctypes.c_void_p))
    else:
        Wp = torch.empty((3, N, K), dtype=torch.bfloat16, device='cuda')
        _lib.call("kws_bf16x3_split_batch", P(W.data_ptr()), P(Wp.data_ptr()), I(K), I(N), I(1), 1, S)
        for _ in range(20):
            _lib.call("kws_gemm_nn_bf16x3p_f32", _lib.ptr(A), _lib.ptr(Wp), _lib.ptr(C), M, K, N, None, S)
        torch.cuda.synchronize()
        raw.kws_debug_read_x3_stamps(buf.ctypes.data_as(ctypes.c_void_p))
    n_t = min(2048, ((M + 127) // 128) * ((N + 127) // 128))
    t = buf[:n_t].astype(np.float64)
    G = K // 32
    print("M=%d K=%d N=%d (%d slabs per tile): tile %.0f cycles = %.2f us, clock %.2f GHz" % (
        M, K, N, G, np.median(t[:, 6]), np.median(t[:, 7]) / 100.0, np.median(t[:, 6] / t[:, 7]) * 0.1))
    for i, nm in enumerate(names):
        per = G if i in (1, 2) else (G - 1 if i in (3, 4) else 1)
        print("  %-28s %8.0f cycles (%6.0f per slab)" % (nm, np.median(t[:, i]), np.median(t[:, i]) / per))
