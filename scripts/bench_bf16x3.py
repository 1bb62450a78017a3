"""A/B measurement of the bf16 x 3 split GEMM (EXPERIMENT, csrc/gemm_bf16x3.hip) against the f32-MFMA kernel on the
eleven pointwise-convolution shapes of the batch-1024 step (forward and weight-gradient forms; algorithmic 2 M K N FLOPs per launch)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes
import torch
from speech_recognition_amd import _lib
lib = _lib.load()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
shapes = [(397,128,128),(199,128,192),(197,192,192),(99,192,256),(97,256,256),(49,256,320),(47,320,320),(24,320,384),(22,384,384),(11,384,512),(9,512,512)]
S = _lib.stream_ptr()
def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
def split_planes(W, transpose):
    """bf16 planes [3][N][K] of W^T (transpose) or of W, through kws_bf16x3_split_batch"""
    R, Cn = W.shape
    out = torch.empty((3, Cn, R) if transpose else (3, R, Cn), dtype=torch.bfloat16, device='cuda')
    P, I = ctypes.c_void_p * 1, ctypes.c_int * 1
    _lib.call("kws_bf16x3_split_batch", P(W.data_ptr()), P(out.data_ptr()), I(R), I(Cn), I(int(transpose)), 1, S)
    return out
t1s = tps = fl = tw1s = tw3s = 0.0
for L, K, N in shapes:
    M = B * L
    A = torch.randn(M, K, device='cuda'); W = torch.randn(K, N, device='cuda') * 0.1; Wt = W.t().contiguous()
    C = torch.empty(M, N, device='cuda')
    t1 = timeit(lambda: _lib.call("kws_gemm_nn_f32", _lib.ptr(A), _lib.ptr(W), _lib.ptr(C), M, K, N, None, S))
    Wp = split_planes(W, True)
    tp = timeit(lambda: _lib.call("kws_gemm_nn_bf16x3p_f32", _lib.ptr(A), _lib.ptr(Wp), _lib.ptr(C), M, K, N, None, S))
    G = torch.randn(M, N, device='cuda') * 0.1
    D = torch.empty(K, N, device='cuda')
    ws1 = torch.empty(lib.kws_gemm_tn_workspace_floats(M, K, N), device='cuda')
    ws3 = torch.empty(lib.kws_gemm_tn_bf16x3_workspace_floats(M, K, N), device='cuda')
    tw1 = timeit(lambda: _lib.call("kws_gemm_tn_f32", _lib.ptr(A), _lib.ptr(G), _lib.ptr(D), M, K, N, _lib.ptr(ws1), S))
    tw3 = timeit(lambda: _lib.call("kws_gemm_tn_bf16x3_f32", _lib.ptr(A), _lib.ptr(G), _lib.ptr(D), M, K, N, _lib.ptr(ws3), S))
    tw1s += tw1; tw3s += tw3
    f = 2.0 * M * K * N
    byts = 4.0 * (M * K + K * N + M * N)
    print("M=%7d K=%3d N=%3d  fwd f32 MFMA %6.1f us %6.1f TF | bf16x3 %6.1f us %6.1f TF-eq, %5.2f TB/s algorithmic x%.2f" % (
        M, K, N, t1 * 1e3, f / t1 / 1e9, tp * 1e3, f / tp / 1e9, byts / tp / 1e9, t1 / tp) +
        " || wgrad f32 %6.1f us %6.1f TF | bf16x3 %6.1f us %6.1f TF-eq x%.2f" % (tw1 * 1e3, f / tw1 / 1e9, tw3 * 1e3, f / tw3 / 1e9, tw1 / tw3))
    t1s += t1; tps += tp; fl += f
print("total: fwd f32 MFMA %.3f ms (%.1f TF) | bf16x3 %.3f ms (%.1f TF-equivalent) x%.2f" % (
    t1s, fl / t1s / 1e9, tps, fl / tps / 1e9, t1s / tps) +
    " || wgrad f32 %.3f ms | bf16x3 %.3f ms x%.2f" % (tw1s, tw3s, tw1s / tw3s))
