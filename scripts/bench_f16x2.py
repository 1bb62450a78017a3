"""A/B measurement of the fp16 x 2 split GEMM (A/B arm, csrc/gemm_f16x2.hip) against the f32-MFMA kernel on the eleven
pointwise-convolution shapes of the batch-1024 step (forward and weight-gradient forms)."""
import ctypes, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
def absmax_slots(*tensors):
    n = len(tensors)
    slots = torch.zeros((n, 256), dtype=torch.int32, device='cuda')
    PP, LL = ctypes.c_void_p * n, ctypes.c_int64 * n
    _lib.call("kws_absmax_batch_f32", PP(*[t.data_ptr() for t in tensors]), LL(*[t.numel() for t in tensors]), _lib.ptr(slots), n, S)
    return slots
P, I = ctypes.c_void_p * 1, ctypes.c_int * 1
t1s = t2s = fl = tw1s = tw2s = 0.0
for L, K, N in shapes:
    M = B * L
    A = torch.randn(M, K, device='cuda'); W = torch.randn(K, N, device='cuda') * 0.1
    C = torch.empty(M, N, device='cuda'); C2 = torch.empty(M, N, device='cuda')
    G = torch.randn(M, N, device='cuda') * 1e-6
    slots = absmax_slots(A, W, G)
    Wp2 = torch.empty((2, N, K), dtype=torch.float16, device='cuda')
    _lib.call("kws_f16x2_split_batch", P(W.data_ptr()), P(Wp2.data_ptr()), I(K), I(N), I(1), P(slots[1].data_ptr()), 1, S)
    t1 = timeit(lambda: _lib.call("kws_gemm_nn_f32", _lib.ptr(A), _lib.ptr(W), _lib.ptr(C), M, K, N, None, S))
    t2 = timeit(lambda: _lib.call("kws_gemm_nn_f16x2_f32", _lib.ptr(A), _lib.ptr(Wp2), _lib.ptr(C2), M, K, N, _lib.ptr(slots[0]), _lib.ptr(slots[1]), None, S))
    D = torch.empty(K, N, device='cuda')
    ws1 = torch.empty(lib.kws_gemm_tn_workspace_floats(M, K, N), device='cuda')
    ws2 = torch.empty(lib.kws_gemm_tn_f16x2_workspace_floats(M, K, N), device='cuda')
    tw1 = timeit(lambda: _lib.call("kws_gemm_tn_f32", _lib.ptr(A), _lib.ptr(G), _lib.ptr(D), M, K, N, _lib.ptr(ws1), S))
    tw2 = timeit(lambda: _lib.call("kws_gemm_tn_f16x2_f32", _lib.ptr(A), _lib.ptr(G), _lib.ptr(D), M, K, N, _lib.ptr(slots[0]), _lib.ptr(slots[2]), _lib.ptr(ws2), S))
    tw1s += tw1; tw2s += tw2
    ref = (A[:4096].double() @ W.double())
    e2 = float((C2[:4096].double() - ref).abs().max() / ref.abs().max()); e1 = float((C[:4096].double() - ref).abs().max() / ref.abs().max())
    f = 2.0 * M * K * N
    print("M=%7d K=%3d N=%3d  f32 MFMA %6.1f us %6.1f TF (err %.1e) | f16x2 %6.1f us %6.1f TF-eq x%.2f (err %.1e)" % (
        M, K, N, t1 * 1e3, f / t1 / 1e9, e1, t2 * 1e3, f / t2 / 1e9, t1 / t2, e2) +
        " || wgrad f32 %6.1f us | f16x2 %6.1f us x%.2f" % (tw1 * 1e3, tw2 * 1e3, tw1 / tw2))
    t1s += t1; t2s += t2; fl += f
print("total: f32 MFMA %.3f ms | f16x2 %.3f ms (%.1f TF-equivalent) x%.2f || wgrad f32 %.3f ms | f16x2 %.3f ms x%.2f" % (
    t1s, t2s, fl / t2s / 1e9, t1s / t2s, tw1s, tw2s, tw1s / tw2s))
