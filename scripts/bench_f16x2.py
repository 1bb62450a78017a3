"""A/B measurement of the fp16 x 2 split GEMM (EXPERIMENT 2, csrc/gemm_f16x2.hip) against the f32-MFMA kernel and the
bf16 x 3 form on the eleven pointwise-convolution shapes of the batch-1024 step (forward form)."""
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
def pow2scale(t):
    m = float(t.abs().max())
    return 1.0 if m == 0 else 2.0 ** (14 - math.floor(math.log2(m)))
P, I, F = ctypes.c_void_p * 1, ctypes.c_int * 1, ctypes.c_float * 1
t1s = t3s = t2s = fl = 0.0
for L, K, N in shapes:
    M = B * L
    A = torch.randn(M, K, device='cuda'); W = torch.randn(K, N, device='cuda') * 0.1
    C = torch.empty(M, N, device='cuda'); C2 = torch.empty(M, N, device='cuda')
    Wp3 = torch.empty((3, N, K), dtype=torch.bfloat16, device='cuda')
    _lib.call("kws_bf16x3_split_batch", P(W.data_ptr()), P(Wp3.data_ptr()), I(K), I(N), I(1), 1, S)
    sA, sB = pow2scale(A), pow2scale(W)
    Wp2 = torch.empty((2, N, K), dtype=torch.float16, device='cuda')
    _lib.call("kws_f16x2_split_batch", P(W.data_ptr()), P(Wp2.data_ptr()), I(K), I(N), I(1), F(sB), 1, S)
    t1 = timeit(lambda: _lib.call("kws_gemm_nn_f32", _lib.ptr(A), _lib.ptr(W), _lib.ptr(C), M, K, N, None, S))
    t3 = timeit(lambda: _lib.call("kws_gemm_nn_bf16x3p_f32", _lib.ptr(A), _lib.ptr(Wp3), _lib.ptr(C), M, K, N, None, S))
    t2 = timeit(lambda: _lib.call("kws_gemm_nn_f16x2_f32", _lib.ptr(A), _lib.ptr(Wp2), _lib.ptr(C2), M, K, N, ctypes.c_float(sA), ctypes.c_float(sB), None, S))
    ref = (A[:4096].double() @ W.double())
    e2 = float((C2[:4096].double() - ref).abs().max() / ref.abs().max()); e3 = float((C[:4096].double() - ref).abs().max() / ref.abs().max())
    f = 2.0 * M * K * N
    print("M=%7d K=%3d N=%3d  f32 MFMA %6.1f us %6.1f TF | bf16x3 %6.1f us x%.2f (err %.1e) | f16x2 %6.1f us %6.1f TF-eq x%.2f (err %.1e)" % (
        M, K, N, t1 * 1e3, f / t1 / 1e9, t3 * 1e3, t1 / t3, e3, t2 * 1e3, f / t2 / 1e9, t1 / t2, e2))
    t1s += t1; t3s += t3; t2s += t2; fl += f
print("total: f32 MFMA %.3f ms | bf16x3 %.3f ms x%.2f | f16x2 %.3f ms (%.1f TF-equivalent) x%.2f" % (t1s, t3s, t1s / t3s, t2s, fl / t2s / 1e9, t1s / t2s))
