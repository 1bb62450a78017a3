"""NN GEMM (forward with BN stats / dgrad without) timing on a few layer shapes; used for kernel A/B runs:
KWS_LIB_PATH=variants/libkws_X.so KWS_GEMM_WS=1 python scripts/bench_nn.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speech_recognition_amd import _lib
lib = _lib.load()
shapes = [(406528, 128, 128), (201728, 192, 192), (99328, 256, 256), (48128, 320, 320), (9216, 512, 512)]
S = _lib.stream_ptr()


def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n


out = []
for M, K, N in shapes:
    A = torch.randn(M, K, device='cuda'); W = torch.randn(K, N, device='cuda') * 0.1; C = torch.empty(M, N, device='cuda')
    part = torch.empty(lib.kws_gemm_num_row_tiles(M) * 2 * N, device='cuda')
    f_stats = lambda: _lib.call("kws_gemm_nn_f32", _lib.ptr(A), _lib.ptr(W), _lib.ptr(C), M, K, N, _lib.ptr(part), S)
    f_plain = lambda: _lib.call("kws_gemm_nn_f32", _lib.ptr(A), _lib.ptr(W), _lib.ptr(C), M, K, N, None, S)
    if os.environ.get("PLAIN_FIRST"):
        t2 = timeit(f_plain); t1 = timeit(f_stats)
    else:
        t1 = timeit(f_stats); t2 = timeit(f_plain)
    f = 2.0 * M * K * N
    out.append("K=%d: %.0f/%.0f us (%.0f/%.0f TF)" % (K, t1 * 1e3, t2 * 1e3, f / t1 / 1e9, f / t2 / 1e9))
print(os.environ.get("KWS_LIB_PATH", "default"), " | ".join(out))
